"""GPU: the device building blocks in isolation (cross-lane reductions, split-fp16 MFMA tile
contractions incl. the transposing LDS read), against numpy."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import _lib
    return _lib.load()


def test_wave_ops(lib):
    rng = np.random.default_rng(0)
    for trial in range(4):
        x = rng.standard_normal(64).astype(np.float32)
        if trial == 3:
            x[[5, 37]] = x.max() + 1.0  # tie: the lowest index must win
        xd = torch.as_tensor(x, device="cuda:0")
        out = torch.full((384,), float("nan"), device="cuda:0")
        assert lib.ge2e_selftest_wave_ops(xd.data_ptr(), out.data_ptr(), None) == 0
        o = out.cpu().numpy()
        assert np.allclose(o[:64], x.astype(np.float64).sum(), rtol=1e-5, atol=1e-5)
        assert np.all(o[64:128] == x.max())
        assert np.allclose(o[128:192], np.repeat(x.reshape(4, 16).sum(1), 16), rtol=1e-5, atol=1e-5)
        assert np.allclose(o[192:256], np.repeat(x.reshape(16, 4).sum(1), 4), rtol=1e-5, atol=1e-5)
        assert np.all(o[256:320] == np.argmax(x))
        assert np.all(o[320:384] == np.repeat(4 * np.arange(16) + x.reshape(16, 4).argmax(1), 4))


@pytest.mark.parametrize("kind", ["int", "unit", "ties"])
def test_split_fp16_tile_contractions(lib, kind):
    rng = np.random.default_rng(1)
    if kind == "int":  # exactly representable: any fragment / transpose mix-up shows as O(1) errors
        A = rng.integers(-8, 9, (64, 256)).astype(np.float32) / 16
        Bm = rng.integers(-8, 9, (64, 256)).astype(np.float32) / 16
        G = rng.integers(0, 17, (64, 64)).astype(np.float32) / 16
        tol = 0.0
    elif kind == "ties":
        # x * 2^8 exactly half-way between two fp16 values: the residual must be taken against the
        # stored hi bits (hipcc otherwise converts twice with different tie rounding)
        base = rng.integers(1024, 2048, (64, 256)).astype(np.float64)  # 11-bit mantissas
        expo = rng.integers(-12, -3, (64, 256))
        A = ((base + 0.5) * 2.0 ** expo / 256 * rng.choice([-1, 1], (64, 256))).astype(np.float32)
        Bm = ((rng.integers(1024, 2048, (64, 256)) + 0.5) * 2.0 ** rng.integers(-12, -3, (64, 256)) / 256).astype(np.float32)
        G = ((rng.integers(1024, 2048, (64, 64)) + 0.5) * 2.0 ** rng.integers(-12, -3, (64, 64)) / 256).astype(np.float32)
        tol = 2e-6  # every |lo| is maximal here, so the dropped lo.lo term (2^-22 per product) is too;
        #             an inconsistent hi/lo pair would show as ~2^-11 (5e-4)
    else:  # unit rows / probabilities: what the loss feeds it; error must be fp32-grade
        A = rng.standard_normal((64, 256)).astype(np.float32)
        A /= np.linalg.norm(A, axis=1, keepdims=True)
        Bm = rng.standard_normal((64, 256)).astype(np.float32)
        Bm /= np.linalg.norm(Bm, axis=1, keepdims=True)
        G = rng.random((64, 64)).astype(np.float32) ** 4
        tol = 3e-7
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)  # noqa: E731
    X = torch.full((64, 64), float("nan"), device=dev)
    GE = torch.full((64, 256), float("nan"), device=dev)
    GC = torch.full((64, 256), float("nan"), device=dev)
    a, b, g = t(A), t(Bm), t(G)
    assert lib.ge2e_selftest_split_gemm(a.data_ptr(), b.data_ptr(), g.data_ptr(), X.data_ptr(), GE.data_ptr(),
                                        GC.data_ptr(), None) == 0
    torch.cuda.synchronize()
    A64, B64, G64 = A.astype(np.float64), Bm.astype(np.float64), G.astype(np.float64)
    for name, got, ref in (("X", X, A64 @ B64.T), ("GE", GE, G64 @ A64), ("GC", GC, G64.T @ B64)):
        err = np.abs(got.cpu().numpy() - ref).max()
        scale = max(1.0, np.abs(ref).max())
        scale = max(scale, float((np.abs(A64).max() * np.abs(B64).max()) * 64)) if kind == "ties" else scale
        assert err <= tol * scale, f"{name}: max err {err} (scale {scale})"


@pytest.mark.parametrize("kind", ["int", "unit"])
def test_rows16_chain(lib, kind):
    """16x16x32 tiles: X = CH.R^T, then X (from the accumulators) . CH, plain and through the quad transpose."""
    rng = np.random.default_rng(3)
    if kind == "int":
        CH = rng.integers(-4, 5, (64, 256)).astype(np.float32) / 64
        R = rng.integers(-4, 5, (16, 256)).astype(np.float32) / 64
        tol = 1e-7
    else:
        CH = rng.standard_normal((64, 256)).astype(np.float32)
        CH /= np.linalg.norm(CH, axis=1, keepdims=True)
        R = rng.standard_normal((16, 256)).astype(np.float32)
        R /= np.linalg.norm(R, axis=1, keepdims=True)
        tol = 3e-7
    dev = "cuda:0"
    d = lambda a: torch.as_tensor(a, device=dev)
    CHd, Rd = d(CH), d(R)
    XT = torch.full((64, 16), float("nan"), device=dev)
    GE = torch.full((16, 256), float("nan"), device=dev)
    GT = torch.full((16, 256), float("nan"), device=dev)
    assert lib.ge2e_selftest_rows16(CHd.data_ptr(), Rd.data_ptr(), XT.data_ptr(), GE.data_ptr(), GT.data_ptr(), None) == 0
    torch.cuda.synchronize()
    xt_ref = CH.astype(np.float64) @ R.astype(np.float64).T
    assert np.abs(XT.cpu().numpy() - xt_ref).max() <= tol
    ge_ref = XT.cpu().numpy().astype(np.float64).T @ CH.astype(np.float64)
    assert np.abs(GE.cpu().numpy() - ge_ref).max() <= 4 * tol
    assert np.array_equal(GT.cpu().numpy(), GE.cpu().numpy())


def test_team_formation_and_l2_handoff(lib):
    """Every CU gets one workgroup; eight per XCD form a team; payloads handed through L2 arrive intact."""
    dev = "cuda:0"
    grid = torch.cuda.get_device_properties(0).multi_processor_count
    payload, rounds = 512, 200                       # 8 KB per member per round
    nbytes = lib.ge2e_selftest_team_bytes(payload)
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
    out = torch.zeros(16, dtype=torch.int32, device=dev)
    rc = lib.ge2e_selftest_team(ws.data_ptr(), ws.numel(), grid, rounds, payload, out.data_ptr(), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    assert o[10] == 0, "a spin ran into its bound"
    assert o[2:10].sum() == grid
    assert o[0] == sum(c // 8 for c in o[2:10]) and o[0] >= 1
    assert o[1] == 0, f"{o[1]} stale or torn float4 of {o[0] * 8 * 8 * payload * rounds}"
