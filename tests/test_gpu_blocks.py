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
