"""GPU parity: HIP path (through the C ABI) vs golden vectors and the fp64 oracle.

Tolerances (BASELINE.json north_star: loss rtol 1e-4; SURVEY 8d parity gate):
    loss  rtol 1e-4      (the exact-fp32 impls are held to 5e-6)
    dE    rel-Frobenius <= 1e-4 and max-abs <= 1e-4 * max|dE|   (exact-fp32: 5e-6)
    dw    rtol 1e-4 (+ tiny atol)   db  atol 1e-4 (cancellation residue, never rtol)
"""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden, rel_fro
from oracle import ge2e_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {  # impl -> (loss rtol, dE rel-fro / relative max-abs, dw rtol, cos atol)
    "generic": (5e-6, 1e-5, 2e-5, 2e-6),
    "fused_f32": (5e-6, 1e-5, 2e-5, 2e-6),
    "fused_split": (2e-5, 2e-5, 5e-5, 5e-6),
    "tiled": (2e-5, 2e-5, 5e-5, 5e-6),
    "team": (2e-5, 2e-5, 5e-5, 5e-6),
    "wave": (5e-6, 1e-5, 2e-5, 2e-6),
    "auto": (1e-4, 1e-4, 1e-4, 1e-5),
}


@pytest.fixture(scope="module")
def GF():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import functional
    return functional


def impls_for(GF, B, N, M, D, variant="softmax"):
    out = []
    for name in ("generic", "fused_f32", "fused_split", "tiled", "team", "wave"):
        try:
            GF.resolve_impl(B, N, M, D, variant, name)
            out.append(name)
        except RuntimeError:
            pass
    assert out, "no implementation accepts this shape"
    return out + ["auto"]


def run_hip(GF, E, w, b, variant="softmax", impl="auto", eps=1e-6):
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    wt = torch.tensor(float(w), device=dev)
    bt = torch.tensor(float(b), device=dev)
    # poison every output first: a kernel that skips rows must not pass on stale allocator memory
    shp = tuple(e.shape) if e.dim() == 4 else (1,) + tuple(e.shape)
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    out = GF.LossOutputs(loss=nan(shp[0]), per=nan(*shp[:3]), dE=nan(*shp), dw=nan(shp[0]), db=nan(shp[0]))
    o = GF.loss_fwd_bwd(e, wt, bt, variant=variant, impl=impl, eps=eps, out=out)
    torch.cuda.synchronize()
    squeeze = (lambda t: t[0]) if e.dim() == 3 else (lambda t: t)
    return {k: squeeze(getattr(o, k)).cpu().numpy() for k in ("loss", "per", "dE", "dw", "db")}


def check(o, ref, impl, what="", strict=False):
    """strict = the BASELINE configs and the reference's own fixtures: the stated gate with no per-row floors --
    db atol 1e-4 flat, dw rtol + 1e-5.  The floors that scale with the row count are for fuzz / ragged shapes only, where
    dw and db are sums of thousands of terms of both signs that cancel to ~0."""
    lt, gt, wt, _ = TOL[impl]
    loss_ref = np.asarray(ref["loss"], np.float64)
    # loss is a sum of NM terms of size ~|S|: fp32 noise floor scales with that, not with |loss|
    floor = 3e-7 * np.abs(np.asarray(ref["per"], np.float64)).sum(axis=(-1, -2))
    nm = int(np.prod(np.asarray(ref["per"]).shape[-2:]))   # rows per batch: an absolute fp32 floor per row (log(1 + tiny) at N = 1)
    assert np.all(np.abs(o["loss"] - loss_ref) <= lt * np.abs(loss_ref) + floor + 1e-6 + 2e-7 * nm), \
        f"{what} loss {o['loss']} vs {loss_ref}"
    assert np.allclose(o["per"], ref["per"], rtol=20 * lt, atol=2e-5), f"{what} per"
    scale = max(1.0, float(np.abs(ref["dE"]).max()))
    # (a gradient that vanishes -- one speaker: p_jj = 1 - O(eps) -- is held to an absolute bound instead)
    assert rel_fro(o["dE"], ref["dE"]) <= gt or np.abs(o["dE"] - ref["dE"]).max() <= 1e-8, \
        f"{what} dE rel-fro {rel_fro(o['dE'], ref['dE'])}"
    assert np.abs(o["dE"] - ref["dE"]).max() <= 4 * gt * scale, f"{what} dE max-abs"
    dw_ref = np.asarray(ref["dw"], np.float64)
    # dw = sum G (cos + eps) cancels when the rows' terms have both signs: an absolute floor per row beside the relative bound
    rowfloor = 0.0 if strict else 1.0
    assert np.all(np.abs(o["dw"] - dw_ref) <= wt * np.abs(dw_ref) + 1e-5 + 1e-7 * nm * rowfloor), f"{what} dw {o['dw']} vs {dw_ref}"
    # db cancels row by row: an fp32 floor per row -- for the non-BASELINE shapes only
    assert np.allclose(o["db"], ref["db"], rtol=0, atol=1e-4 + 3e-7 * nm * rowfloor), f"{what} db {o['db']} vs {ref['db']}"


@pytest.mark.parametrize("name", golden_names())
def test_golden_vectors(GF, name):
    """Every fixture produced by the imported reference (tests/golden/make_golden.py)."""
    g = load_golden(name)
    N, M, D = g["E"].shape
    ref = {"loss": g["loss64"], "per": g["per64"], "dE": g.get("dE64", g["dE"]), "dw": g["dw64"], "db": g["db64"]}
    for impl in impls_for(GF, 1, N, M, D):
        o = run_hip(GF, g["E"], g["w"], g["b"], impl=impl)
        if "degenerate" in name:
            # 1e8-scale gradients on the clamped rows: compare relative to the largest entry
            assert np.abs(o["dE"] - ref["dE"]).max() <= 1e-5 * np.abs(ref["dE"]).max()
            assert np.allclose(o["loss"], ref["loss"], rtol=1e-5)
            continue
        check(o, ref, impl, f"{name}/{impl}", strict=True)
        # and against the reference's own fp32 run, at the north-star tolerance
        assert np.allclose(o["loss"], g["loss"], rtol=1e-4, atol=1e-5)
        assert rel_fro(o["dE"], g["dE"]) <= 1e-4


CONFIGS = {  # BASELINE.json configs (B, N, M, D, variant)
    "cfg1": (1, 4, 5, 256, "softmax"),
    "cfg2": (3, 64, 10, 256, "softmax"),
    "cfg3": (3, 64, 10, 256, "contrast"),
    "cfg4": (2, 256, 10, 256, "softmax"),
}


@pytest.mark.parametrize("cfg", list(CONFIGS))
@pytest.mark.parametrize("kind", ["unit", "clustered"])
def test_baseline_configs_vs_oracle(GF, cfg, kind):
    B, N, M, D, variant = CONFIGS[cfg]
    E = orc.synth_embeddings((B, N, M, D), kind, seed=1234)
    ref = orc.closed_form(E, 10.0, -5.0, variant=variant)
    for impl in impls_for(GF, B, N, M, D, variant):
        check(run_hip(GF, E, 10.0, -5.0, variant, impl), ref, impl, f"{cfg}/{kind}/{impl}", strict=True)


def test_config5_large(GF):
    """N=1024, M=10, D=768: the reference cannot run it (2 x 32 GB expands); fp64 closed form is the oracle."""
    B, N, M, D = 1, 1024, 10, 768
    E = orc.synth_embeddings((B, N, M, D), "clustered", seed=5)
    ref = orc.closed_form(E, 10.0, -5.0)
    o = run_hip(GF, E, 10.0, -5.0, impl="auto")
    check(o, ref, "auto", "cfg5", strict=True)


def test_config5_benched_launch_sampled(GF):
    """cfg5 at the launch size bench.py times (bench.CONFIGS["cfg5"]["B"] batches, generated on the device): sampled
    batches against the fp64 closed form."""
    import bench
    B, N, M, D = bench.CONFIGS["cfg5"]["B"], 1024, 10, 768
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    e = torch.nn.functional.normalize(torch.randn(B, N, M, D, generator=g, device=dev), dim=-1)
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    out = GF.LossOutputs(loss=nan(B), per=nan(B, N, M), dE=nan(B, N, M, D), dw=nan(B), db=nan(B))
    o = GF.loss_fwd_bwd(e, torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev), impl="auto", out=out)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.loss).all()) and bool(torch.isfinite(o.dE).all())
    for i in (0, B // 2 - 1, B - 1):
        ref = orc.closed_form(e[i].cpu().numpy(), 10.0, -5.0)
        check({k: getattr(o, k)[i].cpu().numpy() for k in ("loss", "per", "dE", "dw", "db")}, ref, "auto", f"cfg5 B={B} batch {i}",
              strict=True)


@pytest.mark.parametrize("shape", [(2, 1, 2, 1), (1, 2, 2, 3), (3, 5, 3, 7), (2, 7, 4, 65), (1, 65, 2, 33),
                                   (2, 33, 9, 130), (1, 130, 3, 20), (4, 16, 16, 128), (1, 8, 40, 256),
                                   # one wave per batch: every instantiated M at its largest N, small and odd D, B > grid
                                   (5, 6, 2, 256), (3, 5, 3, 36), (9, 4, 4, 128), (2100, 4, 5, 256), (7, 3, 6, 64),
                                   (3, 3, 8, 252), (4, 2, 10, 4), (6, 2, 16, 256), (2, 1, 16, 8),
                                   # ... and its large (one wave per SIMD) instantiations at their largest N
                                   (3, 12, 2, 64), (2, 10, 3, 128), (5, 10, 4, 256), (3, 8, 5, 256), (2, 8, 6, 100),
                                   (3, 8, 8, 256), (2, 6, 10, 256), (4, 3, 16, 256), (2, 7, 5, 32),
                                   # D that is no multiple of 64 (any multiple of 4 up to 256): zero-padded inside the
                                   # load / store stages of the team kernel and of the one-workgroup-per-batch kernel
                                   (3, 64, 10, 200), (5, 40, 7, 80), (2, 20, 16, 4), (37, 64, 10, 132), (3, 9, 20, 100),
                                   # ... and of the tiled pipeline (N > 64 or D > 256): any multiple of 8 (ragged K-steps and tiles)
                                   (2, 130, 3, 72), (2, 256, 10, 200), (1, 300, 4, 264), (3, 100, 5, 40), (1, 520, 2, 776), (2, 64, 10, 328)])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_ragged_shapes(GF, shape, variant):
    """Odd sizes: N not a multiple of the wave, D not a multiple of 4, M = 2, N = 1."""
    E = orc.synth_embeddings(shape, "raw", seed=sum(shape))
    if variant == "contrast" and shape[1] == 1:
        pytest.skip("contrast needs >= 2 speakers")
    ref = orc.closed_form(E, 7.5, -2.0, variant=variant)
    impls = impls_for(GF, *shape, variant)
    if shape[3] % 4 == 0 and shape[3] <= 256 and 16 <= shape[1] <= 64 and shape[2] <= 16 and (shape[1] + 7) // 8 * shape[2] <= 80:
        assert "team" in impls and "fused_split" in impls, impls      # e.g. D = 200, D = 80: not the VALU kernel's business
    if shape[3] % 8 == 0 and (shape[1] > 64 or shape[3] > 256):
        assert "tiled" in impls and GF.resolve_impl(*shape, variant, "auto") == "tiled", impls
    for impl in impls:
        check(run_hip(GF, E, 7.5, -2.0, variant, impl), ref, impl, f"{shape}/{variant}/{impl}")


@pytest.mark.parametrize("w,b", [(-3.0, 0.5), (0.0, 1.0), (1.0, 0.0), (40.0, -20.0)])
def test_w_b_values_no_clamp(GF, w, b):
    """w is never clamped (s3:22 is a no-op): negative and zero w must follow the formula."""
    E = orc.synth_embeddings((2, 12, 5, 96), "unit", seed=3)
    ref = orc.closed_form(E, w, b)
    for impl in impls_for(GF, 2, 12, 5, 96):
        check(run_hip(GF, E, w, b, impl=impl), ref, impl, f"w={w}")


def test_large_w_stays_finite(GF):
    """The reference overflows to inf here (unstabilised exp); the kernel uses the
    shifted form, which is identical whenever the reference is finite."""
    E = orc.synth_embeddings((1, 8, 4, 64), "clustered", seed=1)
    ref = orc.closed_form(E, 200.0, -100.0, stable=True)
    o = run_hip(GF, E, 200.0, -100.0)
    assert np.isfinite(o["loss"]).all() and np.isfinite(o["dE"]).all()
    assert np.allclose(o["loss"], ref["loss"], rtol=1e-4, atol=1e-3)


def test_scale_invariance_property_full_size(GF):
    """Size-independent properties at BASELINE's full metric shape: cosines are invariant to
    a global positive rescale of E, so sum(E * dE) = 0 per batch (Euler), loss is unchanged
    under E -> 3E and dE shrinks by 3; permuting speakers permutes dE."""
    B, N, M, D = 8, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=77)
    for impl in impls_for(GF, B, N, M, D):
        o1 = run_hip(GF, E, 10.0, -5.0, impl=impl)
        euler = (E.astype(np.float64) * o1["dE"]).sum(axis=(1, 2, 3))
        assert np.abs(euler).max() < 1e-3 * np.abs(o1["dE"]).sum(axis=(1, 2, 3)).min()
        o3 = run_hip(GF, 3.0 * E, 10.0, -5.0, impl=impl)
        assert np.allclose(o3["loss"], o1["loss"], rtol=2e-6)
        assert rel_fro(3.0 * o3["dE"], o1["dE"]) < 1e-5
        perm = np.random.default_rng(0).permutation(N)
        op = run_hip(GF, np.ascontiguousarray(E[:, perm]), 10.0, -5.0, impl=impl)
        assert np.allclose(op["loss"], o1["loss"], rtol=2e-6)
        assert rel_fro(op["dE"], o1["dE"][:, perm]) < 1e-5


def test_batches_are_independent(GF):
    B, N, M, D = 5, 16, 6, 128
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=9)
    for impl in impls_for(GF, B, N, M, D):
        ob = run_hip(GF, E, 10.0, -5.0, impl=impl)
        for i in (0, B - 1):
            o1 = run_hip(GF, E[i], 10.0, -5.0, impl=impl)
            assert np.array_equal(o1["loss"], ob["loss"][i])
            assert np.array_equal(o1["dE"], ob["dE"][i])


def test_many_batches_grid_stride(GF):
    """More batches than workgroups in flight: every batch must still be written."""
    B, N, M, D = 1500, 6, 3, 32
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=4)
    ref = orc.closed_form(E, 10.0, -5.0)
    for impl in impls_for(GF, B, N, M, D):
        o = run_hip(GF, E, 10.0, -5.0, impl=impl)
        assert np.allclose(o["loss"], ref["loss"], rtol=1e-5)
        assert rel_fro(o["dE"], ref["dE"]) < 1e-5


@pytest.mark.parametrize("name", golden_names())
def test_static_helpers_on_every_fixture(GF, name):
    """get_centroids, get_utterance_centroids, get_cos_sim, calc_loss (s3:33-127) against the values the imported
    reference produced for every fixture -- not against each other."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    hp = HParams("cuda:0")
    g = load_golden(name)
    N, M, D = g["E"].shape
    e = torch.as_tensor(g["E"], device="cuda:0")
    cent = GE2ELoss.get_centroids(e)
    assert np.allclose(cent.cpu().numpy(), g["cent64"], rtol=1e-5, atol=1e-7)
    loo = GE2ELoss.get_utterance_centroids(e)
    assert np.allclose(loo.cpu().numpy(), g["loo"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(loo[N - 1, M - 1], GE2ELoss.get_centroid(e, N - 1, M - 1), atol=1e-5)   # the dead-code stub, eq. 8
    cos = GE2ELoss.get_cos_sim(e, cent, hp)
    assert cos.shape == (N, M, N)
    assert np.allclose(cos.cpu().numpy(), g["cos64"], atol=3e-6)
    w, b = float(g["w"]), float(g["b"])
    loss, per = GE2ELoss.calc_loss(w * cos + b, hp)                 # the 2-tuple of s3:127
    assert np.allclose(loss.item(), g["loss64"], rtol=2e-5, atol=1e-5)
    assert np.allclose(per.cpu().numpy(), g["per64"], rtol=2e-4, atol=2e-5)
    o = GF.loss_fwd_bwd(e, torch.tensor(w, device="cuda:0"), torch.tensor(b, device="cuda:0"), need_grad=False, need_per=True)
    assert o.dE is None and np.allclose(o.loss.item(), g["loss64"], rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize("name", golden_names())
def test_static_helpers_are_differentiable_like_the_reference(GF, name):
    """s3:19-30 rebuilt from the static helpers and backpropagated: the gradients the reference's autograd gave."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    hp = HParams("cuda:0")
    g = load_golden(name)
    e = torch.as_tensor(g["E"], device="cuda:0").requires_grad_(True)
    w = torch.tensor(float(g["w"]), device="cuda:0", requires_grad=True)
    b = torch.tensor(float(g["b"]), device="cuda:0", requires_grad=True)
    cent = GE2ELoss.get_centroids(e)
    cos = GE2ELoss.get_cos_sim(e, cent, hp)
    assert cos.requires_grad
    loss, per = GE2ELoss.calc_loss(w * cos + b, hp)
    loss.backward()
    dE64 = g["dE64"] if "dE64" in g else g["dE"]
    assert rel_fro(e.grad.cpu().numpy(), dE64) < 2e-5, rel_fro(e.grad.cpu().numpy(), dE64)
    assert np.allclose(w.grad.item(), g["dw64"], rtol=1e-4, atol=1e-5)
    assert np.allclose(b.grad.item(), g["db64"], atol=1e-4)
    # the second return value carries gradient too, and the leave-one-out helper is its own adjoint
    e2 = torch.as_tensor(g["E"], device="cuda:0").requires_grad_(True)
    _, per2 = GE2ELoss.calc_loss(float(g["w"]) * GE2ELoss.get_cos_sim(e2, GE2ELoss.get_centroids(e2), hp) + float(g["b"]), hp)
    per2.sum().backward()
    assert rel_fro(e2.grad.cpu().numpy(), dE64) < 2e-5
    x = torch.as_tensor(g["E"], device="cuda:0").requires_grad_(True)
    r = torch.as_tensor(np.random.default_rng(0).standard_normal(g["E"].shape).astype(np.float32), device="cuda:0")
    (GE2ELoss.get_utterance_centroids(x) * r).sum().backward()
    assert torch.allclose(x.grad, GE2ELoss.get_utterance_centroids(r), atol=1e-5)


def test_contrast_calc_loss_backward(GF):
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    hp = HParams("cuda:0")
    E = orc.synth_embeddings((6, 4, 32), "unit", seed=5)
    ref = orc.closed_form(E, 10.0, -5.0, variant="contrast")
    e = torch.as_tensor(E, device="cuda:0").requires_grad_(True)
    cos = GE2ELoss.get_cos_sim(e, GE2ELoss.get_centroids(e), hp)
    loss, _ = GE2ELoss.calc_loss(10.0 * cos - 5.0, hp, variant="contrast")
    loss.backward()
    assert np.allclose(loss.item(), ref["loss"], rtol=2e-5)
    assert rel_fro(e.grad.cpu().numpy(), ref["dE"]) < 2e-5


def test_module_autograd_matches_reference_semantics(GF):
    """Drop-in surface (s4:33-42, s4:196-203): parameters w,b; loss.backward(); clip; SGD."""
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    g = load_golden("g3_cfg2_n64_m10_d256")
    mod = GE2ELoss(HParams("cuda:0"))
    assert list(mod.state_dict().keys()) == ["w", "b"]
    e = torch.as_tensor(g["E"], device="cuda:0").requires_grad_(True)
    loss = mod(e)
    assert loss.dim() == 0
    (2.0 * loss).backward()  # arbitrary upstream gradient
    assert np.allclose(loss.item(), g["loss64"], rtol=1e-5)
    assert rel_fro(e.grad.cpu().numpy(), 2.0 * g["dE"]) < 1e-4
    assert np.allclose(mod.w.grad.item(), 2.0 * g["dw64"], rtol=1e-4)
    assert np.allclose(mod.b.grad.item(), 2.0 * g["db64"], atol=2e-4)
    opt = torch.optim.SGD([{"params": mod.parameters()}], lr=0.05)
    torch.nn.utils.clip_grad_norm_(mod.parameters(), 1.0)
    opt.step()
    assert float(mod.w) != 10.0
    # non-contiguous input raises like the reference's .view() (s3:49-52)
    with pytest.raises(RuntimeError):
        mod(e.detach().transpose(0, 1))
    # fp64 input is accepted (dtype-generic reference), computed in fp32
    l64 = mod(torch.as_tensor(g["E"], device="cuda:0").double())
    assert np.allclose(l64.item(), mod(torch.as_tensor(g["E"], device="cuda:0")).item())
    # ... and the loss comes back in the input's dtype, as from the reference (s3:19-30 computes in whatever it is given)
    assert l64.dtype == torch.float64
    e16 = torch.as_tensor(g["E"], device="cuda:0").half().requires_grad_(True)
    l16 = mod(e16)
    assert l16.dtype == torch.float16
    l16.backward()
    assert e16.grad is not None and e16.grad.dtype == torch.float16


def test_batched_module_backward(GF):
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    B, N, M, D = 3, 8, 4, 64
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=21)
    ref = orc.closed_form(E, 10.0, -5.0)
    mod = GE2ELoss(HParams("cuda:0"))
    e = torch.as_tensor(E, device="cuda:0").requires_grad_(True)
    losses = mod(e)
    assert losses.shape == (B,)
    losses.sum().backward()
    assert rel_fro(e.grad.cpu().numpy(), ref["dE"]) < 1e-5
    assert np.allclose(mod.w.grad.item(), ref["dw"].sum(), rtol=1e-4)


def test_get_cos_sim_uses_the_callers_centroids(GF):
    """s3:42-80: other-speaker columns come from the `centroids` ARGUMENT, the own-speaker column from the
    leave-one-out centroid of `embeddings`; the reference's eval (s5:36-44) calls it on graph-attached tensors."""
    import warnings
    from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
    hp = HParams("cuda:0")
    rng = np.random.default_rng(11)
    for (N, M, D) in ((4, 5, 256), (7, 3, 40), (64, 10, 256)):
        E = orc.synth_embeddings((N, M, D), "raw", seed=N + M)
        Cn = rng.standard_normal((N, D)).astype(np.float32)            # NOT the centroids of E
        ref = orc.expand_form_cos_sim(torch.as_tensor(E), torch.as_tensor(Cn)).numpy()
        e = torch.as_tensor(E, device="cuda:0").requires_grad_(True)   # like a model output in s5
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            cos = GE2ELoss.get_cos_sim(e, torch.as_tensor(Cn, device="cuda:0"), hp)
        assert cos.shape == (N, M, N) and cos.requires_grad and not rec
        assert np.allclose(cos.detach().cpu().numpy(), ref, atol=3e-6)
        # gradients with respect to BOTH arguments against autograd of the op-for-op restatement (fp64)
        c = torch.as_tensor(Cn, device="cuda:0").requires_grad_(True)
        wgt = torch.as_tensor(rng.standard_normal((N, M, N)).astype(np.float32), device="cuda:0")
        e.grad = None
        (GE2ELoss.get_cos_sim(e, c, hp) * wgt).sum().backward()
        e64 = torch.as_tensor(E, dtype=torch.float64).requires_grad_(True)
        c64 = torch.as_tensor(Cn, dtype=torch.float64).requires_grad_(True)
        (orc.expand_form_cos_sim(e64, c64) * wgt.cpu().double()).sum().backward()
        assert rel_fro(e.grad.cpu().numpy(), e64.grad.numpy()) < 1e-5
        assert rel_fro(c.grad.cpu().numpy(), c64.grad.numpy()) < 1e-5
        # with C = get_centroids(E) it is the fused kernels' similarity matrix
        own = GE2ELoss.get_cos_sim(e.detach(), GE2ELoss.get_centroids(e.detach()), hp)
        ref_own = orc.expand_form_cos_sim(torch.as_tensor(E), orc.centroids(torch.as_tensor(E))).numpy()
        assert np.allclose(own.cpu().numpy(), ref_own, atol=3e-6)
    with pytest.raises(RuntimeError):
        GE2ELoss.get_cos_sim(e.detach(), torch.zeros(3, 256, device="cuda:0"), hp)   # 3 centroids for 64 speakers


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 10, 256), (256, 10, 256), (17, 3, 64), (40, 5, 192), (16, 2, 128)])
def test_cos_sim_on_the_matrix_cores(GF, shape):
    """ge2e_cos_sim takes the tiled kernel's similarity contraction (split-fp16 MFMA) where the shape allows: against the
    fp32 VALU kernel fed with the same centroids, and against the oracle's expand form (s3:42-80)."""
    N, M, D = shape
    E = orc.synth_embeddings((2, N, M, D), "unit", seed=41 + N)
    e = torch.as_tensor(E, device="cuda:0")
    cos = GF.cos_sim(e)                                       # MFMA route (N >= 16, D % 64 == 0)
    ref_valu = GF.cos_sim(e, GF.centroids(e))                 # exact-fp32 VALU kernel, caller's centroids
    torch.cuda.synchronize()
    assert cos.shape == (2, N, M, N) and bool(torch.isfinite(cos).all())
    assert float((cos - ref_valu).abs().max()) < 5e-6
    for bi in range(2):
        t = torch.as_tensor(E[bi])
        ref = orc.expand_form_cos_sim(t, orc.centroids(t)).numpy()
        assert np.abs(cos[bi].cpu().numpy() - ref).max() < 5e-6


@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_tiled_more_than_64_utterances_per_speaker(GF, variant):
    """M = 70: a speaker's rows outnumber the lanes of the wave that sums their coefficients in k_spk (the lane loop wraps);
    against the fp64 closed form."""
    B, N, M, D = 3, 130, 70, 128
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=70)
    ref = orc.closed_form(E, 9.0, -4.0, variant=variant)
    check(run_hip(GF, E, 9.0, -4.0, variant, "tiled"), ref, "tiled", f"M=70 {variant}")


def test_cos_sim_through_the_walked_dma_tiles(GF):
    """ge2e_cos_sim at a shape whose similarity contraction takes the DMA-fed 256 x 256 tile with one workgroup per CU
    walking the tiles (24 x 11 x 2 = 528 tiles for 256 workgroups; ragged row, slot and K edges as in the loss test)."""
    B, N, M, D = 24, 288, 9, 320
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    e = torch.nn.functional.normalize(torch.randn(B, N, M, D, generator=g, device=dev), dim=-1)
    cos = GF.cos_sim(e)
    ref_valu = GF.cos_sim(e, GF.centroids(e))
    torch.cuda.synchronize()
    assert cos.shape == (B, N, M, N) and bool(torch.isfinite(cos).all())
    assert float((cos - ref_valu).abs().max()) < 5e-6
    t = e[B - 1].cpu()
    assert np.abs(cos[B - 1].cpu().numpy() - orc.expand_form_cos_sim(t, orc.centroids(t)).numpy()).max() < 5e-6


@pytest.mark.parametrize("shape", [(3, 256, 10, 256), (2, 200, 4, 64), (2, 129, 7, 128), (1, 256, 2, 64), (2, 255, 3, 192)])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_tiled_fused_similarity_and_row_pass(GF, shape, variant):
    """129 <= N <= 256 (one 256-slot tile holds a similarity row): TILED runs the similarity contraction and the row pass
    as ONE kernel (ge2e_tiled_simrows).  Uneven N (pad slots), row tiles that end inside a batch, both variants."""
    B, N, M, D = shape
    E = orc.synth_embeddings(shape, "raw", seed=N + M)
    ref = orc.closed_form(E, 7.5, -2.0, variant=variant)
    check(run_hip(GF, E, 7.5, -2.0, variant, "tiled"), ref, "tiled", f"{shape}/{variant}")
    # forward only (dE = NULL) through the same kernel
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    o = GF.loss_fwd_bwd(e, torch.tensor(7.5, device=dev), torch.tensor(-2.0, device=dev), variant=variant, impl="tiled",
                        need_grad=False, need_per=True)
    torch.cuda.synchronize()
    assert np.allclose(o.loss.cpu().numpy(), ref["loss"], rtol=2e-5)
    assert np.allclose(o.per.cpu().numpy(), ref["per"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_tiled_dma_contractions_at_ragged_tile_edges(GF, variant):
    """The 256 x 256 contractions whose operands arrive by buffer_load ... lds (K a multiple of 32, enough tiles to fill the
    chip): N = 288 (a slot tile of 32 valid columns), N M = 2592 (a row tile of 32 valid rows), D = 320 (a d tile of 64),
    B = 48 so that k_gc takes the big tile too and cuts its rows into pieces (192 tiles -> 4 pieces).  Sampled batches
    against the fp64 closed form."""
    B, N, M, D = 48, 288, 9, 320
    E = orc.synth_embeddings((B, N, M, D), "raw", seed=77)
    o = run_hip(GF, E, 7.5, -2.0, variant, "tiled")
    for i in (0, 23, 47):
        ref = orc.closed_form(E[i], 7.5, -2.0, variant=variant)
        check({k: v[i] for k, v in o.items()}, ref, "tiled", f"dma ragged {variant} batch {i}")


@pytest.mark.parametrize("shape", [(64, 10, 256), (4, 5, 256), (3, 64, 10, 256)])
def test_cpp_autograd_node_equals_the_python_one(GF, shape):
    """torch.ops.ge2e_amd.loss (libge2e_torch.so) against functional._GE2ELossFunction: the same two C-ABI calls, so loss
    and every gradient agree bit for bit -- scalar loss of one batch, and a (B,) loss vector with a vector of incoming grads."""
    if GF._cpp_loss_op() is None:
        pytest.fail("libge2e_torch.so is not built (speaker_embedding_ge2e_loss_amd.build.build_torch_ext)")
    dev = torch.device("cuda:0")
    E = orc.synth_embeddings(shape, "unit", seed=8)
    res = []
    for cpp in (True, False):
        GF.use_cpp_autograd(cpp)
        e = torch.as_tensor(E, device=dev).requires_grad_(True)
        w = torch.tensor(10.0, device=dev, requires_grad=True)
        b = torch.tensor(-5.0, device=dev, requires_grad=True)
        loss = GF.ge2e_loss(e, w, b)
        if loss.dim() == 0:
            loss.backward()
        else:
            loss.backward(torch.linspace(0.5, 1.5, loss.numel(), device=dev))
        torch.cuda.synchronize()
        res.append((loss.detach().clone(), e.grad.clone(), w.grad.clone(), b.grad.clone()))
    GF.use_cpp_autograd(True)
    for a, c in zip(*res):
        assert torch.equal(a, c)
    assert res[0][2].shape == torch.Size([]) and res[0][0].shape == (torch.Size([]) if len(shape) == 3 else torch.Size([shape[0]]))
    with torch.no_grad():                                   # forward only through the op (dE = NULL inside)
        e = torch.as_tensor(E, device=dev)
        l2 = GF.ge2e_loss(e, torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev))
    assert torch.allclose(l2, res[0][0], rtol=1e-6)
    ref = orc.closed_form(E if len(shape) == 3 else E[0], 10.0, -5.0)
    got = float(res[0][0]) if len(shape) == 3 else float(res[0][0][0])
    assert abs(got - ref["loss"]) <= 2e-5 * abs(ref["loss"])
