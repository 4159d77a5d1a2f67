"""GPU: the pipelined forward-only team kernel (csrc/ge2e_team_fwd.hip; dE = NULL through the C ABI: similarity + loss, the
launch of s4:61-110 and s5:42-44).  Two batches are in flight per workgroup and the members' scalars reach member 0 two
iterations late, so the cases walk every pipeline length (teams with 0, 1, 2, 3 and many batches), uneven members, every
supported D, both variants, and the (dw, db) outputs a caller may ask for without dE."""
import numpy as np
import pytest
import torch

from oracle import ge2e_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def GF():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import functional
    return functional


def run_fwd(GF, E, w, b, variant="softmax", impl="team", want_wb=True):
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    B, N, M, _ = e.shape
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731  (poisoned: a skipped batch must not pass)
    out = GF.LossOutputs(loss=nan(B), per=nan(B, N, M), dE=None, dw=nan(B) if want_wb else None, db=nan(B) if want_wb else None)
    GF.loss_fwd_bwd(e, torch.tensor(float(w), device=dev), torch.tensor(float(b), device=dev), variant=variant, impl=impl,
                    need_grad=False, out=out)
    torch.cuda.synchronize()
    return {k: (getattr(out, k).cpu().numpy() if getattr(out, k) is not None else None) for k in ("loss", "per", "dw", "db")}


def check_fwd(o, ref, what):
    nm = ref["per"].shape[-1] * ref["per"].shape[-2]
    assert np.isfinite(o["loss"]).all() and np.isfinite(o["per"]).all(), what
    assert np.allclose(o["loss"], ref["loss"], rtol=2e-5, atol=1e-6 + 2e-7 * nm), f"{what} loss {o['loss']} vs {ref['loss']}"
    assert np.allclose(o["per"], ref["per"], rtol=1e-4, atol=2e-5), f"{what} per"
    if o["dw"] is not None:
        assert np.allclose(o["dw"], ref["dw"], rtol=5e-5, atol=1e-5 + 1e-7 * nm), f"{what} dw {o['dw']} vs {ref['dw']}"
        assert np.allclose(o["db"], ref["db"], rtol=0, atol=1e-4 + 3e-7 * nm), f"{what} db"


@pytest.mark.parametrize("B", [1, 2, 3, 31, 32, 33, 64, 65, 97, 150])
def test_pipeline_lengths_metric_shape(GF, B):
    """32 teams on a whole MI355X: B = 1..150 gives teams with 0 to 5 batches, incl. the one- and two-batch tails."""
    N, M, D = 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=100 + B)
    assert GF.resolve_impl(B, N, M, D, "softmax", "team") == "team"
    ref = orc.closed_form(E, 10.0, -5.0)
    check_fwd(run_fwd(GF, E, 10.0, -5.0), ref, f"B={B}")


@pytest.mark.parametrize("shape", [(40, 23, 7, 128), (70, 9, 5, 64), (3, 32, 16, 64), (33, 64, 2, 192), (5, 17, 4, 256),
                                   (66, 16, 4, 64), (9, 40, 16, 128), (35, 57, 9, 256), (4, 64, 10, 64),
                                   (5, 64, 10, 200), (34, 24, 6, 80), (3, 64, 10, 4)])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_uneven_members_every_d(GF, shape, variant):
    B, N, M, D = shape
    assert GF.resolve_impl(B, N, M, D, variant, "team") == "team"
    E = orc.synth_embeddings(shape, "raw", seed=sum(shape))
    ref = orc.closed_form(E, 6.0, -1.5, variant=variant)
    check_fwd(run_fwd(GF, E, 6.0, -1.5, variant), ref, f"{shape} {variant}")


@pytest.mark.parametrize("variant", ["softmax", "contrast"])
@pytest.mark.parametrize("kind", ["clustered", "raw"])
def test_metric_shape_kinds_and_no_wb(GF, variant, kind):
    """Peaked softmax (clustered) and non-unit rows at the compile-time instantiation; with and without (dw, db)."""
    B, N, M, D = 37, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), kind, seed=7)
    ref = orc.closed_form(E, 10.0, -5.0, variant=variant)
    o1 = run_fwd(GF, E, 10.0, -5.0, variant, want_wb=True)
    o0 = run_fwd(GF, E, 10.0, -5.0, variant, want_wb=False)
    check_fwd(o1, ref, f"{kind} {variant}")
    # (without dw / db the softmax launch takes the fast form of S: another order of the same sums -- last-bit agreement)
    assert np.allclose(o0["loss"], o1["loss"], rtol=1e-5) and np.allclose(o0["per"], o1["per"], rtol=5e-5, atol=2e-6) and o0["dw"] is None


@pytest.mark.parametrize("wb", [(-3.0, 0.5), (0.0, 1.0), (60.0, -5.0), (1.0, 0.0)])
def test_w_sign_zero_and_large(GF, wb):
    """w < 0 (the reference does not clamp, s3:22), w = 0, and a w at which the unshifted exp of the reference overflows."""
    B, N, M, D = 5, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "clustered", seed=3)
    ref = orc.closed_form(E, wb[0], wb[1])
    check_fwd(run_fwd(GF, E, wb[0], wb[1]), ref, f"w,b={wb}")


def test_equals_the_training_launch_and_is_bitwise_repeatable(GF):
    """Forward-only loss / per-row losses against the fwd+bwd launch of the same batches (another kernel: last-bit agreement),
    and 20 launches of the forward kernel bit for bit (every cross-member sum runs in a fixed order)."""
    B, N, M, D = 300, 64, 10, 256
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    e = torch.nn.functional.normalize(torch.randn(B, N, M, D, device=dev, generator=g), dim=-1)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    og = GF.loss_fwd_bwd(e, w, b, impl="team", need_per=True)
    first = None
    for _ in range(20):
        of = GF.loss_fwd_bwd(e, w, b, impl="team", need_grad=False, need_per=True)
        torch.cuda.synchronize()
        cur = (of.loss.clone(), of.per.clone())
        if first is None:
            first = cur
        assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])
    assert torch.allclose(first[0], og.loss, rtol=1e-6) and torch.allclose(first[1], og.per, rtol=2e-6, atol=2e-6)


def test_workspace_is_left_clean_for_the_next_call(GF):
    """The forward kernel ends with two extra signals on the team counters; its last workgroup hands the control
    block back zeroed: a training launch and another forward launch on the SAME workspace stay correct, no fall-back."""
    B, N, M, D = 70, 64, 10, 256
    dev = torch.device("cuda:0")
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=9)
    e = torch.as_tensor(E, device=dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
    ref = orc.closed_form(E, 10.0, -5.0)
    for k in range(3):
        of = GF.loss_fwd_bwd(e, w, b, impl="team", need_grad=False, need_per=True, workspace=ws)
        og = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws)
        torch.cuda.synchronize()
        assert np.allclose(of.loss.cpu().numpy(), ref["loss"], rtol=2e-5), k
        assert np.allclose(og.loss.cpu().numpy(), ref["loss"], rtol=2e-5), k
        assert np.allclose(og.dw.cpu().numpy(), ref["dw"], rtol=1e-4), k
    assert GF.workspace_fallback_count(ws) == 0
