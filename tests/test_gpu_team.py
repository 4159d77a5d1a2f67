"""GPU: GE2E_IMPL_TEAM (eight workgroups of one XCD per batch) beyond the shared parity tests:
several batches per team (the rotated software pipeline), uneven speaker counts per member,
forward-only, and agreement with the one-workgroup-per-batch kernel."""
import numpy as np
import pytest
import torch

from conftest import rel_fro
from oracle import ge2e_oracle as orc
from test_gpu_parity import check, run_hip
import bench

BENCH_B = bench.CONFIGS["cfg2"]["B"]      # batches per launch of the metric config's bench line

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def GF():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from speaker_embedding_ge2e_loss_amd import functional
    return functional


TEAM_IMPLS = ("team",)


@pytest.mark.parametrize("impl", TEAM_IMPLS)
def test_many_batches_per_team_full_size(GF, impl):
    """B = 150 at the metric shape: 32 teams, 4-5 batches each -> every stage of the pipeline
    (prologue, steady state, drain) runs; all outputs written; matches fused_split and the oracle."""
    B, N, M, D = 150, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=21)
    ot = run_hip(GF, E, 10.0, -5.0, impl=impl)
    of = run_hip(GF, E, 10.0, -5.0, impl="fused_split")
    assert not np.isnan(ot["dE"]).any() and not np.isnan(ot["loss"]).any() and not np.isnan(ot["per"]).any()
    assert np.allclose(ot["loss"], of["loss"], rtol=2e-6)
    assert np.allclose(ot["dw"], of["dw"], rtol=2e-5)
    assert np.allclose(ot["db"], of["db"], atol=2e-5)
    for i in range(B):
        assert rel_fro(ot["dE"][i], of["dE"][i]) < 5e-6, i
    for i in (0, 31, 32, 149):
        ref = orc.closed_form(E[i], 10.0, -5.0)
        assert np.allclose(ot["loss"][i], ref["loss"], rtol=2e-5)
        assert rel_fro(ot["dE"][i], ref["dE"]) < 2e-5
        assert np.allclose(ot["per"][i], ref["per"], rtol=1e-4, atol=1e-4)
        check({k: v[i] for k, v in ot.items()}, ref, impl, f"batch {i}", strict=True)


@pytest.mark.parametrize("shape", [(40, 23, 7, 128), (70, 9, 5, 64), (3, 32, 16, 64), (33, 64, 2, 192), (5, 17, 4, 256)])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
@pytest.mark.parametrize("impl", TEAM_IMPLS)
def test_uneven_members(GF, shape, variant, impl):
    """N not a multiple of 8 (members with fewer or no speakers), M up to 16, every supported D."""
    B, N, M, D = shape
    assert GF.resolve_impl(B, N, M, D, variant, impl) == impl
    E = orc.synth_embeddings(shape, "raw", seed=sum(shape))
    ref = orc.closed_form(E, 6.0, -1.5, variant=variant)
    o = run_hip(GF, E, 6.0, -1.5, variant, impl)
    assert np.allclose(o["loss"], ref["loss"], rtol=2e-5), variant
    assert np.allclose(o["per"], ref["per"], rtol=1e-4, atol=1e-4)
    assert np.allclose(o["dw"], ref["dw"], rtol=1e-4, atol=1e-4)
    assert np.allclose(o["db"], ref["db"], atol=1e-4)
    for i in range(B):
        assert rel_fro(o["dE"][i], ref["dE"][i]) < 2e-5, (i, variant)


@pytest.mark.parametrize("impl", TEAM_IMPLS)
def test_forward_only(GF, impl):
    B, N, M, D = 37, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "clustered", seed=2)
    e = torch.as_tensor(E, device="cuda:0")
    w, b = torch.tensor(10.0, device="cuda:0"), torch.tensor(-5.0, device="cuda:0")
    o = GF.loss_fwd_bwd(e, w, b, impl=impl, need_grad=False, need_per=True)
    torch.cuda.synchronize()
    assert o.dE is None
    ref = orc.closed_form(E, 10.0, -5.0, want_grad=False)
    assert np.allclose(o.loss.cpu().numpy(), ref["loss"], rtol=2e-5)
    assert np.allclose(o.per.cpu().numpy(), ref["per"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(5, 64, 10, 64), (5, 40, 16, 64), (3, 64, 10, 128)])
@pytest.mark.parametrize("impl", ("team", "auto"))
def test_small_d_full_members(GF, shape, impl):
    """D = 64 / 128 with all 80 image rows in use (the shapes whose G images overflowed the previous team kernel's LDS)."""
    B, N, M, D = shape
    E = orc.synth_embeddings(shape, "unit", seed=sum(shape))
    ref = orc.closed_form(E, 10.0, -5.0)
    o = run_hip(GF, E, 10.0, -5.0, "softmax", impl)
    assert np.allclose(o["loss"], ref["loss"], rtol=2e-5)
    assert np.allclose(o["dw"], ref["dw"], rtol=1e-4, atol=1e-4)
    for i in range(B):
        assert rel_fro(o["dE"][i], ref["dE"][i]) < 2e-5, i
    if N == 64 and M == 10:      # the metric's (N, M): the flat gate of the BASELINE configs (db atol 1e-4, dw rtol + 1e-5)
        check(o, ref, impl, f"{shape}/{impl}", strict=True)


def test_fallback_when_no_team_forms(GF):
    """The team launch with its abort word raised: no team forms, every workgroup finds the word up at the end of the launch,
    stays, and together they redo the call with the one-workgroup-per-batch body (same launch) -- finite and correct, never
    NaN or stale memory."""
    from speaker_embedding_ge2e_loss_amd import _lib
    lib = _lib.load()
    for (B, N, M, D) in ((1, 64, 10, 256), (70, 64, 10, 256), (9, 23, 7, 128)):
        E = orc.synth_embeddings((B, N, M, D), "unit", seed=B + N)
        ref = orc.closed_form(E, 10.0, -5.0)
        dev = torch.device("cuda:0")
        e = torch.as_tensor(E, device=dev)
        w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
        nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
        loss, dE, dw, db = nan(B), nan(B, N, M, D), nan(B), nan(B)
        ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
        rc = lib.ge2e_selftest_team_fallback(e.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0,
                                             loss.data_ptr(), None, dE.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                             ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.allclose(loss.cpu().numpy(), ref["loss"], rtol=2e-5)
        assert np.allclose(dw.cpu().numpy(), ref["dw"], rtol=1e-4, atol=1e-4)
        for i in range(B):
            assert rel_fro(dE[i].cpu().numpy(), ref["dE"][i]) < 2e-5, (B, i)


def test_abort_raised_in_the_middle_of_the_grid(GF):
    """ONE workgroup raises the abort word when it reaches the end of the launch: the workgroups that finished before it have
    left, the later ones stay -- the redo's size is decided once (TeamCtl.go, by the last finisher), so the stayers' strides
    tile the batches whatever each of them saw.  Outputs are NaN-poisoned: a batch nobody redid, or wrote half, shows.
    Afterwards the control block must be clean: the next ordinary call forms its teams and counts no further fall-back."""
    from speaker_embedding_ge2e_loss_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    for (B, N, M, D), fwd_only in (((70, 64, 10, 256), False), ((300, 64, 10, 256), False), ((9, 23, 7, 128), False),
                                   ((70, 64, 10, 256), True), ((33, 40, 6, 192), True)):
        E = orc.synth_embeddings((B, N, M, D), "unit", seed=B + N)
        ref = orc.closed_form(E, 10.0, -5.0)
        e = torch.as_tensor(E, device=dev)
        w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
        nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
        ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
        for rep in range(3):
            loss, dE, dw, db = nan(B), nan(B, N, M, D), nan(B), nan(B)
            rc = lib.ge2e_selftest_team_abort_midgrid(e.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0,
                                                      loss.data_ptr(), None, None if fwd_only else dE.data_ptr(),
                                                      None if fwd_only else dw.data_ptr(), None if fwd_only else db.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            assert np.allclose(loss.cpu().numpy(), ref["loss"], rtol=2e-5), (B, N, rep)
            if not fwd_only:
                assert np.allclose(dw.cpu().numpy(), ref["dw"], rtol=1e-4, atol=1e-4)
                for i in range(B):
                    assert rel_fro(dE[i].cpu().numpy(), ref["dE"][i]) < 2e-5, (B, i, rep)
            assert GF.workspace_fallback_count(ws) == rep + 1          # counted once per call, by a single writer
        # the block was left clean: an ordinary call on the same workspace runs with teams
        o = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws, need_grad=not fwd_only)
        torch.cuda.synchronize()
        assert np.allclose(o.loss.cpu().numpy(), ref["loss"], rtol=2e-5)
        assert GF.workspace_fallback_count(ws) == 3


def test_team_beside_a_busy_stream(GF):
    """A long-running kernel on a second stream holds CUs while the team launch goes out (its workgroups wait for each
    other): the result must still be finite and correct, through the team kernel or through its fall-back."""
    B, N, M, D = 1, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=3)
    ref = orc.closed_form(E, 10.0, -5.0)
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(6):
                a = torch.tanh(a @ a) * 1e-2         # tens of milliseconds of whole-chip work on the side stream
        o = GF.loss_fwd_bwd(e, w, b, impl="auto")
        torch.cuda.synchronize()
        assert torch.isfinite(o.loss).all() and torch.isfinite(o.dE).all()
        assert np.allclose(o.loss.cpu().numpy(), ref["loss"], rtol=2e-5)
        assert rel_fro(o.dE[0].cpu().numpy(), ref["dE"][0]) < 2e-5


def test_auto_uses_team_for_few_batches(GF):
    assert GF.resolve_impl(1, 64, 10, 256, "softmax", "auto") == "team"
    assert GF.resolve_impl(4096, 64, 10, 256, "softmax", "auto") == "team"            # ties in time, far less traffic
    assert GF.resolve_impl(1024, 64, 10, 256, "softmax", "auto") == "team"
    assert GF.resolve_impl(256, 64, 10, 256, "softmax", "auto") == "fused_split"   # one workgroup per CU, exactly one round
    assert GF.resolve_impl(1, 4, 5, 256, "softmax", "auto") == "wave"            # a few dozen rows: one wave per batch
    assert GF.resolve_impl(1, 8, 5, 256, "softmax", "auto") == "fused_split"     # 40 rows on one wave: only for many batches
    assert GF.resolve_impl(1000, 8, 5, 256, "softmax", "auto") == "wave"
    assert GF.resolve_impl(1, 12, 6, 256, "softmax", "auto") == "fused_split"    # too few speakers for eight members, too many rows for a wave
    with pytest.raises(RuntimeError):
        GF.resolve_impl(1, 64, 20, 256, "softmax", "team")                        # M > 16


@pytest.mark.parametrize("impl", TEAM_IMPLS)
def test_hand_off_protocol_stress(GF, impl):
    """400 launches of a multi-batch-per-team problem: every launch bitwise equal to the first
    (a stale or torn read in the L2 hand-offs would show as a different bit pattern somewhere)."""
    B, N, M, D = 72, 64, 10, 256
    E = orc.synth_embeddings((B, N, M, D), "unit", seed=99)
    dev = torch.device("cuda:0")
    e = torch.as_tensor(E, device=dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", impl), dev)
    first = None
    for it in range(400):
        out = GF.LossOutputs(loss=torch.full((B,), float("nan"), device=dev), per=None,
                             dE=torch.full((B, N, M, D), float("nan"), device=dev),
                             dw=torch.full((B,), float("nan"), device=dev), db=torch.full((B,), float("nan"), device=dev))
        GF.loss_fwd_bwd(e, w, b, impl=impl, out=out, workspace=ws)
        if first is None:
            torch.cuda.synchronize()
            first = (out.loss.clone(), out.dE.clone(), out.dw.clone(), out.db.clone())
            ref = orc.closed_form(E[7], 10.0, -5.0)
            assert rel_fro(first[1][7].cpu().numpy(), ref["dE"]) < 2e-5
        elif it % 25 == 0 or it == 399:
            torch.cuda.synchronize()
            assert torch.equal(out.loss, first[0]) and torch.equal(out.dE, first[1]), it
            assert torch.equal(out.dw, first[2]) and torch.equal(out.db, first[3]), it


def _device_batches(B, N, M, D, seed):
    """normalize(randn) batches generated ON the device (a 2.7 GB stack is slow to make on the host)."""
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    e = torch.randn(B, N, M, D, generator=g, device="cuda:0", dtype=torch.float32)
    return torch.nn.functional.normalize(e, dim=-1)


def _check_sampled(e, o, picks, w=10.0, b=-5.0):
    for i in picks:
        ref = orc.closed_form(e[i].cpu().numpy(), w, b)
        assert np.allclose(float(o.loss[i]), ref["loss"], rtol=2e-5), i
        assert rel_fro(o.dE[i].cpu().numpy(), ref["dE"]) < 2e-5, i
        assert np.allclose(float(o.dw[i]), ref["dw"], rtol=1e-4, atol=1e-4), i
        assert np.allclose(float(o.db[i]), ref["db"], atol=1e-4), i


@pytest.mark.parametrize("impl", ("team", "auto"))
def test_benched_launch_size(GF, impl):
    """The launch bench.py times: bench.CONFIGS["cfg2"]["B"] batches at the metric shape (64 pipelined batches per team).
    Every output finite, sampled batches (first, last, team boundaries, the middle, a few random ones) against the fp64
    oracle, and the whole launch against the one-workgroup-per-batch kernel."""
    B, N, M, D = BENCH_B, 64, 10, 256
    e = _device_batches(B, N, M, D, 77)
    w, b = torch.tensor(10.0, device="cuda:0"), torch.tensor(-5.0, device="cuda:0")
    nan = lambda *s: torch.full(s, float("nan"), device="cuda:0")  # noqa: E731
    out = GF.LossOutputs(loss=nan(B), per=None, dE=nan(B, N, M, D), dw=nan(B), db=nan(B))
    o = GF.loss_fwd_bwd(e, w, b, impl=impl, out=out)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.loss).all()) and bool(torch.isfinite(o.dw).all()) and bool(torch.isfinite(o.db).all())
    assert bool(torch.isfinite(o.dE).all())
    rng = np.random.default_rng(5)
    _check_sampled(e, o, [0, 31, 32, 255, 256, B // 2 - 1, B // 2, B - 32, B - 1] + [int(x) for x in rng.integers(0, B, 4)])
    of = GF.loss_fwd_bwd(e, w, b, impl="fused_split")
    torch.cuda.synchronize()
    assert torch.allclose(o.loss, of.loss, rtol=2e-6)
    assert torch.allclose(o.dw, of.dw, rtol=2e-5, atol=1e-6) and torch.allclose(o.db, of.db, atol=2e-5)
    worst = 0.0
    for c0 in range(0, B, 2048):     # in pieces: the difference of two 10 GB tensors need not exist at once
        num = (o.dE[c0:c0 + 2048] - of.dE[c0:c0 + 2048]).flatten(1).norm(dim=1)
        den = of.dE[c0:c0 + 2048].flatten(1).norm(dim=1)
        worst = max(worst, float((num / den).max()))
    assert worst < 5e-6


@pytest.mark.parametrize("impl", ("team", "auto"))
def test_forward_only_at_the_benched_launch_size(GF, impl):
    """bench.py's forward_only leg: the benched batch count at the metric shape with dE = NULL (similarity + loss, s4:61-110 /
    s5:42-44).  loss and per-row losses: all finite, equal to the fwd+bwd launch's, sampled batches against the oracle."""
    B, N, M, D = BENCH_B, 64, 10, 256
    e = _device_batches(B, N, M, D, 78)
    w, b = torch.tensor(10.0, device="cuda:0"), torch.tensor(-5.0, device="cuda:0")
    nan = lambda *s: torch.full(s, float("nan"), device="cuda:0")  # noqa: E731
    of = GF.loss_fwd_bwd(e, w, b, impl=impl, need_grad=False,
                         out=GF.LossOutputs(loss=nan(B), per=nan(B, N, M), dE=None, dw=None, db=None))
    og = GF.loss_fwd_bwd(e, w, b, impl=impl, need_per=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(of.loss).all()) and bool(torch.isfinite(of.per).all())
    # the forward-only launch is an instantiation of its own (no gradient code compiled in): same arithmetic, but the
    # compiler may contract an fma differently -- last-bit agreement, not bitwise
    assert torch.allclose(of.loss, og.loss, rtol=1e-6) and torch.allclose(of.per, og.per, rtol=2e-6, atol=2e-6)
    for i in (0, 31, 32, B // 2, B - 1):
        ref = orc.closed_form(e[i].cpu().numpy(), 10.0, -5.0, want_grad=False)
        assert abs(float(of.loss[i]) - ref["loss"]) <= 2e-5 * abs(ref["loss"]), i
        assert np.abs(of.per[i].cpu().numpy() - ref["per"]).max() <= 2e-5 * max(1.0, np.abs(ref["per"]).max()), i


@pytest.mark.parametrize("shape,per_team", [((16, 4, 64), 1100), ((64, 10, 256), 1000)])
def test_thousand_batches_through_one_team(GF, shape, per_team):
    """The launch capped to 64 workgroups = one team per XCD (ge2e_selftest_team_grid): every team works through a
    thousand batches, its hand-off counters reach 8000 (c1, c2) and N x 1000 (c3) -- far beyond the 128 batches per
    team of the benched launch.  Results against the one-workgroup-per-batch kernel (all) and the oracle (sampled)."""
    from speaker_embedding_ge2e_loss_amd import _lib
    lib = _lib.load()
    N, M, D = shape
    B = 8 * per_team
    dev = torch.device("cuda:0")
    e = _device_batches(B, N, M, D, 91)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    nan = lambda *s: torch.full(s, float("nan"), device=dev)  # noqa: E731
    o = GF.LossOutputs(loss=nan(B), per=None, dE=nan(B, N, M, D), dw=nan(B), db=nan(B))
    ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", "team"), dev)
    rc = lib.ge2e_selftest_team_grid(e.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0,
                                     o.loss.data_ptr(), None, o.dE.data_ptr(), o.dw.data_ptr(), o.db.data_ptr(),
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream, 64)
    assert rc == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.loss).all()) and bool(torch.isfinite(o.dE).all())
    _check_sampled(e, o, [0, 7, 8, B // 2, B - 9, B - 1])
    of = GF.loss_fwd_bwd(e, w, b, impl="fused_split")
    torch.cuda.synchronize()
    assert torch.allclose(o.loss, of.loss, rtol=2e-6)
    num = (o.dE - of.dE).flatten(1).norm(dim=1)
    den = of.dE.flatten(1).norm(dim=1)
    assert float((num / den).max()) < 5e-6


def test_control_block_cleans_itself(GF):
    """No zeroing launch in front of the team kernel and none behind it.  A call leaves the control block clean (the last
    workgroup of the launch rewrites it), a workspace that was never initialised -- or that another implementation has written
    over -- makes the call fall back ONCE and come out clean, and `workspace_fallback_count` says which happened."""
    dev = torch.device("cuda:0")
    B, N, M, D = 5, 64, 10, 256
    e = _device_batches(B, N, M, D, 13)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    nbytes = GF.workspace_bytes(B, N, M, D, "softmax", "team")
    ws_ok = GF.alloc_workspace(nbytes, dev)                                    # initialised: team from the first call on
    ref = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws_ok)
    torch.cuda.synchronize()
    assert GF.workspace_fallback_count(ws_ok) == 0
    ws_raw = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device=dev)        # garbage where the control block goes
    o1 = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws_raw)               # -> in-call fall-back, block cleaned up
    torch.cuda.synchronize()
    assert GF.workspace_fallback_count(ws_raw) == 1
    assert torch.allclose(o1.loss, ref.loss, rtol=2e-6) and float(((o1.dE - ref.dE).norm() / ref.dE.norm())) < 5e-6
    for _ in range(3):                                                          # ... and team ever after, bit for bit
        o2 = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws_raw)
        torch.cuda.synchronize()
        assert torch.equal(o2.loss, ref.loss) and torch.equal(o2.dE, ref.dE) and torch.equal(o2.dw, ref.dw)
    assert GF.workspace_fallback_count(ws_raw) == 1
    # another implementation writes over the head of the same workspace in between
    big = max(nbytes, GF.workspace_bytes(B, N, M, D, "softmax", "tiled"))
    ws = GF.alloc_workspace(big, dev)
    GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws)
    GF.loss_fwd_bwd(e, w, b, impl="tiled", workspace=ws)
    o3 = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws)
    o4 = GF.loss_fwd_bwd(e, w, b, impl="team", workspace=ws)
    torch.cuda.synchronize()
    assert torch.allclose(o3.loss, ref.loss, rtol=2e-6) and torch.equal(o4.dE, ref.dE)
    assert float(((o3.dE - ref.dE).norm() / ref.dE.norm())) < 5e-6
