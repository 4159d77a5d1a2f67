"""Speaker-sharded exact loss: the sum of the ranks' local losses is the single-device loss over all N
speakers and every rank's gradient is its slice of that loss's gradient.

CPU: world_size 2 over gloo with the ORACLE's helpers plugged in as ``ops`` (the product helpers have no CPU path) --
what is under test is the gather / reduce-scatter orchestration and the masking.
GPU: a 256-speaker batch cut into 2 and 8 shards, run one after the other on the one card through the HIP helpers,
against the single fused launch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ge2e_oracle as orc
from speaker_embedding_ge2e_loss_amd import sharded


class OracleOps:
    """The reference's static helpers (s3:34-38, s3:42-80, s3:115-127) as restated by the oracle."""

    @staticmethod
    def centroids(e):
        return orc.centroids(e)

    @staticmethod
    def cos_sim(e, c, eps=orc.SMALL_ERR):
        return orc.expand_form_cos_sim(e.contiguous(), c, eps)

    @staticmethod
    def calc_loss(sim, eps=orc.SMALL_ERR, variant="softmax"):
        per = orc._softmax_rows(sim, eps) if variant == "softmax" else orc._contrast_rows(sim)
        return per.sum(), per

    # The local-rows forms sharded_ge2e_loss asks for, composed from the whole-batch helpers above (which want as many
    # centroids as speakers, s3:77-78): the n local speakers in their global place in an (N, M, .) block of zeros, the
    # other speakers' rows masked out afterwards.  (Round 3's product composition; G-fold redundant row work.)
    @staticmethod
    def cos_sim_rows(e_local, C, first, eps=orc.SMALL_ERR):
        n, M, D = e_local.shape
        N = C.shape[0]
        e_pad = torch.cat([e_local.new_zeros(first, M, D), e_local, e_local.new_zeros(N - first - n, M, D)], dim=0)
        return OracleOps.cos_sim(e_pad, C, eps)[first:first + n]

    @staticmethod
    def calc_loss_rows(sim_rows, first, eps=orc.SMALL_ERR, variant="softmax"):
        n, M, N = sim_rows.shape
        s_pad = torch.cat([sim_rows.new_zeros(first, M, N), sim_rows, sim_rows.new_zeros(N - first - n, M, N)], dim=0)
        per = OracleOps.calc_loss(s_pad, eps, variant)[1][first:first + n]
        return per.sum(), per


def padded_sharded_loss(GF, e_local, w, b, rank, world, C, eps=1e-6, variant="softmax"):
    """Round 3's composition on the HIP helpers (timing baseline of test_gpu_local_rows_beat_the_padded_composition)."""
    n, M, D = e_local.shape
    N = n * world
    e_pad = torch.cat([e_local.new_zeros(rank * n, M, D), e_local, e_local.new_zeros(N - (rank + 1) * n, M, D)], dim=0).contiguous()
    sim = w * GF.cos_sim(e_pad, C, eps=eps) + b
    _, per = GF.calc_loss(sim, eps=eps, variant=variant)
    return per[rank * n:(rank + 1) * n].sum()


def _global_batch(N=6, M=4, D=16, seed=5):
    return torch.from_numpy(orc.synth_embeddings((N, M, D), "unit", seed=seed)).float()


def _reference(e, w0=10.0, b0=-5.0, variant="softmax"):
    e = e.clone().requires_grad_(True)
    w = torch.tensor(w0, requires_grad=True)
    b = torch.tensor(b0, requires_grad=True)
    loss = orc.expand_form_loss(e, w, b, variant=variant)[0]
    loss.backward()
    return float(loss.detach()), e.grad.numpy(), float(w.grad), float(b.grad)


def _worker(rank, world, port, variant, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    e = _global_batch()
    n = e.shape[0] // world
    mine = e[rank * n:(rank + 1) * n].clone().requires_grad_(True)
    w = torch.tensor(10.0, requires_grad=True)
    b = torch.tensor(-5.0, requires_grad=True)
    loss = sharded.sharded_ge2e_loss(mine, w, b, rank, world, ops=OracleOps, variant=variant)
    loss.backward()
    wb = torch.stack([w.grad, b.grad])
    dist.all_reduce(wb)                       # (w, b): partial per rank, summed like the trainer's bucket would
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    out.put((rank, float(tot), mine.grad.numpy(), wb.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_two_ranks_gloo_equal_the_unsharded_loss(variant):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, variant, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    loss, de, dw, db = _reference(_global_batch(), variant=variant)
    got = np.concatenate([r[2] for r in res], axis=0)
    for r in res:
        assert abs(r[1] - loss) <= 2e-5 * abs(loss)
        assert np.allclose(r[3], [dw, db], rtol=2e-5, atol=2e-6)
    assert np.allclose(got, de, rtol=2e-5, atol=1e-7), np.abs(got - de).max()


def test_single_process_gather_hook():
    """world 3 emulated in one process: ``gather`` = concatenation of every shard's centroids (autograd's cat backward
    is the reduce-scatter)."""
    e = _global_batch(N=6)
    shards = [e[2 * k:2 * k + 2].clone().requires_grad_(True) for k in range(3)]
    w = torch.tensor(10.0, requires_grad=True)
    b = torch.tensor(-5.0, requires_grad=True)
    cents = [OracleOps.centroids(s) for s in shards]
    total = sum(sharded.sharded_ge2e_loss(shards[k], w, b, k, 3, ops=OracleOps, gather=lambda c: torch.cat(cents, 0))
                for k in range(3))
    total.backward()
    loss, de, dw, db = _reference(e)
    assert abs(float(total) - loss) <= 2e-5 * abs(loss)
    got = np.concatenate([s.grad.numpy() for s in shards], 0)
    assert np.allclose(got, de, rtol=2e-5, atol=1e-7)
    assert np.allclose([float(w.grad), float(b.grad)], [dw, db], rtol=2e-5, atol=2e-6)   # db cancels to ~1e-4


def test_gather_size_is_checked():
    e = _global_batch(N=4)[:2].clone()
    w, b = torch.tensor(10.0), torch.tensor(-5.0)
    with pytest.raises(RuntimeError, match="centroids"):
        sharded.sharded_ge2e_loss(e, w, b, 0, 2, ops=OracleOps, gather=lambda c: c)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_gpu_shards_equal_the_single_launch(world, variant):
    from speaker_embedding_ge2e_loss_amd import functional as GF

    N, M, D = 256, 10, 256
    dev = torch.device("cuda:0")
    e = torch.from_numpy(orc.synth_embeddings((N, M, D), "unit", seed=77)).float().to(dev)
    w0, b0 = 10.0, -5.0

    ef = e.clone().requires_grad_(True)
    wf = torch.tensor(w0, device=dev, requires_grad=True)
    bf = torch.tensor(b0, device=dev, requires_grad=True)
    full = GF.ge2e_loss(ef, wf, bf, variant=variant)
    full.backward()

    n = N // world
    shards = [e[k * n:(k + 1) * n].clone().requires_grad_(True) for k in range(world)]
    w = torch.tensor(w0, device=dev, requires_grad=True)
    b = torch.tensor(b0, device=dev, requires_grad=True)
    cents = [GF.centroids(s) for s in shards]
    total = sum(sharded.sharded_ge2e_loss(shards[k], w, b, k, world, gather=lambda c: torch.cat(cents, 0),
                                          variant=variant) for k in range(world))
    total.backward()
    torch.cuda.synchronize()

    assert abs(float(total) - float(full)) <= 2e-5 * abs(float(full)), (float(total), float(full))
    got = torch.cat([s.grad for s in shards], 0)
    rel = float((got - ef.grad).norm() / ef.grad.norm())
    assert rel <= 2e-5, rel
    assert abs(float(w.grad) - float(wf.grad)) <= 2e-5 * abs(float(wf.grad)) + 1e-6
    assert abs(float(b.grad) - float(bf.grad)) <= 1e-4      # db cancels to ~1e-3: the suite's absolute bound


@pytest.mark.gpu
def test_gpu_eight_shards_end_to_end_time():
    """N = 256, M = 10, D = 256 (cfg4's batch) as ONE launch against the same loss cut into 8 speaker shards, forward +
    backward, end to end on one device (the shards one after the other: what eight ranks would each do once, plus the
    gather's cat and its backward).  A record, not a gate beyond sanity: printed, and quoted in DESIGN.md section 6."""
    import time
    from speaker_embedding_ge2e_loss_amd import functional as GF

    N, M, D, world = 256, 10, 256, 8
    n = N // world
    dev = torch.device("cuda:0")
    e = torch.from_numpy(orc.synth_embeddings((N, M, D), "unit", seed=78)).float().to(dev)

    def single():
        a = e.clone().requires_grad_(True)
        w = torch.tensor(10.0, device=dev, requires_grad=True)
        b = torch.tensor(-5.0, device=dev, requires_grad=True)
        GF.ge2e_loss(a, w, b).backward()
        return a.grad

    def shards8():
        sh = [e[k * n:(k + 1) * n].clone().requires_grad_(True) for k in range(world)]
        w = torch.tensor(10.0, device=dev, requires_grad=True)
        b = torch.tensor(-5.0, device=dev, requires_grad=True)
        cents = [GF.centroids(s_) for s_ in sh]
        tot = sum(sharded.sharded_ge2e_loss(sh[k], w, b, k, world, gather=lambda c: torch.cat(cents, 0)) for k in range(world))
        tot.backward()
        return torch.cat([s_.grad for s_ in sh], 0)

    t = {}
    for name, fn in (("single launch", single), ("eight shards", shards8)):
        for _ in range(3):
            g = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g = fn()
        torch.cuda.synchronize()
        t[name] = ((time.perf_counter() - t0) / 10, g)
    rel = float((t["eight shards"][1] - t["single launch"][1]).norm() / t["single launch"][1].norm())
    print(f"N=256 M=10 D=256 forward + backward, one device: single launch {t['single launch'][0] * 1e6:.0f} us, "
          f"8 shards one after the other {t['eight shards'][0] * 1e6:.0f} us ({t['eight shards'][0] / world * 1e6:.0f} us per shard); "
          f"dE rel-Frobenius difference {rel:.1e}")
    assert rel <= 2e-5


@pytest.mark.gpu
def test_gpu_single_rank_nccl_group():
    """The default gather (all-gather forward, reduce-scatter backward) over RCCL with one rank: world 1 is the plain loss."""
    from speaker_embedding_ge2e_loss_amd import functional as GF

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        dev = torch.device("cuda:0")
        e = torch.from_numpy(orc.synth_embeddings((16, 5, 64), "unit", seed=3)).float().to(dev)
        a = e.clone().requires_grad_(True)
        w = torch.tensor(10.0, device=dev, requires_grad=True)
        b = torch.tensor(-5.0, device=dev, requires_grad=True)
        loss = sharded.sharded_ge2e_loss(a, w, b, 0, 1)
        loss.backward()
        a2 = e.clone().requires_grad_(True)
        w2 = torch.tensor(10.0, device=dev, requires_grad=True)
        b2 = torch.tensor(-5.0, device=dev, requires_grad=True)
        ref = GF.ge2e_loss(a2, w2, b2)
        ref.backward()
        assert abs(float(loss) - float(ref)) <= 2e-5 * abs(float(ref))
        assert float((a.grad - a2.grad).norm() / a2.grad.norm()) <= 2e-5
    finally:
        if own:
            dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["softmax", "contrast"])
def test_gpu_local_rows_helpers_equal_the_whole_batch_ones(variant):
    """ge2e_cos_sim_rows / ge2e_calc_loss_rows (+ backward) on a slice of speakers against the whole-batch helpers: the
    slice's similarity rows, per-row losses, row gradients and the PARTIAL centroid gradient."""
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    N, M, D, j0, n = 24, 5, 96, 8, 6
    e = torch.from_numpy(orc.synth_embeddings((N, M, D), "raw", seed=9)).float().to(dev)
    c = torch.randn(N, D, device=dev)                       # the caller's centroids: NOT the means of e
    gsim = torch.randn(n, M, N, device=dev)
    # whole batch through the reference-shaped helpers; gradient only through the slice's rows of the output
    ef = e.clone().requires_grad_(True); cf = c.clone().requires_grad_(True)
    cos_f = GF.cos_sim(ef, cf)
    (cos_f[j0:j0 + n] * gsim).sum().backward()
    el = e[j0:j0 + n].clone().requires_grad_(True); cl = c.clone().requires_grad_(True)
    cos_l = GF.cos_sim_rows(el, cl, j0)
    (cos_l * gsim).sum().backward()
    torch.cuda.synchronize()
    assert torch.allclose(cos_l, cos_f[j0:j0 + n], rtol=1e-6, atol=1e-6)
    assert torch.allclose(el.grad, ef.grad[j0:j0 + n], rtol=1e-5, atol=1e-6)
    assert torch.allclose(cl.grad, cf.grad, rtol=1e-5, atol=1e-6)      # only the slice's rows carried a gradient
    assert float(ef.grad[:j0].abs().max()) == 0.0
    # calc_loss
    sf = (7.5 * cos_f.detach() - 2.0).requires_grad_(True)
    sl = (7.5 * cos_l.detach() - 2.0).requires_grad_(True)
    _, per_f = GF.calc_loss(sf, variant=variant)
    loss_l, per_l = GF.calc_loss_rows(sl, j0, variant=variant)
    per_f[j0:j0 + n].sum().backward()
    loss_l.backward()
    torch.cuda.synchronize()
    assert torch.allclose(per_l, per_f[j0:j0 + n], rtol=1e-6, atol=1e-6)
    assert torch.allclose(loss_l, per_f[j0:j0 + n].sum(), rtol=1e-6)
    assert torch.allclose(sl.grad, sf.grad[j0:j0 + n], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_gpu_local_rows_beat_the_padded_composition():
    """N = 256 in 8 shards on one device: the local-rows kernels against round 3's padded composition (the local rows in
    an (N,M,D) block of zeros): same loss and gradients, and the time of one shard's forward + backward."""
    import time
    from speaker_embedding_ge2e_loss_amd import functional as GF
    dev = torch.device("cuda:0")
    N, M, D, world = 256, 10, 256, 8
    n = N // world
    e = torch.from_numpy(orc.synth_embeddings((N, M, D), "unit", seed=5)).float().to(dev)
    cents = GF.centroids(e).detach()
    out = {}
    for padded in (False, True):
        def one():
            a = e[3 * n:4 * n].clone().requires_grad_(True)
            w = torch.tensor(10.0, device=dev, requires_grad=True)
            b = torch.tensor(-5.0, device=dev, requires_grad=True)
            cc = cents.clone().requires_grad_(True)
            loss = (padded_sharded_loss(GF, a, w, b, 3, world, cc) if padded
                    else sharded.sharded_ge2e_loss(a, w, b, 3, world, gather=lambda c_: cc))
            loss.backward()
            return loss.detach(), a.grad, cc.grad, w.grad
        for _ in range(3):
            r = one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            r = one()
        torch.cuda.synchronize()
        out[padded] = (r, (time.perf_counter() - t0) / 10)
    (l0, g0, c0, w0), t_rows = out[False]
    (l1, g1, c1, w1), t_pad = out[True]
    print(f"sharded loss, one of 8 shards of N=256: local rows {t_rows * 1e6:.0f} us, padded {t_pad * 1e6:.0f} us")
    assert torch.allclose(l0, l1, rtol=2e-6) and torch.allclose(g0, g1, rtol=1e-4, atol=1e-6)
    assert torch.allclose(c0, c1, rtol=1e-4, atol=1e-6) and torch.allclose(w0, w1, rtol=1e-5)
    assert t_rows < t_pad
