"""CPU, world_size 2 and 4 (unequal per-rank batch contents), gloo: the DP step's flat-bucket all-reduce gives the same update as one
process that averages the two ranks' gradients.  The loss here is the ORACLE's expand form (the
product loss has no CPU path); what is under test is the trainer's collective and step logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ge2e_oracle as orc
from speaker_embedding_ge2e_loss_amd.trainer import DPTrainer


class OracleLoss(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor(10.0))
        self.b = torch.nn.Parameter(torch.tensor(-5.0))

    def forward(self, emb):
        return orc.expand_form_loss(emb, self.w, self.b)[0]


class TinyEncoder(torch.nn.Module):
    """Stand-in with the reference encoder's contract (s2:27-35): (B,T,F) -> unit-norm (B,D)."""

    def __init__(self, feat=6, dim=8):
        super().__init__()
        self.lstm = torch.nn.LSTM(feat, 12, num_layers=1, batch_first=True)
        self.proj = torch.nn.Linear(12, dim)

    def forward(self, x):
        y, _ = self.lstm(x.float())
        y = self.proj(y[:, -1])
        return y / torch.norm(y, dim=1, keepdim=True)


def make(seed):
    torch.manual_seed(seed)
    return TinyEncoder(), OracleLoss()


def batch(rank):
    """Every rank its own batch -- and not the same kind of batch: rank 1's features are ten times larger, rank 2's utterances
    of a speaker are near copies of each other (the peaked-softmax regime: tiny gradients), rank 3's are plain noise again."""
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(3, 4, 5, 6, generator=g)
    if rank == 1:
        x = 10.0 * x
    if rank == 2:
        x = x[:, :1] + 0.05 * x
    return x


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model, loss = make(seed=rank)  # different inits: the trainer must broadcast rank 0's
    tr = DPTrainer(model, loss, lr=0.05, seed=7 + rank)
    losses = [float(tr.step(batch(rank))) for _ in range(2)]
    tr.halve_lr()
    flat = torch.cat([p.detach().reshape(-1) for p in tr._params])
    out.put((rank, flat.numpy(), losses, [g["lr"] for g in tr.optimizer.param_groups]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_dp_step_matches_single_process_mean_gradient(world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w0, l0, lr0), (_, w1, l1, lr1) = res[0], res[1]
    for r in res[1:]:
        assert np.array_equal(w0, r[1]), f"rank {r[0]} diverged"
        assert r[3] == [0.025, 0.05]          # halving touches group 0 only (s4:264)
    assert lr0 == [0.025, 0.05]

    # single process: same init (rank 0's), average the two ranks' gradients by hand
    torch.set_num_threads(1)
    model, loss = make(seed=0)
    params = list(model.parameters()) + list(loss.parameters())
    opt = torch.optim.SGD([{"params": model.parameters()}, {"params": loss.parameters()}], lr=0.05)
    for _ in range(2):
        grads = []
        for r in range(world):
            opt.zero_grad()
            x = batch(r)
            emb = model(x.reshape(12, 5, 6)).reshape(3, 4, -1).contiguous()  # perm/unperm is a no-op mathematically
            loss(emb).backward()
            grads.append([p.grad.clone() for p in params])
        for i, p in enumerate(params):
            p.grad = sum(g[i] for g in grads) / world
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3.0)
        torch.nn.utils.clip_grad_norm_(loss.parameters(), 1.0)
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
    assert np.allclose(w0, ref, rtol=1e-4, atol=1e-6), np.abs(w0 - ref).max()
    assert len({tuple(r[2]) for r in res}) == world  # each rank saw its own batch


def test_grads_are_views_of_one_bucket():
    model, loss = make(seed=3)
    tr = DPTrainer(model, loss)
    tr.step(batch(0))
    base = tr.flat_grad.untyped_storage().data_ptr()
    assert all(p.grad.untyped_storage().data_ptr() == base for p in tr._params)
    assert tr.flat_grad.numel() == sum(p.numel() for p in tr._params)
    assert float(tr.flat_grad.abs().sum()) > 0
