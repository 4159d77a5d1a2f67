"""Speaker-sharded EXACT GE2E loss: every rank holds the rows of N / G speakers and the loss is the
single-device loss over all N speakers -- same value, same gradient -- not G independent losses over fewer negatives.

    local centroids  (get_centroids, s3:34-38)                          [n, D]
    ALL-GATHER       -> every rank has all N centroids                  N D 4 bytes (256 KB at N = 256, D = 256)
    similarities of the LOCAL rows against all N centroids, leave-one-out centroid on the own column (get_cos_sim, s3:42-80)
    S = w cos + b, per-row loss (calc_loss, s3:115-127), summed over the local rows
    backward: ... -> dC [N, D] from the local rows -> REDUCE-SCATTER (sum) -> each rank's own n rows of it -> centroids_bwd

The two collectives are the only exchange; the sum of the ranks' local losses IS GE2ELoss(all N speakers) and every
rank's gradient is the matching slice of its gradient (tests/test_sharded.py: a 256-speaker batch cut into 2 and 8 shards
against the single launch).  (w, b) receive each rank's partial gradient: sum them over the ranks (an all-reduce, e.g. the
trainer's bucket) for the full one.

Built from the differentiable static helpers in their LOCAL-ROWS form (ge2e_cos_sim_rows / ge2e_calc_loss_rows and their
backward kernels, include/ge2e_hip.h): a rank sweeps its own n M rows against the N gathered centroids, the own column of
local speaker jl being rank n + jl.  (The round-3 composition -- the local rows embedded in an (N, M, D) block of zeros so
that the whole-batch helpers could be used -- lives on only in tests/test_sharded.py, as the CPU oracle's form of the two
local-rows helpers and as the baseline of a timing comparison.)  The data-parallel path the north star prescribes (whole
batches per rank, trainer.py) is unaffected.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


class _AllGatherRows(torch.autograd.Function):
    """rows [n, D] on every rank -> [G n, D] in rank order; backward = reduce-scatter (sum) of the gradient."""

    @staticmethod
    def forward(ctx, rows, group):
        ctx.group = group
        world = dist.get_world_size(group)
        parts = [torch.empty_like(rows) for _ in range(world)]
        dist.all_gather(parts, rows.contiguous(), group=group)
        return torch.cat(parts, dim=0)

    @staticmethod
    def backward(ctx, grad):
        world, rank = dist.get_world_size(ctx.group), dist.get_rank(ctx.group)
        grad = grad.contiguous()
        n = grad.shape[0] // world
        # The collective is chosen ONCE from the group's backend, never from an exception: a rank that caught a genuine
        # RCCL failure and switched to another collective would issue a different one from its peers (a hang or a
        # mismatched collective instead of the error).  gloo has no reduce_scatter: all-reduce, keep the own rows.
        if dist.get_backend(ctx.group) == "gloo":
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=ctx.group)
            return grad[rank * n:(rank + 1) * n].clone(), None
        out = torch.empty_like(grad[:n])
        dist.reduce_scatter_tensor(out, grad, op=dist.ReduceOp.SUM, group=ctx.group)
        return out, None


def all_gather_rows(rows: torch.Tensor, group: Optional["dist.ProcessGroup"] = None) -> torch.Tensor:
    return _AllGatherRows.apply(rows, group)


def sharded_ge2e_loss(e_local: torch.Tensor, w: torch.Tensor, b: torch.Tensor, rank: int, world: int, *,
                      gather: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, ops=None,
                      eps: float = 1e-6, variant: str = "softmax") -> torch.Tensor:
    """This rank's share of the exact N-speaker loss: ``e_local`` (n, M, D) are the rows of speakers
    ``rank * n .. rank * n + n - 1`` of the global batch (N = world * n).  Returns the sum of the per-row losses of the
    LOCAL rows (a scalar that backpropagates into ``e_local``, ``w``, ``b`` and -- through the gather -- into the other
    ranks' rows); the global loss is the sum of the returns over the ranks.

    ``gather``: [n, D] -> [N, D] in rank order, differentiable (default: RCCL / gloo all-gather whose backward is a
    reduce-scatter).  ``ops``: where ``centroids``, ``cos_sim_rows(e, C, first_speaker, eps=)`` and
    ``calc_loss_rows(sim, first_speaker, eps=, variant=)`` come from (default: this package's HIP ones, ``functional``; the
    multi-process CPU test plugs the oracle's in -- what it tests is the collective orchestration)."""
    if ops is None:
        from . import functional as ops
    if gather is None:
        gather = all_gather_rows
    n, M, D = e_local.shape
    N = n * world
    c_local = ops.centroids(e_local)                              # (n, D)
    C = gather(c_local)                                           # (N, D): everybody's centroids
    if C.shape[0] != N:
        raise RuntimeError(f"gather returned {C.shape[0]} centroids for {N} speakers")
    # n M rows against N centroids, own column rank n + jl; the gradient w.r.t. C is this shard's partial one (the gather's
    # backward sums the shards)
    cos = ops.cos_sim_rows(e_local, C, rank * n, eps=eps)         # (n, M, N)
    sim = w * cos + b                                             # s3:27
    loss, _ = ops.calc_loss_rows(sim, rank * n, eps=eps, variant=variant)
    return loss
