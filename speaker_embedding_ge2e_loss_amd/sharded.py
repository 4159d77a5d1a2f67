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

Built from the differentiable static helpers in their LOCAL-ROWS form (round 4: ge2e_cos_sim_rows / ge2e_calc_loss_rows and
their backward kernels, include/ge2e_hip.h): a rank sweeps its own n M rows against the N gathered centroids, the own
column of local speaker jl being rank n + jl.  (Round 3 embedded the local rows in an (N, M, D) block of zeros because the
whole-batch helpers want as many centroids as speakers, s3:77-78: G-fold redundant row work; that composition is kept for
`ops` without the local-rows forms -- the CPU oracle of the gloo test -- and as `padded=True` for comparison.)  The
data-parallel path the north star prescribes (whole batches per rank, trainer.py) is unaffected.
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
                      eps: float = 1e-6, variant: str = "softmax", padded: bool = False) -> torch.Tensor:
    """This rank's share of the exact N-speaker loss: ``e_local`` (n, M, D) are the rows of speakers
    ``rank * n .. rank * n + n - 1`` of the global batch (N = world * n).  Returns the sum of the per-row losses of the
    LOCAL rows (a scalar that backpropagates into ``e_local``, ``w``, ``b`` and -- through the gather -- into the other
    ranks' rows); the global loss is the sum of the returns over the ranks.

    ``gather``: [n, D] -> [N, D] in rank order, differentiable (default: RCCL / gloo all-gather whose backward is a
    reduce-scatter).  ``ops``: an object with the reference's static helpers ``centroids``, ``cos_sim(e, c, eps=)``,
    ``calc_loss(sim, eps=, variant=)`` (default: this package's HIP ones, ``functional``, which also have the local-rows
    forms ``cos_sim_rows`` / ``calc_loss_rows``: n M rows against N centroids instead of an (N, M, D) block of zeros;
    ``padded=True`` forces the padded composition, for comparison)."""
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
    if not padded and hasattr(ops, "cos_sim_rows") and hasattr(ops, "calc_loss_rows"):
        # local-rows kernels (round 4): n M rows against N centroids, own column rank n + jl -- no padding, no G-fold
        # redundant row work, and the gradient w.r.t. C is this shard's partial one (the gather's backward sums the shards)
        cos = ops.cos_sim_rows(e_local, C, rank * n, eps=eps)         # (n, M, N)
        sim = w * cos + b                                             # s3:27
        loss, _ = ops.calc_loss_rows(sim, rank * n, eps=eps, variant=variant)
        return loss
    # compatibility composition for `ops` that only have the reference's whole-batch helpers (the CPU oracle in the gloo
    # test): the local rows in their global place; the other speakers' rows are zeros (cosine 0, loss masked out below)
    pad_before = e_local.new_zeros(rank * n, M, D)
    pad_after = e_local.new_zeros(N - (rank + 1) * n, M, D)
    e_pad = torch.cat([pad_before, e_local, pad_after], dim=0).contiguous()
    cos = ops.cos_sim(e_pad, C, eps=eps)                          # (N, M, N); own column: leave-one-out centroid
    sim = w * cos + b                                             # s3:27
    _, per = ops.calc_loss(sim, eps=eps, variant=variant)         # (N, M)
    return per[rank * n:(rank + 1) * n].sum()
