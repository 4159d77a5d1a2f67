"""Data-parallel restatement of the reference's training step.

Mirrors ``embedding_model_GE2E/s4_train_embed_model.py`` (``TrainEmbedModel``):
  s4:35-42    SGD over two param groups: encoder, then the loss's (w, b)
  s4:167-192  (N,M,T,F) -> (N*M,T,F), random permutation, encoder, un-permute, (N,M,D)
  s4:196-203  loss, zero_grad, backward, clip_grad_norm_(encoder, 3.0), clip_grad_norm_(loss, 1.0), step
  s4:261-264  LR halving touches param_groups[0] only

New here (the reference is single-process): one process per GPU, whole (N,M) batches per rank
 and ONE flat-bucket all-reduce (mean) of encoder + (w,b) gradients between
``backward()`` and the clips -- RCCL over xGMI when the process group's backend is "nccl".
The parameters' ``.grad`` are views into one persistent flat buffer, so the collective needs no
per-step flatten/unflatten and there is exactly one message per step (5.9 MB for the reference
encoder: latency-bound on xGMI, so it is not split into buckets).  No per-step host sync: the
loss is returned as a device tensor (the reference's s4:205 ``.to("cpu")`` is left to the caller).
"""
from __future__ import annotations

import random
from typing import Optional

import torch
import torch.distributed as dist


class DPTrainer:
    def __init__(self, model: torch.nn.Module, loss_module: torch.nn.Module, lr: float = 0.05,
                 clip_model: float = 3.0, clip_loss: float = 1.0,
                 process_group: Optional["dist.ProcessGroup"] = None, seed: Optional[int] = None):
        self.model = model
        self.ge2e_loss = loss_module
        self.lr = lr
        self.clip_model, self.clip_loss = clip_model, clip_loss
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.optimizer = torch.optim.SGD(
            [{"params": self.model.parameters()}, {"params": self.ge2e_loss.parameters()}], lr=lr)  # s4:35-42
        self._rng = random.Random(seed)
        # one flat gradient bucket; every .grad is a view into it
        self._params = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        total = sum(p.numel() for p in self._params)
        dev, dt = self._params[0].device, self._params[0].dtype
        self.flat_grad = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in self._params:
            n = p.numel()
            p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n
        if self.world > 1:  # start from identical weights (rank 0's)
            for p in self._params:
                dist.broadcast(p.data, src=0, group=self.pg)

    def embed(self, mel: torch.Tensor) -> torch.Tensor:
        """(N,M,T,F) -> (N,M,D) through the encoder with the reference's perm/unperm (s4:174-192)."""
        n_spk, n_utt = mel.shape[0], mel.shape[1]
        total = n_spk * n_utt
        flat = mel.reshape(total, mel.shape[2], mel.shape[3])
        perm = self._rng.sample(range(total), total)
        unperm = [0] * total
        for i, j in enumerate(perm):
            unperm[j] = i
        emb = self.model(flat[perm])[unperm]
        return emb.reshape(n_spk, n_utt, emb.shape[1]).contiguous()

    def step(self, mel: torch.Tensor) -> torch.Tensor:
        """One training step on this rank's (N,M,T,F) batch.  Returns the local loss (device tensor)."""
        emb = self.embed(mel)
        loss = self.ge2e_loss(emb)  # s4:196
        self.flat_grad.zero_()      # s4:199 (grads stay views of the bucket)
        loss.backward()             # s4:200
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_grad.div_(self.world)  # mean over ranks: keeps lr and the clip thresholds meaningful
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_model)      # s4:201
        torch.nn.utils.clip_grad_norm_(self.ge2e_loss.parameters(), self.clip_loss)   # s4:202
        self.optimizer.step()       # s4:203
        return loss.detach()

    def halve_lr(self):
        """s4:261-264: only the encoder group's lr is halved; the (w,b) group keeps its own."""
        self.lr = self.lr / 2
        self.optimizer.param_groups[0]["lr"] = self.lr
