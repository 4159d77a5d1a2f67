"""Data-parallel restatement of the reference's training step.

Mirrors ``embedding_model_GE2E/s4_train_embed_model.py`` (``TrainEmbedModel``):
  s4:35-42    SGD over two param groups: encoder, then the loss's (w, b)
  s4:167-192  (N,M,T,F) -> (N*M,T,F), random permutation, encoder, un-permute, (N,M,D)
  s4:196-203  loss, zero_grad, backward, clip_grad_norm_(encoder, 3.0), clip_grad_norm_(loss, 1.0), step
  s4:261-264  LR halving touches param_groups[0] only
  s4:61-110   batched test loss: eval mode, same perm/unperm, mean of the per-batch losses
  s4:112-135  checkpoint = the ENCODER's state_dict only (the loss's w, b are not saved)

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
                 process_group: Optional["dist.ProcessGroup"] = None, seed: Optional[int] = None,
                 fused_tail: bool = False, shared_device: bool = False):
        # fused_tail: the encoder returns its raw projection (SpeakerEncoder(normalize=False)) and the
        # L2-normalise + un-permute + (N,M,D) layout run as one HIP kernel instead of
        # norm / divide / index_select / contiguous
        self.fused_tail = fused_tail
        self.model = model
        self.ge2e_loss = loss_module
        # shared_device: other streams or processes keep CUs of this GPU busy while the loss runs (communication overlapped
        # with the next step, several trainers on one device).  The loss's AUTO then leaves out the eight-CU team kernel,
        # whose workgroups wait for each other (impl "auto_no_team", include/ge2e_hip.h).  Not needed for this trainer's
        # own collective: the all-reduce below is issued on the same stream between backward() and the next step's loss,
        # so by stream order the two never run side by side (an argument from the ordering, not a measurement: the
        # single-rank RCCL test only smoke-tests the code path).
        if shared_device and getattr(loss_module, "impl", None) == "auto":
            loss_module.impl = "auto_no_team"
        want_norm = not fused_tail
        if getattr(model, "normalize", want_norm) != want_norm:
            raise ValueError(f"fused_tail={fused_tail} needs an encoder with normalize={want_norm}: the L2-normalisation "
                             "of s2:34 must happen exactly once (in the encoder, or in the fused tail kernel)")
        self.lr = lr
        self.clip_model, self.clip_loss = clip_model, clip_loss
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.optimizer = torch.optim.SGD(
            [{"params": self.model.parameters()}, {"params": self.ge2e_loss.parameters()}], lr=lr)  # s4:35-42
        self._rng = random.Random(seed)
        # one flat gradient bucket; every .grad is a view into it
        self._params = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        if not self._params:
            raise ValueError("nothing to train: no parameter requires grad")
        total = sum(p.numel() for p in self._params)
        dev, dt = self._params[0].device, self._params[0].dtype
        for p in self._params:  # one bucket = one dtype on one device
            if p.device != dev or p.dtype != dt:
                raise ValueError(f"all trainable parameters must share one device and dtype "
                                 f"(got {p.device}/{p.dtype} beside {dev}/{dt})")
        self.flat_grad = torch.zeros(total, device=dev, dtype=dt)
        self._views = []
        off = 0
        for p in self._params:
            n = p.numel()
            self._views.append(self.flat_grad[off:off + n].view(p.shape))
            off += n
        self._bind_grads()
        if self.world > 1:  # start from identical weights (rank 0's)
            for p in self._params:
                dist.broadcast(p.data, src=0, group=self.pg)

    def _bind_grads(self):
        """Make every .grad the parameter's view of the bucket again.  A caller's
        ``optimizer.zero_grad()`` (set_to_none is torch's default) or ``model.zero_grad()`` drops
        them; autograd would then allocate fresh .grad tensors and the collective would reduce a
        bucket nobody writes."""
        for p, v in zip(self._params, self._views):
            if p.grad is not v:
                p.grad = v

    def embed(self, mel: torch.Tensor) -> torch.Tensor:
        """(N,M,T,F) -> (N,M,D) through the encoder with the reference's perm/unperm (s4:174-192)."""
        n_spk, n_utt = mel.shape[0], mel.shape[1]
        total = n_spk * n_utt
        flat = mel.reshape(total, mel.shape[2], mel.shape[3])
        perm = self._rng.sample(range(total), total)
        unperm = [0] * total
        for i, j in enumerate(perm):
            unperm[j] = i
        if self.fused_tail:
            from . import functional as GF
            return GF.normalize_unperm(self.model(flat[perm]), unperm, shape=(n_spk, n_utt))
        emb = self.model(flat[perm])[unperm]
        return emb.reshape(n_spk, n_utt, emb.shape[1]).contiguous()

    def _loss_of(self, mel: torch.Tensor) -> torch.Tensor:
        """embed + loss (s4:174-196).  With the fused tail and a shape the raw entry takes (the reference's own training
        shapes), normalise + un-permute + loss + their backward are ONE launch on the encoder's raw projection
        (functional.ge2e_loss_raw); otherwise embed() and the loss module."""
        # the raw entry picks its own kernel and never runs the module's forward: only for a loss module that leaves the
        # choice open (impl "auto" / "auto_no_team") and has no forward hooks registered
        if self.fused_tail and getattr(self.ge2e_loss, "variant", None) is not None \
                and getattr(self.ge2e_loss, "impl", "auto") in ("auto", "auto_no_team") \
                and not self.ge2e_loss._forward_hooks and not self.ge2e_loss._forward_pre_hooks:
            from . import functional as GF
            n_spk, n_utt = mel.shape[0], mel.shape[1]
            d_out = getattr(self.model, "embedding_size", None)
            if d_out is not None and GF.raw_supported(n_spk, n_utt, int(d_out)):
                total = n_spk * n_utt
                flat = mel.reshape(total, mel.shape[2], mel.shape[3])
                perm = self._rng.sample(range(total), total)
                unperm = [0] * total
                for i, j in enumerate(perm):
                    unperm[j] = i
                y = self.model(flat[perm])
                return GF.ge2e_loss_raw(y, unperm, self.ge2e_loss.w, self.ge2e_loss.b, (n_spk, n_utt),
                                        eps=self.ge2e_loss.hp.general.small_err, variant=self.ge2e_loss.variant)
        return self.ge2e_loss(self.embed(mel))  # s4:196

    def step(self, mel: torch.Tensor) -> torch.Tensor:
        """One training step on this rank's (N,M,T,F) batch.  Returns the local loss (device tensor)."""
        loss = self._loss_of(mel)
        self._bind_grads()
        self.flat_grad.zero_()      # s4:199 (grads stay views of the bucket)
        loss.backward()             # s4:200
        if self.world > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            self.flat_grad.div_(self.world)  # mean over ranks: keeps lr and the clip thresholds meaningful
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.clip_model)      # s4:201
        torch.nn.utils.clip_grad_norm_(self.ge2e_loss.parameters(), self.clip_loss)   # s4:202
        self.optimizer.step()       # s4:203
        return loss.detach()

    @torch.no_grad()
    def eval_loss(self, mel_batches, require_batches: bool = False) -> float:
        """s4:61-110: mean over the test batches of the loss in eval mode (same perm/unperm, drawn
        from the same generator as the training steps, like the reference's global ``random``).
        The per-batch losses stay on the device; there is ONE host read at the end.  An empty iterable gives NaN like
        the reference's np.mean([]) -- or raises with ``require_batches`` (``fit`` asks for that, so that a NaN it sees
        is a diverged loss and not an exhausted one-shot generator)."""
        was_training = self.model.training
        self.model.eval()       # s4:69
        losses = []
        for mel in mel_batches:
            losses.append(self._loss_of(mel).detach().reshape(()))
        if was_training:
            self.model.train()  # s4:107
        if not losses:
            if require_batches:
                raise ValueError("test_batches yielded no batch (pass a re-iterable, not a one-shot generator)")
            return float("nan")  # np.mean([]) in the reference (s4:109)
        mean = torch.stack(losses).mean()
        if self.world > 1:      # every rank evaluated its own share of the test set
            dist.all_reduce(mean, op=dist.ReduceOp.SUM, group=self.pg)
            mean = mean / self.world
        return float(mean)

    def save_checkpoint(self, path: str) -> None:
        """s4:112-135: the encoder's ``state_dict`` only, as CPU tensors (the reference moves the
        model to the CPU and back; here the weights are copied and the model stays put).  The
        loss's (w, b) are NOT in the file -- s4:130 saves ``self.model.state_dict()``.  Rank 0
        writes; the other ranks return."""
        if self.world > 1 and dist.get_rank(self.pg) != 0:
            return
        state = {k: v.detach().to("cpu", copy=True) for k, v in self.model.state_dict().items()}
        torch.save(state, path)

    def load_checkpoint(self, path: str) -> None:
        """s4:24-30 / s2:57-60: restore the encoder from an encoder-only checkpoint."""
        dev = self._params[0].device
        self.model.load_state_dict(torch.load(path, map_location=dev))

    def fit(self, train_batches, epochs: int, test_batches=None, lr_reduce: int = 2000, epoch_print: int = 100,
            checkpoint_dir: Optional[str] = None, checkpoint_interval: int = 200, save_best_weights: bool = False,
            min_test_loss: float = float("inf"), restore_existing_model: Optional[str] = None):
        """The epoch loop of s4:137-276 without its printing: per epoch the mean of the step losses
        (s4:215), every ``epoch_print`` epochs the batched test loss (s4:225-227), LR halving every
        ``lr_reduce`` epochs (s4:261-264), a checkpoint every ``checkpoint_interval`` (s4:266-267),
        a final one (s4:270) and, with ``save_best_weights``, the best test loss so far (s4:243-254).  ``train_batches`` is any re-iterable of (N,M,T,F) tensors.
        The step losses are reduced on the device: one host read per epoch instead of one per step
        (s4:205).  ``restore_existing_model``: path of an encoder-only checkpoint to start from (s4:24-30)."""
        import os
        if restore_existing_model:      # hp.m_ge2e.restore_existing_model + model_path (s4:24-30): resume from an encoder-only file
            self.load_checkpoint(restore_existing_model)
        self.model.train()
        train_losses, test_losses = [], []
        mean = float("nan")
        best = None
        for e in range(epochs):
            step_losses = [self.step(mel) for mel in train_batches]
            if not step_losses:
                # a one-shot generator is exhausted after the first epoch: NaN means and a checkpoint named after them
                # would be the silent result
                raise ValueError(f"epoch {e + 1}: train_batches yielded no batch (pass a re-iterable, e.g. "
                                 "GE2EBatchSampler.loader(N), not a one-shot generator)")
            mean = float(torch.stack(step_losses).mean())
            train_losses.append(mean)
            if test_batches is not None and (e + 1) % epoch_print == 0:
                try:
                    tl = self.eval_loss(test_batches, require_batches=True)
                except ValueError as ex:
                    raise ValueError(f"epoch {e + 1}: {ex}") from None
                test_losses.append(tl)      # a NaN here is the model's own (diverged) test loss: recorded, like s4:227
                # s4:243-254: hp.m_ge2e.save_best_weights -- a test loss under hp.m_ge2e.min_test_loss that is the best so
                # far (ties included) is saved under the name "m_best"
                if save_best_weights and checkpoint_dir is not None and tl < min_test_loss:
                    best = tl if best is None else min(best, tl)
                    if best == tl:
                        self.save_checkpoint(os.path.join(checkpoint_dir, f"m_best_epoch_{e + 1}_L_{tl:.4f}.pth"))
            if (e + 1) % lr_reduce == 0:
                self.halve_lr()
            if checkpoint_dir is not None and (e + 1) % checkpoint_interval == 0:
                self.save_checkpoint(os.path.join(checkpoint_dir, f"ckpt_epoch_{e + 1}_L_{mean:.4f}.pth"))
        if checkpoint_dir is not None and epochs > 0:
            self.save_checkpoint(os.path.join(checkpoint_dir, f"final_epoch_{epochs}_L_{mean:.4f}.pth"))
        return self.model, train_losses, test_losses

    def halve_lr(self):
        """s4:261-264: only the encoder group's lr is halved; the (w,b) group keeps its own."""
        self.lr = self.lr / 2
        self.optimizer.param_groups[0]["lr"] = self.lr


class TrainEmbedModel:
    """The reference's training object (s4:19-59, :137-276) over ``DPTrainer``: same constructor argument (``hp``), same
    attributes (``model``, ``ge2e_loss``, ``optimizer``, ``lr``, ``train_loader``, ``test_loader``, ``train_losses``,
    ``test_losses``, ``total_utterances``, ``hp``) and ``train_model(lr_reduce, epoch_print, dot_print)`` returning
    ``(model, train_losses, test_losses)`` -- so ``train_embedding_model.py:33-36`` runs against this package by changing
    its import.  Everything is built from ``hp``: the encoder (``SpeakerEncoder.from_hp``, optionally restored from
    ``hp.m_ge2e.model_path``), the HIP loss module, the two-group SGD, the resident-store loaders
    (``data.get_train_test_data_loader``) and the checkpoint folder.  ``train_model`` is ``DPTrainer.fit`` with the
    reference's schedule read from ``hp.m_ge2e`` (epochs, checkpoint interval, best-weights rule); under
    ``torch.distributed`` every rank trains on its own batches and the gradients are averaged in one bucket per step."""

    def __init__(self, hp, variant: str = "softmax", fused_tail: bool = False):
        import os
        from .encoder import SpeakerEncoder
        from .loss import GE2ELoss
        from .data import get_train_test_data_loader
        self.hp = hp
        self.model = SpeakerEncoder.from_hp(hp, normalize=not fused_tail)                    # s4:21
        self._restore = None
        if getattr(hp.m_ge2e, "restore_existing_model", False):                             # s4:24-30
            self._restore = os.path.join(hp.general.project_root, hp.m_ge2e.model_path)
            self.model.load_state_dict(torch.load(self._restore, map_location=hp.general.device))
            print(f"Pre-trained model loaded {self._restore}")
        self.ge2e_loss = GE2ELoss(hp, variant=variant)                                       # s4:33
        self.lr = hp.m_ge2e.lr                                                               # s4:41
        self.trainer = DPTrainer(self.model, self.ge2e_loss, lr=self.lr, fused_tail=fused_tail)
        self.optimizer = self.trainer.optimizer                                              # s4:35-42
        self.train_loader, self.test_loader = get_train_test_data_loader(hp)                 # s4:45
        self.checkpoint_dir = None
        if getattr(hp.m_ge2e, "checkpoint_dir", None) is not None:                           # s4:48-49
            self.checkpoint_dir = os.path.join(hp.general.project_root, hp.m_ge2e.checkpoint_dir)
            os.makedirs(self.checkpoint_dir, exist_ok=True)
        self.train_losses, self.test_losses = [], []                                         # s4:52-53
        self.total_utterances = hp.m_ge2e.training_N * hp.m_ge2e.training_M                  # s4:56

    def train_model(self, lr_reduce: int = 2000, epoch_print: int = 100, dot_print: int = 10):
        """s4:137-276.  ``dot_print`` only paces the reference's progress dots; nothing is printed per epoch here."""
        m = self.hp.m_ge2e
        model, tr, te = self.trainer.fit(
            self.train_loader, m.training_epochs, test_batches=self.test_loader, lr_reduce=lr_reduce, epoch_print=epoch_print,
            checkpoint_dir=self.checkpoint_dir, checkpoint_interval=getattr(m, "checkpoint_interval", 200),
            save_best_weights=bool(getattr(m, "save_best_weights", False)),
            min_test_loss=float(getattr(m, "min_test_loss", float("inf"))))
        self.lr = self.trainer.lr
        self.train_losses += tr
        self.test_losses += te
        return model, self.train_losses, self.test_losses
