"""Drop-in for the reference's loss module.

Mirrors ``embedding_model_GE2E/s3_loss_function_GE2E.py`` (class ``GE2ELoss``,
s3:6-127): same constructor argument (``hp`` with ``hp.general.device`` and
``hp.general.small_err``), same parameter names/initial values (``w`` = 10,
``b`` = -5, 0-dim fp32 -- s3:16-17, so ``state_dict`` round-trips and s4:35-42's
second SGD param group works), same ``forward(embeddings (N,M,D)) -> scalar`` and
the same static helpers.  The arithmetic runs in libge2e_hip.so.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as GF


class _HPGeneral:
    def __init__(self, device, small_err):
        self.device = device
        self.small_err = small_err


class _HPSection:
    pass


class HParams:
    """Smallest object with the two fields the loss reads (strings/constants.py:31,34), plus an empty ``m_ge2e``
    section for the callers that write into it (``calculate_ERR`` sets ``test_N`` / ``test_M``, s5:17-18)."""

    def __init__(self, device="cuda:0", small_err=1e-6):
        self.general = _HPGeneral(torch.device(device), small_err)
        self.m_ge2e = _HPSection()


class _GraphedLossFunction(torch.autograd.Function):
    """``graph=True``: forward = copy into the static input + one graph replay (graphed.StaticLossStep); backward (only
    reached when somebody differentiates THROUGH the loss -- a plain ``loss.backward()`` takes the shortcut below) scales
    the static gradients by the incoming one, like functional._GE2ELossFunction."""

    @staticmethod
    def forward(ctx, embeddings, w, b, step):
        step.run(embeddings)
        ctx.step, ctx.serial = step, step.serial
        return step.loss1[0].clone()             # (a copy: callers collect the losses of many steps, e.g. DPTrainer.fit)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        step = ctx.step
        if ctx.serial != step.serial:
            raise RuntimeError("GE2ELoss(graph=True): this loss belongs to an earlier forward; its static gradient buffers "
                               "have been overwritten by a later call (use graph=False to keep several losses alive)")
        from . import _lib
        g = grad_out
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        n, m, d = step.shape
        need_e, need_w, need_b = ctx.needs_input_grad[:3]
        gE = torch.empty_like(step.dE3) if need_e else None
        gwb = torch.empty(2, dtype=torch.float32, device=g.device) if (need_w or need_b) else None
        o = step.out
        with torch.cuda.device(g.device):
            code = _lib.load().ge2e_scale_grads(
                o.dE.data_ptr(), o.dw.data_ptr(), o.db.data_ptr(), g.data_ptr(), 1, 1, n, m, d,
                gE.data_ptr() if need_e else None, gwb.data_ptr() if need_w else None,
                gwb.data_ptr() + 4 if need_b else None, GF._stream_ptr(g))
        _lib.check(code, "ge2e_scale_grads")
        return (gE, gwb[0].reshape(step.dw0.shape) if need_w else None, gwb[1].reshape(step.db0.shape) if need_b else None,
                None)


def _has_hooks(t) -> bool:
    return bool(getattr(t, "_backward_hooks", None)) or bool(getattr(t, "_post_accumulate_grad_hooks", None))


class GE2ELoss(nn.Module):

    def __init__(self, hp, variant: str = "softmax", impl: str = "auto", graph: bool = False):
        """``graph=True`` (not in the reference): the training step of ONE fixed-shape batch per call -- s4:193-205 -- is
        served from a HIP graph over static buffers once the same (N, M, D) has come twice in a row: ``forward`` copies the
        embeddings in and replays the fused launch, and a plain ``loss.backward()`` publishes the launch's own dE / dw / db
        as the gradients without going through the autograd engine (same bits: the engine would multiply them by 1.0).
        The ``.grad`` tensors the shortcut sets are STATIC -- overwritten by the next forward (a ``.grad`` that is still attached
        then is cloned first, so accumulating over several steps stays correct); the returned loss is a copy.  Anything else -- another
        shape, a (B, N, M, D) stack, no-grad mode, a stream that is capturing, hooks on the tensors, ``backward`` with
        arguments -- takes the eager node."""
        super().__init__()
        self.device = hp.general.device  # s3:11
        self.hp = hp  # s3:12
        self.variant = variant
        self.impl = impl
        self.graph = bool(graph)
        self._steps = {}          # key (graphed.StaticLossStep.key_of) -> step; at most _MAX_STEPS, oldest dropped
        self._last_shape = None
        # s3:16-17 -- scale and shift of eq. (5), learnable
        self.w = nn.Parameter(torch.tensor(10.0).to(self.device), requires_grad=True)
        self.b = nn.Parameter(torch.tensor(-5.0).to(self.device), requires_grad=True)

    def forward(self, embeddings):
        """embeddings (N,M,D) [or (B,N,M,D)] on hp.general.device -> loss (s3:19-30).

        Like the reference, w is NOT clamped (s3:22 discards torch.clamp's result), the
        loss is a sum over all (speaker, utterance) rows (s3:126), and the gradient flows
        through both cosine norms.
        """
        if self.graph:
            loss = self._forward_graphed(embeddings)
            if loss is not None:
                return loss
        return GF.ge2e_loss(embeddings, self.w, self.b, eps=self.hp.general.small_err,
                            variant=self.variant, impl=self.impl)

    _MAX_STEPS = 4

    def __getstate__(self):
        # captured graphs and their static buffers belong to THIS object: a copy (copy.deepcopy, pickling, DataParallel's
        # replicate) starts without them and captures its own
        state = self.__dict__.copy()
        state["_steps"] = {}
        state["_last_shape"] = None
        return state

    def _forward_graphed(self, e):
        if (e.dim() != 3 or not e.is_cuda or e.dtype != torch.float32 or not e.is_contiguous() or e.device != self.w.device
                or not torch.is_grad_enabled() or torch.cuda.is_current_stream_capturing()):
            self._last_shape = None
            return None
        from .graphed import StaticLossStep
        key = StaticLossStep.key_of(self, e.shape)
        step = self._steps.get(key)
        if step is None:
            if self._last_shape != key:          # a shape is captured when it comes the second time in a row
                self._last_shape = key
                return None
            while len(self._steps) >= self._MAX_STEPS:
                self._steps.pop(next(iter(self._steps)))
            with torch.no_grad():
                step = self._steps[key] = StaticLossStep(self, e.shape, e.device)
        w, b = self.w, self.b
        # a .grad that still IS one of the static buffers (the caller did not set it to None): keep its value out of the replay's way
        for t in (e, w, b):
            g = t.grad if t.is_leaf else None
            if g is not None and g.data_ptr() in step.static_ptrs:
                t.grad = g.clone()
        loss = _GraphedLossFunction.apply(e, w, b, step)
        serial = step.serial
        import weakref
        wloss = weakref.ref(loss)

        def backward(gradient=None, retain_graph=None, create_graph=False, inputs=None):
            if (gradient is not None or create_graph or inputs is not None or step.serial != serial
                    or _has_hooks(e) or _has_hooks(w) or _has_hooks(b) or _has_hooks(wloss())):
                return torch.Tensor.backward(wloss(), gradient, retain_graph, create_graph, inputs)
            for t, g in ((w, step.dw0), (b, step.db0)):
                if t.requires_grad:
                    if t.grad is None:
                        t.grad = g
                    else:
                        t.grad.add_(g)
            if e.requires_grad:
                if not e.is_leaf:                # the encoder's graph continues behind the embeddings
                    torch.autograd.backward(e, step.dE3, retain_graph=retain_graph)
                elif e.grad is None:
                    e.grad = step.dE3
                else:
                    e.grad.add_(step.dE3)

        loss.backward = backward                 # instance attribute: shadows Tensor.backward for this loss only
        return loss

    # eq. (1) -- s3:33-38
    @staticmethod
    def get_centroids(embeddings):
        return GF.centroids(embeddings)

    # eq. (5) cosines -- s3:41-80 (differentiable in both arguments, like the reference's)
    @staticmethod
    def get_cos_sim(embeddings, centroids, hp):
        # like the reference: `centroids` feeds every other-speaker column, the own-speaker column uses the
        # leave-one-out centroid of `embeddings` (s3:44-57, 64-78)
        return GF.cos_sim(embeddings, centroids, eps=hp.general.small_err)

    # eq. (8) -- s3:83-93, dead code in the reference (no caller); kept as an API stub
    @staticmethod
    def get_centroid(embeddings, speaker_num, utterance_num):
        spk = embeddings[speaker_num]
        return (spk.sum(dim=0) - spk[utterance_num]) / (spk.shape[0] - 1)

    # s3:95-112
    @staticmethod
    def get_utterance_centroids(embeddings):
        return GF.utterance_centroids(embeddings)

    # eq. (6) -- s3:114-127: returns (loss, per_embedding_loss (N,M))
    @staticmethod
    def calc_loss(sim_matrix, hp, variant: str = "softmax"):
        return GF.calc_loss(sim_matrix, eps=hp.general.small_err, variant=variant)
