"""The loss step -- ``GE2ELoss.forward`` + ``loss.backward()`` (s3:19-30, s4:196-200) -- captured ONCE in a HIP graph and
replayed.  At one (N, M) batch per step the eager module path is host-bound: five device operations worth 25-40 us sit
in 150-220 us of Python, autograd dispatch and allocator work per step.  Replaying the captured step costs what the device
needs (cfg2: 43 us, the reference's N = 2, M = 16: 26 us on MI355X) and gives the same bits as the eager step.

Every launch of libge2e_hip.so is capture-safe (no host synchronisation, no allocation inside the library, the team
kernel's control block is zeroed by a kernel rather than a memset node), so a caller may equally capture its WHOLE training
step -- encoder, loss, optimizer -- with ``torch.cuda.graph``; this class is the loss-only form of that."""
from __future__ import annotations

from typing import Optional

import torch


class GraphedLossStep:
    """``step = GraphedLossStep(loss_module, (N, M, D))``; then per training step::

        loss = step(embeddings)      # copies into the static input, replays forward + backward
        step.input_grad              # dLoss/d embeddings  (N, M, D)
        loss_module.w.grad, loss_module.b.grad

    ``loss`` and the gradients are STATIC tensors, overwritten by the next call (clone what has to outlive it)."""

    def __init__(self, loss_module: torch.nn.Module, shape, *, device: Optional[torch.device] = None, warmup: int = 3,
                 direct: bool = False):
        """``direct``: capture the fused launch alone and publish ITS gradients (the launch produces dE, dw, db next to the
        loss; ``loss.backward()`` only multiplies them by the incoming 1.0): two graph nodes fewer -- autograd's ``ones_like``
        fill and the scaling kernel -- and the same bits.  For a ``GE2ELoss`` of this package (it needs the module's
        ``w``, ``b``, ``variant``, ``impl``)."""
        params = list(loss_module.parameters())
        dev = device or params[0].device
        self.module = loss_module
        self.input = torch.zeros(*shape, dtype=torch.float32, device=dev).requires_grad_(True)
        with torch.no_grad():   # something finite to warm up on
            self.input.copy_(torch.nn.functional.normalize(torch.randn(*shape, device=dev), dim=-1))
        # the step's workspace: allocated and initialised HERE, outside the capture, owned by this object (the library's
        # control block cleans itself after every call, so the captured step needs no initialisation node)
        from . import functional as GF
        n_, m_, d_ = (int(x) for x in shape)
        self.workspace = GF.alloc_workspace(
            GF.workspace_bytes(1, n_, m_, d_, getattr(loss_module, "variant", "softmax"), getattr(loss_module, "impl", "auto")), dev)
        if direct:
            self._capture_direct(dev, warmup)
            return
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), GF.workspace_override(self.workspace):       # torch's capture recipe: warm up on a side stream
            for _ in range(max(1, warmup)):
                self._zero()
                self.module(self.input).backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        self._zero()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), GF.workspace_override(self.workspace):
            self.loss = self.module(self.input)
            self.loss.backward()
        self.input_grad = self.input.grad
        self.loss = self.loss.detach()

    def _capture_direct(self, dev, warmup):
        from . import functional as GF
        m = self.module
        e4 = self.input.detach().unsqueeze(0)
        out = GF.LossOutputs(loss=torch.empty(1, device=dev), per=None, dE=torch.empty_like(e4),
                             dw=torch.empty(1, device=dev), db=torch.empty(1, device=dev))
        eps = float(m.hp.general.small_err)
        w, b = m.w.detach(), m.b.detach()

        def launch():
            GF.loss_fwd_bwd(e4, w, b, eps=eps, variant=m.variant, impl=m.impl, out=out, workspace=self.workspace)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                launch()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            launch()
        self.loss = out.loss[0]
        self.input_grad = out.dE[0]
        self.input.grad = self.input_grad
        m.w.grad = out.dw[0].reshape(m.w.shape)
        m.b.grad = out.db[0].reshape(m.b.shape)

    def _zero(self):
        self.input.grad = None
        for p in self.module.parameters():
            p.grad = None

    def __call__(self, embeddings: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Replays the step on ``embeddings`` (copied into ``self.input``; None: the caller has written ``self.input``)."""
        if embeddings is not None:
            if embeddings.shape != self.input.shape:
                raise ValueError(f"captured for {tuple(self.input.shape)}, got {tuple(embeddings.shape)}")
            with torch.no_grad():
                self.input.copy_(embeddings)
        self.graph.replay()
        return self.loss


class StaticLossStep:
    """The fused launch of ONE (N, M, D) batch -- loss, dE, dw, db (s3:19-30 + s4:200) -- captured alone in a HIP graph over
    STATIC buffers; nothing is published.  ``GE2ELoss(hp, graph=True)`` serves ``module(embeddings)`` /
    ``loss.backward()`` from it (loss.py): ``run`` copies the embeddings into the static input and replays; the gradients
    are then already in ``dE3`` / ``dw0`` / ``db0`` for an incoming gradient of 1."""

    def __init__(self, loss_module: torch.nn.Module, shape, dev: torch.device, warmup: int = 2):
        from . import functional as GF
        n_, m_, d_ = (int(x) for x in shape)
        m = loss_module
        self.shape = (n_, m_, d_)
        self.e4 = torch.nn.functional.normalize(torch.randn(1, n_, m_, d_, device=dev), dim=-1)   # finite values to warm up on
        self.e3 = self.e4[0]
        self.dE = torch.empty_like(self.e4)
        self.sc = torch.empty(3, 1, dtype=torch.float32, device=dev)          # loss | dw | db
        self.out = GF.LossOutputs(loss=self.sc[0], per=None, dE=self.dE, dw=self.sc[1], db=self.sc[2])
        self.dE3 = self.dE[0]
        self.loss1, self.dw0, self.db0 = self.sc[0], self.sc[1, 0].reshape(m.w.shape), self.sc[2, 0].reshape(m.b.shape)
        self.static_ptrs = (self.dE3.data_ptr(), self.dw0.data_ptr(), self.db0.data_ptr())
        self.key = StaticLossStep.key_of(m, self.shape)
        self.workspace = GF.alloc_workspace(GF.workspace_bytes(1, n_, m_, d_, m.variant, m.impl), dev)
        self.serial = 0                       # replays so far: a loss tensor of replay k is stale once k + 1 has run
        eps = float(m.hp.general.small_err)
        w, b = m.w.detach(), m.b.detach()     # their ADDRESSES go into the graph (the key holds them)

        def launch():
            GF.loss_fwd_bwd(self.e4, w, b, eps=eps, variant=m.variant, impl=m.impl, out=self.out, workspace=self.workspace)

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                launch()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            launch()

    @staticmethod
    def key_of(m, shape):
        return (tuple(int(x) for x in shape), float(m.hp.general.small_err), m.variant, m.impl, m.w.data_ptr(), m.b.data_ptr())

    def run(self, embeddings: torch.Tensor) -> None:
        self.e3.copy_(embeddings)
        self.graph.replay()
        self.serial += 1


def measure_step_latency(shape, variant: str = "softmax", impl: str = "auto", steps: int = 100, device: str = "cuda:0",
                         direct: bool = False) -> float:
    """Median device time (us) of one replayed loss step of ``shape`` = (N, M, D), from events on the launch stream."""
    from . import GE2ELoss, HParams
    dev = torch.device(device)
    step = GraphedLossStep(GE2ELoss(HParams(device=dev), variant=variant, impl=impl), tuple(shape), direct=direct)
    step()
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return float(ts[len(ts) // 2]) * 1e3


if __name__ == "__main__":   # python -m speaker_embedding_ge2e_loss_amd.graphed N M D [variant] [impl]  ->  one JSON line
    import json
    import sys
    n, m, d = (int(x) for x in sys.argv[1:4])
    var = sys.argv[4] if len(sys.argv) > 4 else "softmax"
    imp = sys.argv[5] if len(sys.argv) > 5 else "auto"
    print(json.dumps({"latency_module_graph_b1_us": measure_step_latency((n, m, d), var, imp),
                      "latency_module_graph_direct_b1_us": measure_step_latency((n, m, d), var, imp, direct=True)}))
