"""GE2ELoss.forward + loss.backward() captured ONCE in a HIP graph (torch.cuda.CUDAGraph) and replayed: what the module
path costs per step when the launch-bound host work (Python, autograd dispatch, allocator) is taken out of the loop.
Usage: python tools/graph_module_step.py [N M D]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import faulthandler  # noqa: E402

import numpy as np  # noqa: E402

faulthandler.enable()
import torch  # noqa: E402

from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams  # noqa: E402


def measure(N, M, D, steps=200):
    dev = torch.device("cuda:0")
    mod = GE2ELoss(HParams(device=dev))
    e = torch.randn(N, M, D, device=dev)
    e = (e / e.norm(dim=-1, keepdim=True)).requires_grad_(True)

    def step():
        e.grad = None
        mod.zero_grad(set_to_none=True)
        loss = mod(e)
        loss.backward()
        return loss

    # eager reference
    for _ in range(5):
        ref = step()
    torch.cuda.synchronize()
    ref_loss, ref_grad, ref_w = float(ref.detach()), e.grad.clone(), mod.w.grad.clone()
    del ref
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    eager_us = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])) * 1e3

    # capture (side stream warm-up as torch's recipe asks, then one capture of forward + backward)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    e.grad = None
    mod.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        static_loss = mod(e)
        static_loss.backward()
    g.replay()
    torch.cuda.synchronize()
    ok = abs(float(static_loss.detach()) - ref_loss) <= 1e-6 * abs(ref_loss) and torch.equal(e.grad, ref_grad) and torch.equal(mod.w.grad, ref_w)
    ev[0].record()
    for i in range(steps):
        g.replay()
        ev[i + 1].record()
    torch.cuda.synchronize()
    graph_us = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])) * 1e3
    return eager_us, graph_us, ok


if __name__ == "__main__":
    shapes = [tuple(int(x) for x in sys.argv[1:4])] if len(sys.argv) >= 4 else [(2, 16, 256), (4, 5, 256), (64, 10, 256)]
    for (N, M, D) in shapes:
        eager_us, graph_us, ok = measure(N, M, D)
        print(f"N={N} M={M} D={D}: module step eager {eager_us:.1f} us, as a replayed HIP graph {graph_us:.1f} us, "
              f"results {'identical to the eager step' if ok else 'DIFFER'}", flush=True)
