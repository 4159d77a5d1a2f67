"""A few GE2ELoss.forward + backward() steps for a rocprofv3 --kernel-trace run (what the device executes per step).
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/trace_module_step.py [N M D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams  # noqa: E402

N, M, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 10, 256)
dev = torch.device("cuda:0")
mod = GE2ELoss(HParams(device=dev))
e = torch.nn.functional.normalize(torch.randn(N, M, D, device=dev), dim=-1).requires_grad_(True)
for _ in range(30):
    e.grad = None
    mod.zero_grad(set_to_none=True)
    mod(e).backward()
torch.cuda.synchronize()
