"""Development aid for rocprofv3 --pmc runs: a few launches of one implementation at cfg2, B=4096, forward only or with
gradients.  usage: python tools/traffic_probe.py impl {grad|fwd}"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402

impl, mode = sys.argv[1], sys.argv[2]
dev = torch.device("cuda:0")
B, N, M, D = 4096, 64, 10, 256
E = bench.synth(B, N, M, D, 1234, dev)
w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", impl), dev)
out = None
if mode == "grad":
    out = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None, dE=torch.empty_like(E), dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
for _ in range(3):
    if mode == "grad":
        GF.loss_fwd_bwd(E, w, b, impl=impl, out=out, workspace=ws)
    else:
        GF.loss_fwd_bwd(E, w, b, impl=impl, need_grad=False, workspace=ws)
torch.cuda.synchronize()
