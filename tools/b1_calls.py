"""Single-batch calls back to back (the bench's latency_b1 leg on its own, for a kernel trace):
rocprofv3 --kernel-trace --stats -d gpurun_out/b1 -- python3 tools/b1_calls.py [impl]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402

impl = sys.argv[1] if len(sys.argv) > 1 else "auto"
dev = torch.device("cuda:0")
N, M, D = 64, 10, 256
E = bench.synth(1, N, M, D, 1234, dev)
w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
impl1 = GF.resolve_impl(1, N, M, D, "softmax", impl)
ws = GF.alloc_workspace(GF.workspace_bytes(1, N, M, D, "softmax", impl1), dev)
out = GF.loss_fwd_bwd(E, w, b, impl=impl1, workspace=ws)
call = lambda: GF.loss_fwd_bwd(E, w, b, impl=impl1, out=out, workspace=ws)  # noqa: E731
for _ in range(20):
    call()
t = bench.time_launches(call, 200)
print(f"{impl1}: B=1 call period median {np.median(t) * 1e3:.2f} us, min {np.min(t) * 1e3:.2f} us")
