"""Interleaved timing of the one-wave-per-batch kernel against the workgroup-per-batch kernel on the reference's own
shapes (a few dozen rows), B = 1 and 4096, C ABI with reused buffers.  Usage (GPU box): python tools/small_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402

dev = torch.device("cuda:0")
for (N, M, D) in ((4, 5, 256), (2, 16, 256), (3, 8, 256), (2, 10, 256), (4, 4, 64), (6, 2, 128),
                  (8, 5, 256), (8, 8, 256), (6, 10, 256), (3, 16, 256), (10, 4, 256), (12, 2, 256), (10, 3, 128)):
    for B in (1, 4096):
        E = bench.synth(B, N, M, D, 1, dev)
        w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
        res = {}
        for impl in ("wave", "fused_split"):
            out = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None, dE=torch.empty_like(E), dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
            ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", impl), dev)
            f = lambda: GF.loss_fwd_bwd(E, w, b, impl=impl, out=out, workspace=ws)
            for _ in range(5): f()
            torch.cuda.synchronize()
            res[impl] = float(np.median(bench.time_launches(f, 50))) * 1e3
        print((N, M, D), B, {k: round(v, 1) for k, v in res.items()}, flush=True)
