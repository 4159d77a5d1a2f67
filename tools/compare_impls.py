"""Interleaved timing of implementations in ONE process (cdna_hip_programming.md rule 24): cfg2, several B.
usage: python tools/compare_impls.py [impl ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402

impls = sys.argv[1:] or ["fused_split", "team"]
dev = torch.device("cuda:0")
N, M, D = 64, 10, 256
w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
for B in (1, 8, 32, 64, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096):
    E = bench.synth(B, N, M, D, 1234, dev)
    out = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None, dE=torch.empty_like(E), dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
    ws = {i: GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, "softmax", i), dev) for i in impls}
    res = {i: [] for i in impls}
    iters = 200 if B <= 64 else (50 if B <= 1024 else 15)
    for rnd in range(5):
        for i in impls:
            for _ in range(3):
                GF.loss_fwd_bwd(E, w, b, impl=i, out=out, workspace=ws[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                GF.loss_fwd_bwd(E, w, b, impl=i, out=out, workspace=ws[i])
            e1.record()
            torch.cuda.synchronize()
            res[i].append(e0.elapsed_time(e1) / iters * 1e3)
    print(f"B={B:5d}: " + "  ".join(f"{i} {np.median(res[i]):9.1f} us ({B / np.median(res[i]):.3f} M/s)" for i in impls), flush=True)
