"""Debug aid: one TILED launch at a shape that takes the LDS-DMA kernels, each output against the fp64 oracle, with the
structure of the dE error (which rows / columns are off).  usage: python tools/dbg_tiled.py [B N M D]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import ge2e_oracle as orc  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402

B, N, M, D = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (192, 256, 10, 256)
dev = torch.device("cuda:0")
E = bench.synth(B, N, M, D, 1234, dev)
w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
out = GF.loss_fwd_bwd(E, w, b, impl="tiled", need_per=True)
torch.cuda.synchronize()
for i in (0, B - 1):
    ref = orc.closed_form(E[i].cpu().numpy(), 10.0, -5.0, variant="softmax")
    dE = out.dE[i].cpu().numpy().astype(np.float64).reshape(N * M, D)
    rdE = ref["dE"].reshape(N * M, D)
    per = out.per[i].cpu().numpy().astype(np.float64).reshape(-1)
    print(f"batch {i}: loss {float(out.loss[i]):.6f} ref {ref['loss']:.6f}  per max abs {np.abs(per - ref['per'].reshape(-1)).max():.3e}"
          f"  dw {float(out.dw[i]):.6f} ref {ref['dw']:.6f}  db {float(out.db[i]):.6f} ref {ref['db']:.6f}")
    pe = np.abs(per - ref["per"].reshape(-1))
    bad_p = np.nonzero(pe > 1e-3)[0]
    print(f"   per: rows off {bad_p.size} / {N * M}; first bad {bad_p[:40]}")
    err = np.abs(dE - rdE)
    print(f"   dE rel fro {np.linalg.norm(dE - rdE) / np.linalg.norm(rdE):.3e}; rows off {(err.max(1) > 1e-6).sum()} / {N * M},"
          f" cols off {(err.max(0) > 1e-6).sum()} / {D}")
    bad_r = np.nonzero(err.max(1) > 1e-6)[0]
    bad_c = np.nonzero(err.max(0) > 1e-6)[0]
    print("   first bad rows", bad_r[:24], " first bad cols", bad_c[:24])
