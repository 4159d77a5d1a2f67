#!/usr/bin/env python3
"""Development aid: run team2 twice on the same problem with per-thread checksums of intermediate
registers (-DGE2E_T2_DEBUG build) and report which quantity differs between the two runs."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import ge2e_oracle as orc  # noqa: E402
from speaker_embedding_ge2e_loss_amd import build  # noqa: E402

lib_path = os.path.join(build.PKG_DIR, "libge2e_hip_exp_dbg.so")
subprocess.run([build._hipcc(), "-O3", "-std=c++17", f"--offload-arch={build.ARCH}", "-fPIC", "-shared",
                f"-I{build.INCLUDE}", "-DGE2E_T2_DEBUG", "-o", lib_path] + build.sources(), check=True, stderr=subprocess.DEVNULL)
os.environ["GE2E_HIP_LIB"] = lib_path
from speaker_embedding_ge2e_loss_amd import _lib, functional as GF  # noqa: E402

lib = _lib.load()
raw = C.CDLL(lib_path)
shape = (40, 23, 7, 128)
E = orc.synth_embeddings(shape, "raw", seed=sum(shape))
ref = orc.closed_form(E, 6.0, -1.5)
dev = torch.device("cuda:0")
e = torch.as_tensor(E, device=dev)
w, b = torch.tensor(6.0, device=dev), torch.tensor(-1.5, device=dev)
names = ["xa", "ga", "gsum", "kjp", "cj_prev", "held0", "held1", "held2", "held3", "held4", "ga_at_ge", "acc_rb1", "ehel_rb1", "rc_rb1", "gb_rb1", "-"]
runs = []
for rep in range(6):
    dbg = torch.zeros(shape[0] * 8 * 16 * 512, dtype=torch.int32, device=dev)
    raw.ge2e_debug_set_t2(C.c_void_p(dbg.data_ptr()))
    dump = torch.zeros(shape[0] * 8 * 32768, dtype=torch.int32, device=dev)
    raw.ge2e_debug_set_t2_dump(C.c_void_p(dump.data_ptr()))
    o = GF.loss_fwd_bwd(e, w, b, impl="team2")
    torch.cuda.synchronize()
    dE = o.dE.cpu().numpy()
    bad = [bi for bi in range(shape[0]) if np.linalg.norm(dE[bi] - ref["dE"][bi]) / np.linalg.norm(ref["dE"][bi]) > 2e-5]
    runs.append((dbg.cpu().numpy().reshape(shape[0], 8, 16, 512), bad, dump.cpu().numpy().reshape(shape[0], 8, 32768)))
    print("rep", rep, "bad batches", bad, flush=True)
N, M = shape[1], shape[2]
spm = (N + 7) // 8
def valid(m, it, t):
    if it in (11, 12, 13, 14):
        return 16 + (t & 15) < max(0, min(spm, N - m * spm)) * M
    if 5 <= it <= 9:
        rb = (it - 5) % 5
        r = 16 * rb + (t & 15)
        return r < max(0, min(spm, N - m * spm)) * M
    if it in (2, 3, 4):
        return (t >> 6) < max(0, min(spm, N - m * spm)) and 4 * (t & 63) < shape[3]
    return True
# majority vote per entry = presumed-correct value; report deviations per run
stack = np.stack([r[0] for r in runs])
for rep in range(len(runs)):
    others = [k for k in range(len(runs)) if k != rep]
    ref_ = stack[others[0]]
    for k in others[1:3]:
        pass
    # an entry is "deviant" in this rep if it differs from at least 3 other runs that agree with each other
    agree = (stack[others[0]] == stack[others[1]]) & (stack[others[1]] == stack[others[2]])
    d = agree & (stack[rep] != stack[others[0]])
    idx = np.argwhere(d)
    seen = {}
    for bi, m, it, t in idx:
        if valid(m, it, t):
            seen.setdefault((int(bi), int(m), names[it]), []).append(int(t))
    print("rep", rep, "bad", runs[rep][1], "deviant valid entries:", len(seen))
    for k, v in sorted(seen.items()):
        print("   ", k, "threads", sorted(v)[:12], "n", len(v))

RT = (spm * M + 15) // 16 * 16
P = shape[3] + 16
sections = [("ET", 0, RT * P), ("G", 8192, RT * 72), ("RS", 12288, RT * 8), ("ga", 13312, 4 * 512 * 4)]
dstack = np.stack([r[2] for r in runs])
for rep in range(len(runs)):
    others = [k for k in range(len(runs)) if k != rep]
    agree = (dstack[others[0]] == dstack[others[1]]) & (dstack[others[1]] == dstack[others[2]])
    d = agree & (dstack[rep] != dstack[others[0]])
    for name, off, n in sections:
        idx = np.argwhere(d[:, :, off:off + n])
        if len(idx):
            seen = {}
            for bi, m, k in idx:
                seen.setdefault((int(bi), int(m)), []).append(int(k))
            print("rep", rep, name, "deviant dwords:", {k: (len(v), sorted(v)[:8]) for k, v in sorted(seen.items())[:10]})
