#!/usr/bin/env python3
"""Timing experiment (NOT a product path): how much of the fused_split launch is memory traffic?

Builds variants of the library with -DGE2E_EXP=<mask> in which one stream of the kernel is pointed at
ONE batch (L2-resident) instead of its own rows, so that stream's HBM traffic disappears while the
instruction stream stays the same.  Results are wrong by construction; only the launch time is read.
  bit 1: next-batch speaker sums (sweep 1)   bit 8: sweep 2 rows
  bit 2: sweep 3 raw rows                     bit 4: dE stores
Usage (GPU box): python tools/exp_traffic.py [--masks 0,1,2,3,4,8,15]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import _lib, build  # noqa: E402


def time_variant(mask, B, iters):
    lib_path = os.path.join(build.PKG_DIR, f"libge2e_hip_exp{mask}.so")
    cmd = [build._hipcc(), "-O3", "-std=c++17", f"--offload-arch={build.ARCH}", "-fPIC", "-shared",
           f"-DGE2E_EXP={mask}", f"-I{build.INCLUDE}", "-o", lib_path] + build.sources()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lib = C.CDLL(lib_path)
    for name, (res, argt) in _lib.PROTOTYPES.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = argt
    cfg = bench.CONFIGS["cfg2"]
    N, M, D = cfg["N"], cfg["M"], cfg["D"]
    dev = torch.device("cuda:0")
    E = bench.synth(B, N, M, D, 1234, dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    loss, dw, db, dE = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32), torch.empty_like(E)
    im = _lib.IMPLS["fused_split"]
    ws = torch.empty(lib.ge2e_workspace_bytes(B, N, M, D, 0, im) + 256, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        rc = lib.ge2e_loss_fwd_bwd(E.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0, im,
                                   loss.data_ptr(), None, dE.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                   ws.data_ptr(), ws.numel(), st)
        assert rc == 0, rc
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    os.remove(lib_path)
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--masks", default="0,1,2,4,8,3,11,15")
    ap.add_argument("--batches", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    for m in [int(x) for x in a.masks.split(",")]:
        ms = time_variant(m, a.batches, a.iters)
        print(f"GE2E_EXP={m:2d}: {ms:.3f} ms per launch, {a.batches / ms * 1e3:,.0f} batches/s", flush=True)


if __name__ == "__main__":
    main()
