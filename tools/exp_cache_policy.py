"""Timing experiment: cache-policy bits (0 default, 2 nt, 1 sc0, 16 sc1) on the three E streams of fused_split.
Usage (GPU box): python tools/exp_cache_policy.py"""
import ctypes as C
import itertools
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import _lib, build  # noqa: E402


def run(defs, B=4096, iters=8):
    tag = "_".join(f"{k[-2:]}{v}" for k, v in defs.items())
    lib_path = os.path.join(build.PKG_DIR, f"libge2e_hip_exp{tag}.so")
    cmd = [build._hipcc(), "-O3", "-std=c++17", f"--offload-arch={build.ARCH}", "-fPIC", "-shared",
           f"-I{build.INCLUDE}", "-o", lib_path] + [f"-D{k}={v}" for k, v in defs.items()] + build.sources()
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
        assert lib.ge2e_loss_fwd_bwd(E.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, 0, im,
                                     loss.data_ptr(), None, dE.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                     ws.data_ptr(), ws.numel(), st) == 0
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    os.remove(lib_path)
    ms = e0.elapsed_time(e1) / iters
    print(f"{defs}: {ms:.3f} ms per launch, {B / ms * 1e3:,.0f} batches/s", flush=True)


if __name__ == "__main__":
    for e1, e2, e3, de in [(0, 0, 2, 2), (2, 0, 2, 2), (0, 2, 2, 2), (2, 2, 2, 2), (0, 0, 0, 2), (0, 0, 2, 0), (16, 16, 2, 2)]:
        run({"GE2E_AUX_E1": e1, "GE2E_AUX_E2": e2, "GE2E_AUX_E3": e3, "GE2E_AUX_DE": de})
