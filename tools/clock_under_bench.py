"""Engine clock and socket power while one bench configuration's loss launch keeps the queue full (rocm-smi sampled
between batches of launches).  Usage: python tools/clock_under_bench.py [--config cfg5] [--seconds 6]"""
import argparse
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "-d", "0"], capture_output=True, text=True).stdout
    keep = []
    for ln in out.splitlines():
        if "sclk" in ln or "Power (W)" in ln:
            keep.append(ln.split(":", 2)[-1].strip() if "Power" not in ln else "power " + ln.rsplit(":", 1)[-1].strip() + " W")
    return " | ".join(keep)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5")
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--forward-only", action="store_true", help="dE = NULL: similarity + loss only")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    N, M, D, variant, B = cfg["N"], cfg["M"], cfg["D"], cfg["variant"], cfg["B"]
    dev = torch.device("cuda:0")
    e = bench.synth(B, N, M, D, 1234, dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    print("idle:", smi(), flush=True)
    ng = not args.forward_only
    out = GF.loss_fwd_bwd(e, w, b, variant=variant, need_grad=ng)
    torch.cuda.synchronize()
    t_end = time.time() + args.seconds
    k = 0
    while time.time() < t_end:
        for _ in range(20):
            out = GF.loss_fwd_bwd(e, w, b, variant=variant, need_grad=ng, out=out)
        k += 1
        if k % 4 == 0:
            print(f"{args.config}{' forward-only' if args.forward_only else ''}:", smi(), flush=True)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
