"""Experiment: one TILED call over B batches against the same work as consecutive calls over chunks of the batches (does a
chunk's intermediate data stay in the 256 MB Infinity Cache between its kernels?).  python tools/exp_chunk.py [--config cfg4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg4")
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    N, M, D, variant, B = cfg["N"], cfg["M"], cfg["D"], cfg["variant"], cfg["B"]
    dev = torch.device("cuda:0")
    e = bench.synth(B, N, M, D, 1234, dev)
    w, b = torch.tensor(10.0, device=dev), torch.tensor(-5.0, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    loss, dw, db, dE = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32), torch.empty_like(e)
    for chunk in (B, B // 2, B // 4, B // 8, B // 16):
        if chunk < 1:
            continue
        ws = GF.alloc_workspace(GF.workspace_bytes(chunk, N, M, D, variant, "tiled"), dev)

        def run():
            for c0 in range(0, B, chunk):
                sl = slice(c0, c0 + chunk)
                out = GF.LossOutputs(loss=loss[sl], per=None, dE=dE[sl], dw=dw[sl], db=db[sl])
                GF.loss_fwd_bwd(e[sl], w, b, variant=variant, impl="tiled", out=out, workspace=ws)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            run()
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 10
        print(f"{args.config}: {B // chunk:3d} calls of {chunk:4d} batches: {ms:.3f} ms per {B} batches = {B / ms * 1e3:.0f} batches/s", flush=True)


if __name__ == "__main__":
    main()
