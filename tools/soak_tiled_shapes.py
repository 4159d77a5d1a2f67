"""One-off soak: random shapes in TILED's territory (N up to 600, M 2..12, D a multiple of 64 up to 512) with enough
batches that the 256 x 256 tiles run -- DMA-fed or register-staged as K allows, ragged row / slot / column edges, k_gc cut
along its rows or not, the fused similarity + row kernel for N <= 256 -- against the exact-fp32 VALU kernel (GENERIC) on the
same inputs; forward only on every third shape.  Usage: python tools/soak_tiled_shapes.py [seed] [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from speaker_embedding_ge2e_loss_amd import functional as GF  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = done = 0
    while done < n:
        N = int(rng.integers(130, 601)); M = int(rng.integers(2, 13)); D = int(rng.choice([64, 128, 192, 256, 320, 384, 512]))
        if rng.random() < 0.4:
            N = int(rng.choice([160, 192, 256, 288, 320, 384, 512]))     # K = N a multiple of 32: gE takes the DMA form
        NM = N * M
        tiles = -(-NM // 256)
        B = int(-(-192 // tiles)) + int(rng.integers(0, 6))               # >= 192 row tiles: the big tiles run
        if B * NM * max(N, D) * 4 > 1.5e9 or NM < 256:
            continue
        variant = "contrast" if rng.random() < 0.25 else "softmax"
        try:
            if GF.resolve_impl(B, N, M, D, variant, "tiled") != "tiled":
                continue
        except Exception:
            continue
        done += 1
        g = torch.Generator(device=dev).manual_seed(int(rng.integers(1 << 30)))
        e = torch.randn(B, N, M, D, device=dev, generator=g)
        if rng.random() < 0.5:
            e = e / e.norm(dim=-1, keepdim=True)
        w = torch.tensor(float(rng.uniform(2, 14)), device=dev); b = torch.tensor(float(rng.uniform(-6, 3)), device=dev)
        fwd_only = done % 3 == 0
        a = GF.loss_fwd_bwd(e, w, b, variant=variant, impl="tiled", need_grad=not fwd_only, need_per=True)
        r = GF.loss_fwd_bwd(e, w, b, variant=variant, impl="generic", need_grad=not fwd_only, need_per=True)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(a.loss).all()) and bool(torch.isfinite(a.per).all())
        lerr = float(((a.loss - r.loss).abs() / (r.loss.abs() + 1e-3)).max())
        perr = float((a.per - r.per).abs().max() / (1.0 + float(r.per.abs().max())))
        derr = 0.0
        if not fwd_only:
            ok = ok and bool(torch.isfinite(a.dE).all())
            num = (a.dE - r.dE).flatten(1).norm(dim=1); den = r.dE.flatten(1).norm(dim=1) + 1e-12
            derr = float((num / den).max())
        # contrast: an argmax tie within fp32 resolution may resolve differently in the two kernels (one row's gradient)
        lim = 2e-5 if variant == "softmax" else 5e-2
        flag = "" if (ok and lerr < 2e-5 and perr < 2e-5 and derr < lim) else "   <-- CHECK"
        bad += bool(flag)
        print(f"B={B:3d} N={N:3d} M={M:2d} D={D:3d} {variant:8s} {'fwd' if fwd_only else 'f+b'} w={float(w):5.2f}: loss rel {lerr:.1e}  "
              f"per {perr:.1e}  dE rel-fro (worst batch) {derr:.1e}{flag}", flush=True)
    print(f"{done} shapes, {bad} to check")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
