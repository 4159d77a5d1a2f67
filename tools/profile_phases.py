"""Diagnostic: where does a fused-kernel workgroup spend its cycles?

Builds a SEPARATE library (libge2e_hip_prof.so, -DGE2E_PROFILE) whose kernels stamp
s_memtime at every phase boundary (wave 0's view, after the workgroup barrier), runs one
launch and prints cycles per batch per phase.  The stamps serialise issue at the phase
boundaries, so read the SHARES, not the total.  The shipped library contains no stamps.

    python tools/profile_phases.py [--impl fused_f32] [--config cfg2] [--batches 1024]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_embedding_ge2e_loss_amd import _lib, build  # noqa: E402
import bench  # noqa: E402

PHASES = {
    "fused_f32": ["s1 centroids", "s2a stage+stats", "s2b gemm1 X", "s2c softmax", "s2d gemm3 gC",
                  "finalize dc", "s3a stage", "s3b/c KJ+gemm2", "s3d epilogue", "-"],
    "team": ["P1(cur) rows -> images, centroid out; drain + signals", "wait 1 (centroids of all members)",
             "P2 centroid images -> LDS", "P3 X = CH.ET^T (+ dE stores of prev)", "P4 softmax in registers",
             "P5 KJP, gE tiles, scalars + barrier", "P6 G images", "P7 partial gC, rows request, publish, barrier",
             "wait 2 (partial gradients of prev)", "P8(prev) reduce -> KJ, dE complete"],
    "team": ["A1(cur) centroid out; drain + signals", "A2(cur) rows -> images; next rows requested", "W wait for both hand-offs",
              "X tail: slot scalars -> LDS, barrier", "S softmax, G images + barrier",
              "F1 KJ(prev) + barrier", "dE(prev) stores, scalars out",
              "GC partial gC + publish", "-", "requests: centroid fragments, partial gradients", "GE(prev)", "X contraction",
              "F2 member scalars, KJP(cur)", "-",
              "A1a (-DGE2E_PROF_A1) wait for the rows (= the memory queue) + speaker sum", "A1b centroid, stores, stage",
              "A1c barrier", "A1d k-group form stores", "A1e drain (vmcnt 0)"],
    # the three contractions of the tiled pipeline (diagnostic slots 0.., 8.., 16..): wave 0's cycles per BATCH, summed over
    # the batch's tiles (cfg5: 160 similarity tiles, 12 gC tiles, 120 gE tiles per batch)
    "tiled": ["sim: until the first stage landed", "sim: K loop", "sim: epilogue issued", "sim: stores drained", "sim loop: A blocks 0-3 x B (48 MFMAs) + reads of A blocks 4-7", "sim loop: own pieces landed", "sim loop: barrier", "sim loop: A blocks 4-7 x B (48 MFMAs) + pieces + next step's reads",
              "gc: until the first stage landed", "gc: K loop", "gc: epilogue issued", "gc: stores drained", "gc loop: A blocks 0-3 x B (48 MFMAs) + reads of A blocks 4-7", "gc loop: own pieces landed", "gc loop: barrier", "gc loop: A blocks 4-7 x B (48 MFMAs) + pieces + next step's reads",
              "ge: until the first stage landed", "ge: K loop", "ge: epilogue issued", "ge: stores drained", "ge loop: A blocks 0-3 x B (48 MFMAs) + reads of A blocks 4-7", "ge loop: own pieces landed", "ge loop: barrier", "ge loop: A blocks 4-7 x B (48 MFMAs) + pieces + next step's reads"],
    # the pipelined forward-only team kernel (csrc/ge2e_team_fwd.hip)
    "team_fwd": ["A1(cur) speaker sum -> centroid published; prev's fragments requested underneath", "X(prev) contraction -> XB",
                 "drain (vmcnt 0)", "barrier 1 + signal", "A2(cur) rows -> images; rows of n + 2 requested", "S(prev) softmax, loss",
                 "poll c1(cur) + barrier 2", "member scalars"],
    "fused_split": ["s1 centroids", "s2a stage", "s2b gemm1 X", "s2c softmax", "s2d KJP+gemm3 gC",
                    "finalize", "s3a stage+ring", "s3c gemm2", "s3d rows issue+barrier", "s3d epilogue body"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="fused_f32")
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--batches", type=int, default=1024)
    ap.add_argument("--forward-only", action="store_true", help="dE = NULL (similarity + loss only)")
    ap.add_argument("--no-wb", action="store_true", help="dw = db = NULL as well (the forward kernel's fast form of S)")
    ap.add_argument("--no-build", action="store_true", help="use the libge2e_hip_prof.so that is there (built off the GPU box)")
    ap.add_argument("--lib", default="libge2e_hip_prof.so", help="file name of the stamped library inside the package directory")
    args = ap.parse_args()
    lib_path = os.path.join(build.PKG_DIR, args.lib)
    if not (args.no_build and os.path.exists(lib_path)):
        build.build_variant(lib_path, ["-DGE2E_PROFILE"] + os.environ.get("GE2E_EXTRA_DEFS", "").split())
    lib = C.CDLL(lib_path)
    for name, (res, argt) in _lib.PROTOTYPES.items():
        getattr(lib, name).restype = res
        getattr(lib, name).argtypes = argt
    lib.ge2e_debug_set_prof.argtypes = [C.c_void_p]
    lib.ge2e_debug_set_prof.restype = None

    cfg = bench.CONFIGS[args.config]
    N, M, D, variant = cfg["N"], cfg["M"], cfg["D"], cfg["variant"]
    B = args.batches
    dev = torch.device("cuda:0")
    E = bench.synth(B, N, M, D, 1234, dev)
    w = torch.tensor(10.0, device=dev)
    b = torch.tensor(-5.0, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    loss, dw, db, dE = torch.empty(B, **f32), torch.empty(B, **f32), torch.empty(B, **f32), torch.empty_like(E)
    v, im = _lib.VARIANTS[variant], _lib.IMPLS[args.impl]
    ws = torch.empty(lib.ge2e_workspace_bytes(B, N, M, D, v, im) + 256, dtype=torch.uint8, device=dev)
    prof = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.ge2e_debug_set_prof(prof.data_ptr())

    def run():
        code = lib.ge2e_loss_fwd_bwd(E.data_ptr(), B, N, M, D, w.data_ptr(), b.data_ptr(), 1e-8, 1e-6, v, im,
                                     loss.data_ptr(), None, None if args.forward_only else dE.data_ptr(), None if args.no_wb else dw.data_ptr(), None if args.no_wb else db.data_ptr(),
                                     ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        assert code == 0, code

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    prof.zero_()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    run()
    t1.record()
    torch.cuda.synchronize()
    cyc = prof.cpu().numpy().astype(float) / B
    if args.impl in ("team", "team"):
        cyc /= 8.0          # eight workgroups stamp every batch; report one workgroup's timeline
    tot = cyc.sum()
    names = PHASES.get(args.impl + ("_fwd" if args.forward_only else ""), PHASES.get(args.impl, [f"phase {i}" for i in range(10)]))
    print(f"{args.impl} {args.config}{' forward-only' if args.forward_only else ''} B={B}: launch {t0.elapsed_time(t1):.3f} ms (stamped build); "
          f"{tot:.0f} cycles per batch per workgroup")
    for i, n in enumerate(names):
        if i < len(cyc) and cyc[i] > 0:
            print(f"  {n:34s} {cyc[i]:10.0f} cyc  {100 * cyc[i] / tot:5.1f} %")


if __name__ == "__main__":
    main()
