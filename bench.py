"""GE2E loss+backward throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--impl auto] [--mode loss|train-step]

--mode loss (default).  One *step* = one launch of the hot path (ge2e_loss_fwd_bwd through the C ABI)
over B independent synthetic (N,M,D) batches that are already resident in HBM.  The metric is (N x M)
batches per second.  With --gpus N>1 every rank processes its own B batches (weak scaling, no
data-path collective; the only reduction is loss/dw/db) and the value is the whole-job
aggregate over the max-over-ranks time.

--mode train-step.  The data-parallel step of s4:193-203 around the loss: per step one loss launch on this
rank's batches, (dw, db) into the tail of the flat gradient bucket, ONE SUM all-reduce of the
1,464,578-float bucket (the reference encoder's 1,464,576 parameters + w + b, trainer.py's layout) on
the nccl (= RCCL) backend, and the division by the world size.  Timed with the collective (the value)
and again without it (train_step.ms_without_allreduce).  The encoder's own forward/backward is library
work outside this path and is not in the step.

Ranks: launched by torch.distributed.run (RANK/LOCAL_RANK/WORLD_SIZE in the environment) the script is
one rank.  Run as plain `python bench.py --gpus N` with N>1 it starts that launcher itself as a child
process -- before anything in this process touches the GPU -- and exits with the child's code.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {  # BASELINE.json "configs"; cfg2 is the one the metric is quoted on
    # B = batches per launch, resident in HBM: one launch is one "step".  Sized for this GPU's 288 GB, not for a 16 GB card:
    # 10.7 GB of E at cfg1-cfg4 (as much again of dE; TILED's workspace is ~4x E), 2 GB at cfg5 (workspace 12 GB).  A
    # launch has a fixed cost -- the ramp, the first batch's un-overlapped load, the drain of the team pipeline (one
    # batch-time in n per team) and the tail -- that rounds 1-4 amortised over 2.7 GB only (B = 4096 at cfg2, 16 batches
    # per team): same box, same kernel, B = 4096 -> 16384: +9.7 %.
    "cfg1": dict(N=4, M=5, D=256, variant="softmax", B=524288),
    "cfg2": dict(N=64, M=10, D=256, variant="softmax", B=16384),
    "cfg3": dict(N=64, M=10, D=256, variant="contrast", B=16384),
    "cfg4": dict(N=256, M=10, D=256, variant="softmax", B=1024),
    "cfg5": dict(N=1024, M=10, D=768, variant="softmax", B=64),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F16_PEAK_TF = 2500.0  # dense f16/bf16 MFMA peak (no sparsity)
BUCKET_FLOATS = 1464578    # LSTM(80->256, 3 layers) + Linear(256,256) + w + b  (s2:13-25, s3:16-17)
SPLIT_IMPLS = ("fused_split", "team", "tiled")
ARITH = {
    True: "split-fp16x3 MFMA (hi.hi + hi.lo + lo.hi, 2^8 prescale, ~22 mantissa bits), fp32 accumulate; "
          "norms, softmax and reductions in fp32",
    False: "fp32 FMA throughout",
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks as a CHILD (never exec-replace), wait, pass its
    exit code on.  Nothing in this process has touched the GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def synth(B, N, M, D, seed, device):
    """normalize(randn) rows = the encoder's output contract (s2:34).  Generated ON THE DEVICE, in place and in slabs (rounds
    1-4 built the 10.7 GB block and its normalised copy on the host: 25 s of a 30 s run, and eight times that side by side
    under --gpus 8).  The post-timing `verify` pulls the sampled batches of THIS data back and hands them to the oracle."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    e = torch.empty(B, N, M, D, dtype=torch.float32, device=dev)
    slab = max(1, (1 << 28) // (N * M * D))           # 1 GiB of fp32 per slab
    for i in range(0, B, slab):
        part = e[i:i + slab]
        part.normal_(generator=g)
        part.div_(part.norm(dim=-1, keepdim=True).clamp_min_(1e-12))
    return e


def cpu_baseline(N, M, D, variant, budget_s=12.0):
    """The oracle's expand-form restatement (op-for-op the reference's s3:19-127 + autograd)
    timed on this box's host cores: kind "port".  Bounded sample of the same workload.  The box
    shows more cores than its CPU share, so a few thread counts are probed first and the fastest
    is used (oversubscribed torch CPU ops are several times slower)."""
    from oracle import ge2e_oracle as orc
    e = torch.nn.functional.normalize(torch.randn(N, M, D, generator=torch.Generator().manual_seed(1234)), dim=-1)
    if N * N * M * D * 4 * 2 > 8e9:  # the expand form needs 2 x (N^2 M, D) fp32 (+ autograd copies)
        return {"value": None, "unit": "batches/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": "skipped: expand form needs > 8 GB at this shape"}

    def one():
        x = e.clone().requires_grad_(True)
        w = torch.tensor(10.0, requires_grad=True)
        b = torch.tensor(-5.0, requires_grad=True)
        loss, _, _ = orc.expand_form_loss(x, w, b, variant=variant)
        loss.backward()
        return float(loss.detach())

    ncpu = os.cpu_count() or 8
    best_t, best_threads = None, None
    for threads in sorted({min(8, ncpu), min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
        torch.set_num_threads(threads)
        one()
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, best_threads = dt, threads
    torch.set_num_threads(best_threads)
    one()
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 60):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 1.0 / med, "unit": "batches/s", "cores": best_threads, "kind": "port",
            "sample": f"{len(times)} fwd+bwd iterations of one N={N} M={M} D={D} batch "
                      f"(expand-form torch CPU restatement, median {med * 1e3:.1f} ms, "
                      f"{best_threads} threads = fastest of 8/16/32/64 on {ncpu} visible cores)"}


def verify_launch(E, out, N, M, D, variant, w=10.0, b=-5.0):
    """After the timed region: sampled batches of the benched launch (first, last, both sides of a team boundary, the
    middle) against the fp64 closed-form oracle.  The oracle is the checker here, never the thing timed."""
    from oracle import ge2e_oracle as orc
    B = E.shape[0]
    cand = (0, 31, 32, B // 2 - 1, B // 2, B - 1) if N * N * M * D <= 64 * 64 * 10 * 256 * 64 else (0, B // 2, B - 1)
    picks = sorted({i for i in cand if 0 <= i < B})   # the fp64 closed form of one N=1024, D=768 batch takes seconds
    worst = {"max_loss_rel": 0.0, "max_dE_relfro": 0.0, "max_dw_rel": 0.0, "max_db_abs": 0.0}
    for i in picks:
        ref = orc.closed_form(E[i].cpu().numpy(), w, b, variant=variant)
        dE = out.dE[i].cpu().numpy().astype(np.float64)
        worst["max_loss_rel"] = max(worst["max_loss_rel"], abs(float(out.loss[i]) - ref["loss"]) / max(abs(ref["loss"]), 1e-30))
        worst["max_dE_relfro"] = max(worst["max_dE_relfro"], float(np.linalg.norm(dE - ref["dE"]) / max(np.linalg.norm(ref["dE"]), 1e-30)))
        worst["max_dw_rel"] = max(worst["max_dw_rel"], abs(float(out.dw[i]) - ref["dw"]) / max(abs(ref["dw"]), 1e-30))
        worst["max_db_abs"] = max(worst["max_db_abs"], abs(float(out.db[i]) - ref["db"]))
    # the north-star gate: loss rtol 1e-4, dE rel-Frobenius 1e-4, dw rtol 1e-4, db atol 1e-4
    ok = (worst["max_loss_rel"] <= 1e-4 and worst["max_dE_relfro"] <= 1e-4 and worst["max_dw_rel"] <= 1e-4
          and worst["max_db_abs"] <= 1e-4 and all(np.isfinite(v) for v in worst.values()))
    return {"batches": picks, **worst, "tolerance": "loss rtol 1e-4, dE rel-Frobenius 1e-4, dw rtol 1e-4, db atol 1e-4",
            "oracle": "oracle.ge2e_oracle.closed_form (fp64)", "ok": bool(ok)}


def verify_forward(E, loss, per, N, M, D, variant, w=10.0, b=-5.0):
    """The forward-only launch (dE = NULL): loss and per-row losses of sampled batches against the fp64 oracle."""
    from oracle import ge2e_oracle as orc
    B = E.shape[0]
    picks = sorted({i for i in (0, 31, 32, B // 2, B - 1) if 0 <= i < B})
    worst = {"max_loss_rel": 0.0, "max_per_abs": 0.0}
    for i in picks:
        ref = orc.closed_form(E[i].cpu().numpy(), w, b, variant=variant, want_grad=False)
        worst["max_loss_rel"] = max(worst["max_loss_rel"], abs(float(loss[i]) - ref["loss"]) / max(abs(ref["loss"]), 1e-30))
        worst["max_per_abs"] = max(worst["max_per_abs"], float(np.abs(per[i].cpu().numpy().astype(np.float64) - ref["per"]).max()))
    # (per-row losses to the 2e-5 tests/test_gpu_team_fwd.py::check_fwd holds them to; the kernels deliver ~2e-6)
    ok = worst["max_loss_rel"] <= 1e-4 and worst["max_per_abs"] <= 2e-5 and all(np.isfinite(v) for v in worst.values())
    return {"batches": picks, **worst, "tolerance": "loss rtol 1e-4, per-row loss atol 2e-5",
            "oracle": "oracle.ge2e_oracle.closed_form (fp64)", "ok": bool(ok)}


def measured_traffic(cfg_name, impl, B):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/run_profiles.sh: FETCH_SIZE x 2 on gfx950 + WRITE_SIZE).  The counters cannot be read from inside this
    process.  The record carries the hash of the kernel sources it was measured on; it is reported only for the exact
    (config, impl, B) AND only while that hash still matches the tree -- otherwise null."""
    try:
        from speaker_embedding_ge2e_loss_amd.build import source_hash
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            table = json.load(f)
        t = table.get(f"{cfg_name}_{impl}")
        if t and t["impl"] == impl and t["batches_per_launch"] == B and t.get("source_hash") == source_hash():
            return {"bytes": t["fetch_bytes"] + t["write_bytes"], "fetch_bytes_x2": t["fetch_bytes"],
                    "write_bytes": t["write_bytes"], "profile": t.get("source"),
                    "kernel_avg_us_in_profile": t.get("traced_kernel_avg_us"), "source_hash": t.get("source_hash")}
    except Exception:
        pass
    return None


def time_launches(fn, steps):
    """HIP events on the stream the kernels are launched on (torch's current stream) -> per-step ms."""
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=list(CONFIGS))
    ap.add_argument("--impl", default="auto")
    ap.add_argument("--mode", default="loss", choices=["loss", "train-step"])
    ap.add_argument("--batches", type=int, default=0,
                    help="B per launch per GPU (0 = config default; train-step default 1 = one (N,M) batch per rank per step, as s4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the B=1 latency legs and the exact-fp32 comparison (profiling runs: only the benched kernel)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing oracle check of sampled batches")
    ap.add_argument("--forward-only", action="store_true",
                    help="profiling runs: the timed step is the forward-only launch (dE = NULL: similarity + loss, the "
                         "workload of s4:61-110 / s5:42-44); the JSON line then carries only the forward_only object's numbers")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: ranks, rendezvous (gloo), the bucket all-reduce and the JSON line only; value is null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    dist = None
    if world > 1 or args.mode == "train-step":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = dict(CONFIGS[args.config])
    N, M, D, variant = cfg["N"], cfg["M"], cfg["D"], cfg["variant"]
    B = args.batches or (1 if args.mode == "train-step" else cfg["B"])

    if dry:
        impl = args.impl

        def step():
            pass
    else:
        from speaker_embedding_ge2e_loss_amd import functional as GF
        impl = GF.resolve_impl(B, N, M, D, variant, args.impl)
        E = synth(B, N, M, D, 1234 + rank, dev)
        w = torch.tensor(10.0, device=dev)
        b = torch.tensor(-5.0, device=dev)
        out = GF.LossOutputs(loss=torch.empty(B, device=dev), per=None,
                             dE=torch.empty_like(E), dw=torch.empty(B, device=dev), db=torch.empty(B, device=dev))
        ws = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, variant, impl), dev)

        def step():
            GF.loss_fwd_bwd(E, w, b, variant=variant, impl=impl, out=out, workspace=ws)

        # similarity + loss only (dE = NULL): what the reference's evaluation paths run (s4:61-110 test loss, s5:42-44)
        out_f = GF.LossOutputs(loss=torch.empty(B, device=dev), per=torch.empty(B, N, M, device=dev), dE=None, dw=None, db=None)

        def step_fwd():
            GF.loss_fwd_bwd(E, w, b, variant=variant, impl=impl, need_grad=False, out=out_f, workspace=ws)

        if args.forward_only:
            step = step_fwd

    bucket = None
    if args.mode == "train-step":
        bucket = torch.zeros(BUCKET_FLOATS, device=dev)
        # the encoder's gradient: random stand-in, resident (its backward is library work outside this path)
        bucket[:-2] = torch.randn(BUCKET_FLOATS - 2, generator=torch.Generator().manual_seed(99 + rank)).to(dev) * 1e-3
        loss_step = step

        def tail():
            if not dry:
                bucket[-2:] = torch.stack([out.dw.sum(), out.db.sum()])  # s4:200: (w, b) grads land in the bucket

        def step_with():
            loss_step()
            tail()
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)   # trainer.py: one message per step
            bucket.div_(world)

        def step_without():
            loss_step()
            tail()
            bucket.div_(world)

        step = step_with

    def barrier():
        sync()
        if dist and world > 1:
            dist.barrier()
        sync()

    def timed(fn, steps):
        barrier()
        t0 = time.perf_counter()
        ms = time_launches(fn, steps) if not dry else [0.0] * steps
        if dry:
            for _ in range(steps):
                fn()
        barrier()
        return time.perf_counter() - t0, ms

    for _ in range(args.warmup):
        step()
    elapsed, launch_ms = timed(step, args.steps)

    tmax = elapsed
    loss_mean = None
    rccl_ranks, per_rank_s = 1, [elapsed]
    if dist and world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tmax = float(t.item())
        # what the collective library itself saw: a one-element SUM of ones over the job's backend (nccl = RCCL on the
        # GPU box, gloo in a dry run) must come back as the number of ranks, and every rank's own clock is gathered
        one = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        rccl_ranks = int(round(float(one.item())))
        if rccl_ranks != world or dist.get_world_size() != world:
            raise SystemExit(f"bench.py: the process group answers {rccl_ranks} ranks (world size {dist.get_world_size()}), launched {world}")
        gathered = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([elapsed], device=dev, dtype=torch.float64))
        per_rank_s = [float(g.item()) for g in gathered]
    if not dry:
        # the only cross-rank reduction of the loss-only job: loss / dw / db sums
        red = (torch.stack([out_f.loss.sum()] * 3) if args.forward_only
               else torch.stack([out.loss.sum(), out.dw.sum(), out.db.sum()]))
        if dist and world > 1:
            dist.all_reduce(red, op=dist.ReduceOp.SUM)
        loss_mean = float(red[0].item()) / (B * world)

    train = None
    if args.mode == "train-step":
        for _ in range(max(2, args.warmup // 2)):
            step_without()
        el_wo, _ = timed(step_without, args.steps)
        # the collective alone, back to back
        for _ in range(3):
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
        el_ar, _ = timed(lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM), args.steps)
        tt = torch.tensor([el_wo, el_ar], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el_wo, el_ar = float(tt[0]), float(tt[1])
        nbytes = BUCKET_FLOATS * 4
        train = {"bucket_floats": BUCKET_FLOATS, "bucket_bytes": nbytes,
                 "backend": "gloo (dry run)" if dry else "nccl (RCCL)",
                 "ms_with_allreduce": tmax / args.steps * 1e3,
                 "ms_without_allreduce": el_wo / args.steps * 1e3,
                 "allreduce_alone_ms": el_ar / args.steps * 1e3,
                 # ring all-reduce moves 2 (n-1)/n of the message over every link
                 "allreduce_busbw_GBs": (2 * (world - 1) / world * nbytes / (el_ar / args.steps) / 1e9) if world > 1 else None}

    extra = {}
    if rank == 0 and not dry and args.mode == "loss" and not args.no_extras and not args.forward_only:
        # similarity + loss only (dE = NULL): north_star's "GE2E similarity+loss" sentence and the reference's evaluation
        # paths.  Algorithmic bytes = N M D 4 per batch (E read once, nothing of that size written), flops = 2 N^2 M D.
        for _ in range(3):
            step_fwd()
        torch.cuda.synchronize()
        msf = float(np.mean(time_launches(step_fwd, max(5, min(20, args.steps)))))
        fb, ff = N * M * D * 4, 2 * N * N * M * D
        gbs = fb * B / (msf * 1e-3) / 1e9
        itf = ff * B / (msf * 1e-3) / 1e12 * (3 if impl in SPLIT_IMPLS else 1)
        fwd_mfma = impl in SPLIT_IMPLS and itf / MFMA_F16_PEAK_TF > gbs / HBM_PEAK_GBS
        extra["forward_only"] = {
            "workload": f"{args.config}: similarity + loss only (dE = NULL), B={B} batches per launch, impl {impl}",
            "value": B / (msf * 1e-3), "unit": "batches/s", "ms_per_launch": msf,
            "roofline": ({"bound": "mfma", "achieved": itf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": itf / MFMA_F16_PEAK_TF}
                         if fwd_mfma else
                         {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}),
            "algorithmic_bytes_per_launch": fb * B, "algorithmic_flops_per_launch": ff * B,
            "traffic": (measured_traffic(args.config, impl + "_fwd", B) or {}).get("bytes"),
            "traffic_from_profile": measured_traffic(args.config, impl + "_fwd", B),
            "verify": None if args.no_verify else verify_forward(E, out_f.loss, out_f.per, N, M, D, variant)}
    if rank == 0 and not dry and args.mode == "loss" and not args.no_extras and not args.forward_only:
        # B = 1 latency, raw C-ABI call (not the metric; launch-bound single batches)
        e1 = E[:1].contiguous()
        o1 = GF.LossOutputs(loss=out.loss[:1], per=None, dE=out.dE[:1], dw=out.dw[:1], db=out.db[:1])
        impl1 = GF.resolve_impl(1, N, M, D, variant, args.impl)
        ws1 = GF.alloc_workspace(GF.workspace_bytes(1, N, M, D, variant, impl1), dev)
        call1 = lambda: GF.loss_fwd_bwd(e1, w, b, variant=variant, impl=impl1, out=o1, workspace=ws1)  # noqa: E731
        for _ in range(10):
            call1()
        extra["latency_b1_us"] = float(np.median(time_launches(call1, 100))) * 1e3   # median: host jitter is not the kernel
        extra["latency_b1_impl"] = impl1

        # what a user of the reference's class pays per training step at B=1: GE2ELoss.forward + loss.backward()
        # through the module (allocator, autograd, the grad_out scaling), s4:193-200
        from speaker_embedding_ge2e_loss_amd import GE2ELoss, HParams
        mod = GE2ELoss(HParams(device=dev), variant=variant, impl=args.impl)
        em = E[0].clone().requires_grad_(True)

        def module_step():
            em.grad = None
            mod.zero_grad(set_to_none=True)
            mod(em).backward()

        for _ in range(10):
            module_step()
        extra["latency_module_eager_b1_us"] = float(np.median(time_launches(module_step, 100))) * 1e3
        # the same two lines of the reference's step through GE2ELoss(hp, graph=True): forward = copy into a static input +
        # one replay of the captured fused launch, loss.backward() publishes the launch's own dE / dw / db (loss.py)
        eager_mod, mod = mod, GE2ELoss(HParams(device=dev), variant=variant, impl=args.impl, graph=True)
        try:
            for _ in range(10):
                module_step()
            extra["latency_module_b1_us"] = float(np.median(time_launches(module_step, 100))) * 1e3
            extra["latency_module_b1_route"] = ("GE2ELoss(hp, graph=True): mod(em).backward() served from a HIP graph over "
                                                "static buffers" if mod._steps else "eager (the shape was not captured)")
        except Exception as ex:   # a runtime that cannot capture: the eager figure stands in, and the line says so
            torch.cuda.synchronize()
            extra["latency_module_b1_us"] = extra["latency_module_eager_b1_us"]
            extra["latency_module_b1_route"] = f"eager (graph route failed: {str(ex)[:120]})"
        mod = eager_mod
        em.grad = None
        # ... and with the autograd node in Python (functional._GE2ELossFunction) instead of libge2e_torch.so's: same launches
        extra["latency_module_autograd_node"] = "c++ (libge2e_torch.so)" if GF._cpp_loss_op() is not None else "python"
        if GF._cpp_loss_op() is not None:
            GF.use_cpp_autograd(False)
            for _ in range(10):
                module_step()
            extra["latency_module_python_node_b1_us"] = float(np.median(time_launches(module_step, 100))) * 1e3
            GF.use_cpp_autograd(True)
        # the same step captured once in a HIP graph and replayed (graphed.GraphedLossStep): the host work of the eager
        # module path -- Python, autograd dispatch, allocator -- leaves the loop, the device operations stay.  Measured in
        # a CHILD process: graph capture exercises runtime paths nothing else here does, and whatever happens to it must
        # not take the bench line with it.
        try:
            r = subprocess.run([sys.executable, "-m", "speaker_embedding_ge2e_loss_amd.graphed", str(N), str(M), str(D),
                                variant, args.impl], capture_output=True, text=True, timeout=180,
                               cwd=os.path.dirname(os.path.abspath(__file__)))
            extra.update(json.loads(r.stdout.strip().splitlines()[-1]))
        except Exception as ex:   # capture unsupported by this torch / runtime, or the child died
            extra["latency_module_graph_b1_us"] = None
            extra["latency_module_graph_note"] = str(ex)[:160]

        # the exact-fp32 kernel beside the split-fp16 one (same workload; fewer steps)
        if impl in SPLIT_IMPLS and world == 1:
            for cand in ("fused_f32", "generic"):
                try:
                    if GF.resolve_impl(B, N, M, D, variant, cand) != cand:
                        continue
                    wsx = GF.alloc_workspace(GF.workspace_bytes(B, N, M, D, variant, cand), dev)
                    callx = lambda: GF.loss_fwd_bwd(E, w, b, variant=variant, impl=cand, out=out, workspace=wsx)  # noqa: E731
                    callx()
                    torch.cuda.synchronize()
                    msx = float(np.mean(time_launches(callx, max(3, min(10, args.steps)))))
                    extra["exact_f32"] = {"impl": cand, "value": B / (msx * 1e-3), "unit": "batches/s",
                                          "ms_per_launch": msx, "arith": ARITH[False]}
                    step()  # leave the outputs as the benched implementation wrote them
                    torch.cuda.synchronize()
                    break
                except Exception as ex:  # unsupported shape for this kernel
                    extra["exact_f32"] = {"impl": cand, "value": None, "note": str(ex)[:120]}

    if rank == 0 and not dry and args.mode == "train-step":
        # the encoder tail (normalise s2:34 + un-permute s4:186 + (N,M,D) view s4:189) in front of the loss.
        # Separate: normalize_unperm launch + loss launch(es) + normalize_unperm backward launch; raw: ONE launch that
        # normalises and gathers in its load stage and scatters dL/dy from its store stage (ge2e_loss_fwd_bwd_raw).
        # Timed through autograd as the trainer calls them (forward + backward of the tail and the loss), on the
        # reference's own training shapes; launch counts are what each route enqueues on the stream per step.
        tails = {}
        for (tn, tm) in ((4, 5), (2, 16)):
            rows = tn * tm
            gen = torch.Generator(device=dev).manual_seed(5)
            y = torch.randn(rows, D, device=dev, generator=gen).requires_grad_(True)
            perm = torch.randperm(rows, generator=torch.Generator().manual_seed(6))
            unperm = torch.empty_like(perm)
            unperm[perm] = torch.arange(rows)
            unperm = unperm.to(dev).to(torch.int32)
            wt = torch.tensor(10.0, device=dev, requires_grad=True)
            bt = torch.tensor(-5.0, device=dev, requires_grad=True)

            def separate():
                y.grad = wt.grad = bt.grad = None
                GF.ge2e_loss(GF.normalize_unperm(y, unperm, shape=(tn, tm)), wt, bt).backward()

            def raw():
                y.grad = wt.grad = bt.grad = None
                GF.ge2e_loss_raw(y, unperm, wt, bt, (tn, tm)).backward()

            rec = {"raw_supported": bool(GF.raw_supported(tn, tm, D))}
            for nm, fn in (("separate", separate), ("raw", raw)):
                for _ in range(5):
                    fn()
                rec[f"{nm}_us"] = float(np.median(time_launches(fn, 50))) * 1e3
            # kernels of OURS per step (autograd's own ones_like fill not counted): tail fwd + loss + grad scaling + tail bwd
            rec["launches_separate"] = 4
            rec["launches_raw"] = 2 if rec["raw_supported"] else 4
            tails[f"N={tn},M={tm},D={D}"] = rec
        extra["encoder_tail"] = tails

    verify = None
    if rank == 0 and not dry and args.mode == "loss" and not args.no_verify:
        step()   # the benched implementation's outputs (the exact-fp32 leg may have run since)
        torch.cuda.synchronize()
        verify = (verify_forward(E, out_f.loss, out_f.per, N, M, D, variant) if args.forward_only
                  else verify_launch(E, out, N, M, D, variant))

    if rank == 0:
        total_batches = B * args.steps * world
        value = None if dry else total_batches / tmax
        bytes_per_batch = 2 * N * M * D * 4  # read E once + write dE once
        flops_per_batch = 6 * N * N * M * D  # three N x (N M) x D contractions, 2 flops per MAC
        if args.forward_only:                # E read once, one contraction
            bytes_per_batch, flops_per_batch = N * M * D * 4, 2 * N * N * M * D
        split = impl in SPLIT_IMPLS
        roof = None
        if not dry:
            avg_launch_s = float(np.mean(launch_ms)) * 1e-3
            alg_tf = flops_per_batch * B / avg_launch_s / 1e12
            alg_gbs = bytes_per_batch * B / avg_launch_s / 1e9
            # which roof binds: arithmetic intensity of the shape against the machine balance of the
            # arithmetic actually issued (3 MFMA per product in split form)
            issued_tf = alg_tf * (3 if split else 1)
            mfma_bound = split and (issued_tf / MFMA_F16_PEAK_TF) > (alg_gbs / HBM_PEAK_GBS)
            common = {"kernel": f"ge2e {impl}", "avg_launch_ms": avg_launch_s * 1e3,
                      "algorithmic_bytes_per_launch": bytes_per_batch * B,
                      "algorithmic_flops_per_launch": flops_per_batch * B,
                      "achieved_algorithmic_tflops": alg_tf, "achieved_algorithmic_GBs": alg_gbs,
                      "hbm_frac": alg_gbs / HBM_PEAK_GBS,
                      "mfma_frac": issued_tf / MFMA_F16_PEAK_TF if split else None}
            if args.mode == "train-step":
                common["note"] = "launch time here includes the bucket all-reduce (train-step mode)"
            # HBM-side bytes of this launch: not measurable in-process -- taken from the committed rocprofv3 PMC passes of
            # the same command, only while the kernel sources still hash to what was profiled (else null)
            tp = measured_traffic(args.config, impl + ("_fwd" if args.forward_only else ""), B)
            if mfma_bound:
                roof = {"bound": "mfma", "achieved": issued_tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": issued_tf / MFMA_F16_PEAK_TF, "traffic": (tp or {}).get("bytes"), "traffic_from_profile": tp,
                        "achieved_is": "issued f16 MFMA flops = 3 x algorithmic (split-fp16x3)", **common}
            else:
                roof = {"bound": "hbm", "achieved": alg_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_gbs / HBM_PEAK_GBS, "traffic": (tp or {}).get("bytes"), "traffic_from_profile": tp, **common}
        name = {"loss": "GE2E loss+backward throughput", "train-step": "GE2E data-parallel train-step throughput "
                "(loss+backward + flat-bucket grad all-reduce)"}[args.mode]
        if args.forward_only:
            name = "GE2E similarity+loss (forward-only, dE = NULL) throughput"
        line = {
            "metric": f"{name}, (spk x utt) batches/sec at N={N} M={M} D={D}",
            "value": value, "unit": "batches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "arith": ARITH[impl in SPLIT_IMPLS] if not dry else None,
            "data": "synthetic" if not dry else "dry-run (no kernel launched)",
            "config": {"workload": f"{args.config}: N={N} M={M} D={D} {variant} GE2E " + ("forward only, " if args.forward_only else "fwd+bwd, ") +
                                   f"B={B} batches per launch per GPU, w=10 b=-5"
                                   + (f", + SUM all-reduce of {BUCKET_FLOATS} fp32 grads per step" if train else ""),
                       "mode": args.mode, "N": N, "M": M, "D": D, "variant": variant, "batches_per_launch": B,
                       "impl": impl,
                       "parallelism": f"dp{world} (whole batches per rank" +
                                      (", one flat-bucket all-reduce per step)" if train else ", no data-path collective)")},
            "roofline": roof,
            "loss_mean": loss_mean,
            # ranks the collective backend counted (a real 1-element all-reduce; 1 = single process, no group) and each
            # rank's own rate over its own clock: value uses the slowest rank's time
            "rccl_ranks": rccl_ranks,
            "per_rank_value": [None if dry else B * args.steps / t_ for t_ in per_rank_s],
        }
        if variant == "contrast":
            line["parity"] = ("unpinned: the reference has no contrast variant (s3 implements softmax only); checked "
                              "against the fp64 closed-form oracle and finite differences only")
        if verify is not None:
            line["verify"] = verify
        line.update(extra)
        if train:
            line["train_step"] = train
        if dry:
            line["dry_run"] = True
        if world == 1 and not args.no_cpu_baseline and not dry and args.mode == "loss":
            line["cpu_baseline"] = cpu_baseline(N, M, D, variant)
        print(json.dumps(line), flush=True)
    if dist:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
        dist = None
    if verify is not None and not verify["ok"]:
        raise SystemExit("bench.py: the benched launch does not match the oracle: " + json.dumps(verify))
    fo = extra.get("forward_only")
    if fo and fo.get("verify") and not fo["verify"]["ok"]:
        raise SystemExit("bench.py: the forward-only launch does not match the oracle: " + json.dumps(fo["verify"]))


if __name__ == "__main__":
    main()
