"""Development aid: build experiment variants of the library (extra -D definitions, same per-source flags as the
product build), and for each run a parity probe and the cfg2 bench in a child process.
Usage (GPU box): python tools/bench_variants.py "<defs of variant 1>" "<defs of variant 2>" ...   ("" = product flags)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_embedding_ge2e_loss_amd import build  # noqa: E402

impl = os.environ.get("IMPL", "team")
for k, defs in enumerate(sys.argv[1:] or [""]):
    lib = os.path.join(build.PKG_DIR, f"libge2e_hip_exp_v{k}.so")
    build.build_variant(lib, defs.split())
    env = dict(os.environ, GE2E_HIP_LIB=lib)
    probe = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_impl.py"), impl, "150", "64", "10", "256",
                            "softmax", "unit", "10", "-5"], env=env, capture_output=True, text=True, timeout=300)
    bad = [ln[:40] for ln in probe.stdout.splitlines() if ln.startswith("rep")]
    vals = []
    for _ in range(2):
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--impl", impl, "--no-cpu-baseline", "--steps", "15"],
                           env=env, capture_output=True, text=True, timeout=300)
        try:
            d = json.loads(b.stdout.strip().splitlines()[-1])
            vals.append((round(d["value"] / 1e6, 3), round(d["latency_b1_us"], 1)))
        except Exception as e:  # noqa: BLE001
            vals.append(("bench failed", str(e), b.stderr[-200:]))
    print(f"[{defs or 'product'}] parity {bad} | M batches/s, B=1 us: {vals}", flush=True)
    os.remove(lib)
