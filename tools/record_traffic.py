"""Write one record of profiles/traffic.json from a run_profiles.sh output dir.
usage: record_traffic.py <prof_dir> <config> <impl> <batches_per_launch> <source-label>

<prof_dir>/fetch and /write hold the separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes.  FETCH_SIZE is
doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); both are KiB.  The largest-grid ge2e kernel of the run is
the benched one.  The record carries build.source_hash() so bench.py can tell when it has gone stale."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from speaker_embedding_ge2e_loss_amd.build import source_hash  # noqa: E402


def counter(d, name):
    best = {}
    for f in glob.glob(os.path.join(d, name, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "ge2e" not in r["Kernel_Name"]:
                continue
            key = (r["Kernel_Name"], int(r["Grid_Size"]))
            best.setdefault(key, []).append(float(r["Counter_Value"]))
    if not best:
        return None, None
    # all ge2e kernels of one step, summed per launch (tiled runs several kernels per step)
    per_kernel = {k: max(v) for k, v in best.items()}
    return sum(per_kernel.values()), {f"{k[0][:60]} grid={k[1]}": v for k, v in per_kernel.items()}


def traced_launch_us(d):
    """average duration of the largest-grid ge2e kernel in the kernel-trace pass of the same command (kernel_stats.csv)"""
    try:
        rows = list(csv.DictReader(open(os.path.join(d, "kernel_stats.csv"))))
        rows = [r for r in rows if "ge2e" in r["Name"]]
        best = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        return float(best["AverageNs"]) / 1e3, best["Name"][:80]
    except Exception:
        return None, None


def main():
    d, cfg, impl, B, label = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    f, fk = counter(d, "fetch")
    w, wk = counter(d, "write")
    if f is None or w is None:
        sys.exit(f"no PMC csv under {d}")
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table[f"{cfg}_{impl}"] = {
        "impl": impl, "batches_per_launch": B,
        "fetch_bytes": int(f * 1024 * 2), "write_bytes": int(w * 1024),
        "per_kernel_fetch_KiB": fk, "per_kernel_write_KiB": wk,
        "source_hash": source_hash(),
        "traced_kernel_avg_us": traced_launch_us(d)[0], "traced_kernel": traced_launch_us(d)[1],
        "source": f"{label} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled per "
                  f"MI355X_MICROARCH.md HBM section)",
    }
    json.dump(table, open(path, "w"), indent=1)
    t = table[f"{cfg}_{impl}"]
    print(f"{cfg}_{impl}: fetch {t['fetch_bytes'] / 1e9:.3f} GB + write {t['write_bytes'] / 1e9:.3f} GB per launch, hash {t['source_hash']}")


if __name__ == "__main__":
    main()
