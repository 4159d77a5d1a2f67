"""Condense rocprofv3 output dirs (kernel trace + FETCH_SIZE / WRITE_SIZE passes) into one
text summary for profiles/.  usage: summarize_rocprof.py <prof_dir> <out.txt> [label]

<prof_dir>/trace, /fetch, /write are the -d dirs of:
    rocprofv3 --kernel-trace --stats --output-format csv -d <prof_dir>/trace -- python3 bench.py ...
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <prof_dir>/fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <prof_dir>/write -- python3 bench.py ...
FETCH_SIZE is doubled before use: on gfx950 it reports 1/2 of the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section).  Both counters are in KiB.
"""
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    label = sys.argv[3] if len(sys.argv) > 3 else ""
    lines = [f"# rocprofv3 summary {label}".rstrip(), ""]
    bj = os.path.join(d, "bench_trace.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        lines += ["bench line of the traced run:",
                  json.dumps({k: b[k] for k in ("metric", "value", "unit", "ms_per_step", "config", "roofline")}), ""]
    trace = glob.glob(os.path.join(d, "trace", "*", "*_kernel_trace.csv"))
    if trace:
        rows = list(csv.DictReader(open(trace[0])))
        per = {}
        for r in rows:
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            per.setdefault((r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"),
                            r.get("LDS_Block_Size", "?")), []).append(dur)
        lines.append("kernel-trace (per kernel x grid): calls, avg us, min us, max us, VGPR, AGPR, LDS")
        for (k, g, vg, ag, lds), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"  {k:70s} grid={g:>7s} calls={len(v):3d} avg={sum(v)/len(v):10.1f} min={min(v):10.1f} "
                         f"max={max(v):10.1f} vgpr={vg} agpr={ag} lds={lds}")
        lines.append("")
    stats = glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        lines.append("--stats (kernel_stats.csv):")
        lines += ["  " + ln.rstrip()[:200] for ln in open(stats[0]).read().splitlines()[:6]]
        lines.append("")
    for name, mult in (("fetch", 2.0), ("write", 1.0)):
        f = glob.glob(os.path.join(d, name, "*", "*_counter_collection.csv"))
        if not f:
            continue
        per = {}
        for r in csv.DictReader(open(f[0])):
            per.setdefault((r["Kernel_Name"][:70], r["Grid_Size"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, g, c), v in per.items():
            if "ge2e" not in k:
                continue
            mx = max(v)
            lines.append(f"PMC {c}: {k} grid={g}: launches={len(v)} max={mx:.0f} KiB -> x{mult:g} = "
                         f"{mx * mult * 1024 / 1e9:.3f} GB per launch")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
