"""One line per committed bench line of a round: value, roofline fraction, traffic ratio, forward-only leg, latencies.
Usage: python tools/bench_summary.py [r04]"""
import glob
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else "r04"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{label}_bench_*.json"))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        fo = d.get("forward_only") or {}
        alg = r.get("algorithmic_bytes_per_launch")
        tr = r.get("traffic")
        line = f"{os.path.basename(f):42s} {d['value']:14.1f} {d['unit']:10s} frac {r.get('frac', 0):.4f} ({r.get('bound', '-')})"
        if tr and alg:
            line += f"  traffic {tr / alg:.2f}x"
        if fo:
            line += f"  | forward only {fo['value']:.1f} ({(fo.get('roofline') or {}).get('frac', 0):.3f})"
        if d.get("latency_b1_us"):
            line += f"  | B=1 {d['latency_b1_us']:.1f} us, module {d.get('latency_module_b1_us', 0):.0f} us, graph {d.get('latency_module_graph_b1_us', 0):.0f} us"
        v = d.get("verify")
        if v:
            line += f"  | verify {'ok' if v.get('ok') else 'FAILED'}"
        print(line)


if __name__ == "__main__":
    main()
