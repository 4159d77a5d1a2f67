#!/bin/bash
# GPU box: the L2's memory-side (fabric) READ request counters per kernel of one bench config, each --pmc set its own run:
# how many requests, how many of them 32-byte ones, how many went to DRAM -- to tell real bytes from the request-size class
# FETCH_SIZE assumes (FETCH_SIZE = RDREQ x 64 B; a 128-byte request is tallied as 64: MI355X_MICROARCH.md, HBM).
# usage: [CFG=cfg5] bash tools/run_tcc_ea.sh   -> gpurun_out/tccea_<cfg>_*/ + digest on stdout
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
cfg=${CFG:-cfg5}
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*" | sort -u | tr '\n' ' '; echo
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"; do
  out=$root/gpurun_out/tccea_${cfg}_$i
  rocprofv3 --pmc $set --output-format csv -d $out -- python3 $root/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify ${EXTRA} > /dev/null 2> $out.log
  i=$((i+1))
done
CFG=$cfg python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = os.environ["CFG"]
tot = {}
for f in glob.glob(f"{root}/gpurun_out/tccea_{cfg}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ge2e" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ge2e::", "")[:28]
        tot.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for name, d in sorted(tot.items()):
    t = {k: max(v) for k, v in d.items()}
    if t.get("TCC_EA0_RDREQ_sum", 0) < 1e5:
        continue
    rd, r32 = t.get("TCC_EA0_RDREQ_sum", 0), t.get("TCC_EA0_RDREQ_32B_sum", 0)
    print(f"{cfg} {name:28s} RDREQ {rd:.4g}  32B {r32:.4g} ({r32 / max(rd, 1):.3f})  DRAM {t.get('TCC_EA0_RDREQ_DRAM_sum', 0):.4g}  "
          f"FETCH_SIZE {t.get('FETCH_SIZE', 0) / 1048576:.4g} GiB (KiB counter)  L2 hit {t.get('TCC_HIT_sum', 0) / max(t.get('TCC_HIT_sum', 0) + t.get('TCC_MISS_sum', 0), 1):.3f}  TCC_READ {t.get('TCC_READ_sum', 0):.4g}")
PY
