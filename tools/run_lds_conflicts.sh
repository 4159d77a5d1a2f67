#!/bin/bash
# GPU box: LDS bank-conflict share of one implementation at cfg2 (usage: bash tools/run_lds_conflicts.sh <impl>)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
impl=${1:-team}
out=$root/gpurun_out/lds_${impl}
rm -rf $out
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --impl $impl > /dev/null 2> $out.log
python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$out/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ge2e" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
t = {k: max(v) for k, v in tot.items()}
print("$impl", {k: round(v) for k, v in t.items()}, "conflict share of LDS-active cycles: %.1f %%" % (100 * t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]))
PY
