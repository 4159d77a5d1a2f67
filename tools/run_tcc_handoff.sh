#!/bin/bash
# GPU box: L2 behaviour of the team hand-off in isolation (tools/bench_team_handoff.py: every member publishes a
# payload with plain stores, all eight members read all eight payloads back with sc1 loads, 400 rounds):
# how many of the reads go to the fabric?  usage: bash tools/run_tcc_handoff.sh
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_READ_sum TCC_WRITE_sum"; do
  rocprofv3 --pmc $set --output-format csv -d $out/tcch_$i -- python3 $root/tools/bench_team_handoff.py > $out/tcch_$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
rows = {}
for f in sorted(glob.glob(f"{root}/gpurun_out/tcch_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "selftest_team" not in r["Kernel_Name"]:
            continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, c in sorted(rows.items(), key=lambda kv: int(kv[0])):
    print(d, {k: round(v) for k, v in sorted(c.items())})
PY
tail -4 $out/tcch_0.log
