#!/bin/bash
# GPU box: fabric-side traffic of one implementation at the bench's cfg2 default: TCC hit/miss and the request
# counters behind FETCH_SIZE / WRITE_SIZE, each --pmc set its own run (program directly after "--").
# usage: IMPL=team bash tools/run_tcc_team.sh   (writes gpurun_out/tcc_<impl>_*)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
impl=${IMPL:-team}
out=$root/gpurun_out
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --output-format csv -d $out/tcc_${impl}_$i -- python3 $root/bench.py --impl $impl --steps 3 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> $out/tcc_${impl}_$i.log
  i=$((i+1))
done
IMPL=$impl python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
impl = os.environ["IMPL"]
tot = {}
for f in glob.glob(f"{root}/gpurun_out/tcc_{impl}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ge2e" not in r["Kernel_Name"] or "fused_split" in r["Kernel_Name"] and impl != "fused_split" or int(r["Grid_Size"]) < 100000:
            continue
        tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
t = {k: max(v) for k, v in sorted(tot.items())}
print(impl, {k: round(v) for k, v in t.items()})
B, alg = 4096, 4096 * 1310720
if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
    f, w = t["FETCH_SIZE"] * 1024 * 2, t["WRITE_SIZE"] * 1024
    print(f"{impl}: FETCH_SIZE x2 = {f/1e9:.3f} GB, WRITE_SIZE = {w/1e9:.3f} GB per {B}-batch launch; "
          f"(fetch + write) / algorithmic = {(f + w) / alg:.3f}")
PY
