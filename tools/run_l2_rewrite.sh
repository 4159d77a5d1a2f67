#!/bin/bash
# GPU box: tools/ubench/l2_rewrite under the two HBM-traffic PMC passes -> per-dispatch WRITE_SIZE / FETCH_SIZE
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/l2rw_$c -- $root/tools/ubench/l2_rewrite ${1:-64} ${2:-64} > $root/gpurun_out/l2rw_$c.log 2>&1
done
cat $root/gpurun_out/l2rw_WRITE_SIZE.log
python3 - <<PY
import csv, glob
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    rows = []
    for f in glob.glob("$root/gpurun_out/l2rw_%s/*/*_counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "k_rewrite" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:40], float(r["Counter_Value"])))
    for d, n, v in sorted(rows):
        print(f"{c} dispatch {d:3d} {n:40s} {v / 1024:10.1f} MB (KiB counter)")
PY
