#!/bin/bash
# GPU box: tools/ubench/chat_restream plain (timing) and under the FETCH_SIZE PMC pass (how much of the re-read left the L2)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
$root/tools/ubench/chat_restream 200
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/crs_FETCH -- $root/tools/ubench/chat_restream 200 > /dev/null 2> $root/gpurun_out/crs_FETCH.log
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$root/gpurun_out/crs_FETCH/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_restream" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
cases = [(256, 0), (256, 128), (512, 128), (128, 128), (64, 128), (0, 128), (256, 64)]
timed = rows[1::2]          # every case: a 20-trip warm-up, then the 200-trip dispatch
for (kb, tile), (d, v) in zip(cases, timed):
    req = (kb + tile) * 1024 * 256 * 200 / 1e9
    print(f"block {kb:3d} KB + tile {tile:3d} KB: requested {req:7.2f} GB, FETCH_SIZE x2 {v * 1024 * 2 / 1e9:7.2f} GB per dispatch "
          f"-> {v * 1024 * 2 / 1e9 / max(req, 1e-9):.2f} of the requested bytes crossed the fabric")
PY
