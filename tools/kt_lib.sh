#!/bin/bash
# GPU box: per-kernel average durations of one bench config with a given library.  usage: tools/kt_lib.sh <lib.so> [cfg] [tag]
lib=$1; cfg=${2:-cfg5}; tag=${3:-kt}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${ROUND:-r4}/$tag
cd /tmp && export TMPDIR=/tmp
GE2E_HIP_LIB=$root/speaker_embedding_ge2e_loss_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/bench.py --config $cfg --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-verify > $out.json 2> $out.err
python3 - <<PY
import csv, glob
for f in glob.glob("$out/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ge2e" in r["Name"]:
            print("$lib $cfg", r["Name"].split("(")[0].replace("void ge2e::", "")[:60], r["Calls"], round(float(r["AverageNs"]) / 1000, 1), "us")
PY
