#!/bin/bash
# GPU box, development aid: FETCH_SIZE / WRITE_SIZE of the cfg2 bench for experiment builds of the library.
# usage: bash tools/variant_traffic.sh "<defs of variant 1>" "<defs of variant 2>" ...
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
k=0
for defs in "$@"; do
  lib=/tmp/libge2e_var_$k.so
  python3 -c "
import sys; sys.path.insert(0, '$root')
from speaker_embedding_ge2e_loss_amd import build
build.build_variant('$lib', '$defs'.split())"
  for c in FETCH_SIZE WRITE_SIZE; do
    GE2E_HIP_LIB=$lib rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/vt_${k}_$c -- python3 $root/bench.py --impl ${IMPL:-team} --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $root/gpurun_out/vt_${k}_$c.json 2> $root/gpurun_out/vt_${k}_$c.log
  done
  K=$k DEFS="$defs" python3 - <<PY
import csv, glob, os, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); k = os.environ["K"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    best = 0
    for f in glob.glob(f"{root}/gpurun_out/vt_{k}_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ge2e" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 100000:
                best = max(best, float(r["Counter_Value"]))
    out[c] = best * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9
try:
    v = json.loads(open(f"{root}/gpurun_out/vt_{k}_WRITE_SIZE.json").read().strip().splitlines()[-1])["value"] / 1e6
except Exception:
    v = float("nan")
alg = 4096 * 1310720 / 1e9
print(f"[{os.environ['DEFS']}] fetch x2 {out['FETCH_SIZE']:.3f} GB  write {out['WRITE_SIZE']:.3f} GB  ratio {(out['FETCH_SIZE'] + out['WRITE_SIZE']) / alg:.3f}  ({v:.3f} M/s under the profiler)")
PY
  k=$((k+1))
done
