#!/bin/bash
# GPU box: candidate K-contiguous swizzles of the tiled core (experiment libraries) -- cfg5 rate with the oracle check, and the
# LDS bank-conflict counters of the similarity kernel.  usage: bash tools/kc_variants.sh lib1.so lib2.so ...
root=${GRAFT_REPO_ROOT:-/root/repo}
pk=$root/speaker_embedding_ge2e_loss_amd
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  v=$(GE2E_HIP_LIB=$pk/$lib python3 $root/bench.py --config cfg5 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['frac'],4), (d.get('verify') or {}).get('ok'))")
  out=$root/gpurun_out/kc_$lib
  rm -rf $out
  GE2E_HIP_LIB=$pk/$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $out -- python3 $root/bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-verify > /dev/null 2> $out.log
  c=$(python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("$out/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for k in ("tiled_sim", "tiled_ge", "tiled_gc"):
            if k in n:
                tot.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print("  ".join(f"{k}: conflict {max(d.get('SQ_LDS_BANK_CONFLICT',[0]))/1e6:.0f} M of {max(d.get('SQ_LDS_IDX_ACTIVE',[0]))/1e6:.0f} M" for k, d in sorted(tot.items())))
PY
)
  echo "$lib  cfg5 $v  |  $c"
done
