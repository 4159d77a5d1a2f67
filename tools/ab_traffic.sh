#!/bin/bash
# GPU box: same-box A/B of PREBUILT experiment libraries (speaker_embedding_ge2e_loss_amd/<name>.so, built here with
# build.build_variant) -- two interleaved timed rounds with the oracle check of the benched launch, then FETCH_SIZE / WRITE_SIZE
# of each under rocprofv3 (separate passes).   usage: [CFG=cfg2] [IMPL=auto] [EXTRA=--forward-only] bash tools/ab_traffic.sh libA.so libB.so ...
root=${GRAFT_REPO_ROOT:-/root/repo}
pk=$root/speaker_embedding_ge2e_loss_amd
cfg=${CFG:-cfg2}; impl=${IMPL:-auto}
cd $root
for rep in 1 2 3; do
  for lib in "$@"; do
    v=$(GE2E_HIP_LIB=$pk/$lib python3 bench.py --config $cfg --impl $impl --steps 20 --warmup 5 --no-extras --no-cpu-baseline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4), (d.get('verify') or {}).get('ok'), (d.get('verify') or {}).get('max_dE_relfro'))")
    echo "$cfg $lib $v"
  done
done
cd /tmp && export TMPDIR=/tmp
k=0
for lib in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GE2E_HIP_LIB=$pk/$lib rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/abt_${k}_$c -- python3 $root/bench.py --config $cfg --impl $impl --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify $EXTRA > /dev/null 2> $root/gpurun_out/abt_${k}_$c.log
  done
  K=$k LIB=$lib python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); k = os.environ["K"]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for f in glob.glob(f"{root}/gpurun_out/abt_{k}_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ge2e" in r["Kernel_Name"]:
                n = r["Kernel_Name"].split("(")[0][-28:]
                per[n] = max(per.get(n, 0), float(r["Counter_Value"]))
    out[c] = {n: v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9 for n, v in per.items() if v > 1000}
print(f"[{os.environ['LIB']}] fetch x2 GB {({n: round(v, 3) for n, v in out['FETCH_SIZE'].items()})}  write GB {({n: round(v, 3) for n, v in out['WRITE_SIZE'].items()})}")
PY
  k=$((k+1))
done
