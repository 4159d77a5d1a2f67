#!/bin/bash
# GPU box: SQ instruction-mix / busy counters of the two cfg2 kernels (each --pmc set its own run).
# usage: [CFG=cfg5] [IMPLS="auto team"] bash tools/run_sq_counters.sh   (writes gpurun_out/sq_<impl>_<set>/ and prints a digest,
# per kernel name: the largest value over the launches)   EXTRA="--forward-only": the forward-only launch
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for impl in ${IMPLS:-auto team}; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
             "SQ_INSTS_VMEM SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
             "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
    out=$root/gpurun_out/sq_${impl}_$i
    rocprofv3 --pmc $set --output-format csv -d $out -- python3 $root/bench.py --config ${CFG:-cfg2} --steps 3 --warmup 2 --no-cpu-baseline --no-extras --no-verify --impl $impl ${EXTRA:-} > /dev/null 2> $out.log
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for impl in os.environ.get("IMPLS", "auto team").split():
    tot = {}
    for f in glob.glob(f"{root}/gpurun_out/sq_{impl}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ge2e" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 90000:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ge2e::", "")[:40]
            tot.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for name, d in sorted(tot.items()):
        print(impl, name, {k: round(max(v)) for k, v in sorted(d.items())})
PY
