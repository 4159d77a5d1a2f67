#!/bin/bash
# GPU box: does TEAM's exchange stay in the XCD's L2?  TCC hit/miss + fabric request counters, default build and a build
# whose E loads are nt (streaming, should not evict the exchange lines).  Each --pmc set is its own run.
# usage: bash tools/run_tcc_team.sh   (writes gpurun_out/tcc_*)
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
rocprofv3 -L > $out/counters_list.txt 2>&1
HIPCC=/opt/rocm/bin/hipcc
exp=$root/speaker_embedding_ge2e_loss_amd/libge2e_hip_exp_ent.so
$HIPCC -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -I$root/include -DGE2E_TEAM_E_AUX=2 -o $exp $root/speaker_embedding_ge2e_loss_amd/csrc/*.hip 2> $out/tcc_build.log || exit 1
for tag in def ent; do
  if [ $tag = ent ]; then export GE2E_HIP_LIB=$exp; else unset GE2E_HIP_LIB; fi
  python3 $root/bench.py --impl team --steps 10 --warmup 3 --no-cpu-baseline > $out/tcc_${tag}_bench.json 2> $out/tcc_${tag}_bench.err
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum"; do
    rocprofv3 --pmc $set --output-format csv -d $out/tcc_${tag}_$i -- python3 $root/bench.py --impl team --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> $out/tcc_${tag}_$i.log
    i=$((i+1))
  done
done
python3 - <<PY
import csv, glob, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for tag in ("def", "ent"):
    tot = {}
    for f in glob.glob(f"{root}/gpurun_out/tcc_{tag}_*/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "ge2e" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 100000:
                continue
            tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(tag, {k: round(max(v)) for k, v in sorted(tot.items())})
    try:
        print(tag, open(f"{root}/gpurun_out/tcc_{tag}_bench.json").read()[:300])
    except Exception as e:
        print(e)
PY
