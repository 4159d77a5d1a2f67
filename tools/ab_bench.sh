#!/bin/bash
# same-box A/B of cfg2 throughput: a base library (BASE_LIB, default libge2e_hip_exp_base.so: a copy of an earlier build, or tools/build_rev.sh <rev>) against
# the current one, interleaved; gpurun_out/${ROUND:-r4}/ab.txt
mkdir -p gpurun_out/${ROUND:-r4}
pk=speaker_embedding_ge2e_loss_amd
for rep in 1 2; do
  for lib in ${BASE_LIB:-libge2e_hip_exp_base.so} libge2e_hip.so; do
    v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4))")
    echo "$lib $v" | tee -a gpurun_out/${ROUND:-r4}/ab.txt
  done
done
