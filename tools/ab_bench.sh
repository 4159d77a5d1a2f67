#!/bin/bash
# same-box A/B of cfg2 throughput: the round-2 library (libge2e_hip_exp_head.so, built from the round-2 tree) against
# the current one, interleaved; gpurun_out/r3/ab.txt
mkdir -p gpurun_out/r3
pk=speaker_embedding_ge2e_loss_amd
for rep in 1 2; do
  for lib in ${BASE_LIB:-libge2e_hip_exp_head.so} libge2e_hip.so; do
    v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4))")
    echo "$lib $v" | tee -a gpurun_out/r3/ab.txt
  done
done
