#!/bin/bash
# same-box A/B of one config's throughput: BASE_LIB (default libge2e_hip_exp_base.so) against the current library, interleaved.
# usage: bash tools/ab_bench_cfg.sh cfg4 [cfg5 ...]   -> gpurun_out/${ROUND:-r4}/ab_<cfg>.txt
mkdir -p gpurun_out/${ROUND:-r4}
pk=speaker_embedding_ge2e_loss_amd
for cfg in "$@"; do
  rm -f gpurun_out/${ROUND:-r4}/ab_$cfg.txt
  for rep in 1 2; do
    for lib in ${BASE_LIB:-libge2e_hip_exp_base.so} libge2e_hip.so; do
      v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --config $cfg --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4), (d.get('verify') or {}).get('ok'))")
      echo "$cfg $lib $v" | tee -a gpurun_out/${ROUND:-r4}/ab_$cfg.txt
    done
  done
done
