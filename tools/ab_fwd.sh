#!/bin/bash
# same-box interleaved A/B of the forward-only launch (cfg2, dE = NULL): BASE_LIB (default libge2e_hip_exp_base.so) against the
# current library; appends to gpurun_out/${ROUND:-r5}/ab_fwd.txt.  EXTRA: more bench flags (e.g. "--config cfg3").
mkdir -p gpurun_out/${ROUND:-r5}
pk=speaker_embedding_ge2e_loss_amd
for rep in 1 2 3; do
  for lib in ${BASE_LIB:-libge2e_hip_exp_base.so} libge2e_hip.so; do
    v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --forward-only --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${EXTRA:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4), d['verify']['ok'])")
    echo "$lib $v" | tee -a gpurun_out/${ROUND:-r5}/ab_fwd.txt
  done
done
