#!/bin/bash
# same-box A/B of experiment libraries WITH the oracle check of the benched launch.  usage: bash tools/ab_libs.sh cfg2 libA.so libB.so ...
cfg=$1; shift
pk=speaker_embedding_ge2e_loss_amd
for rep in 1 2; do
  for lib in "$@"; do
    v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --config $cfg --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4), (d.get('verify') or {}).get('ok'), (d.get('verify') or {}).get('max_dE_relfro'))")
    echo "$cfg $lib $v"
  done
done
