#!/bin/bash
# same-box timing of experiment libraries (timing-only builds may compute garbage: --no-verify)
# usage: bash tools/ab_variants.sh cfg5 libA.so libB.so ...
cfg=$1; shift
pk=speaker_embedding_ge2e_loss_amd
for rep in 1 2; do
  for lib in "$@"; do
    v=$(GE2E_HIP_LIB=$PWD/$pk/$lib python bench.py --config $cfg --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['impl'], round(d['value']), round(d['roofline']['frac'],4))")
    echo "$cfg $lib $v"
  done
done
