#!/bin/bash
# development loop on the GPU box: parity probe, team tests, short bench (gpurun_out/${ROUND:-r4}/dev_*.txt)
mkdir -p gpurun_out/${ROUND:-r4}
python tools/check_team.py team 3 64 10 256 > gpurun_out/${ROUND:-r4}/dev_check.txt 2>&1
tail -4 gpurun_out/${ROUND:-r4}/dev_check.txt
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${ROUND:-r4}/dev_bench.json 2> gpurun_out/${ROUND:-r4}/dev_bench.err
ROUND_DIR=gpurun_out/${ROUND:-r4} python - <<'PY'
import json, os
d=json.load(open(os.environ['ROUND_DIR'] + '/dev_bench.json'))
print('bench', d['config']['impl'], round(d['value']), 'batches/s  frac', round(d['roofline']['frac'],4), 'loss_mean', d['loss_mean'])
PY
if [ "$1" = "test" ]; then python -m pytest tests/test_gpu_team.py tests/test_gpu_determinism.py -x -q > gpurun_out/${ROUND:-r4}/dev_tests.txt 2>&1; tail -5 gpurun_out/${ROUND:-r4}/dev_tests.txt; fi
if [ "$1" = "prof" ]; then python tools/profile_phases.py --impl team --batches 4096 > gpurun_out/${ROUND:-r4}/dev_phases.txt 2>&1; tail -16 gpurun_out/${ROUND:-r4}/dev_phases.txt; fi
