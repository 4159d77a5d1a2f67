#!/bin/bash
# round-4 development loop on the GPU box: parity probe (softmax + contrast), interleaved A/B against
# libge2e_hip_exp_base.so, optionally the team / determinism tests.   usage: tools/dev4.sh [tag] [test]
tag=${1:-dev}; out=gpurun_out/r4; mkdir -p $out
python tools/check_team.py team 3 64 10 256 > $out/${tag}_check.txt 2>&1 || { tail -5 $out/${tag}_check.txt; exit 1; }
grep -E "^batch" $out/${tag}_check.txt | cut -c1-200
python tools/check_team.py team 2 64 10 256 contrast 2>&1 | grep -E "^batch" | cut -c1-200 | tee -a $out/${tag}_check.txt
ROUND=r4 tools/ab_bench.sh 2>&1 | sed "s/^/[$tag] /"
if [ "$2" = "test" ]; then python -m pytest tests/test_gpu_team.py tests/test_gpu_determinism.py -x -q > $out/${tag}_tests.txt 2>&1; tail -5 $out/${tag}_tests.txt; fi
