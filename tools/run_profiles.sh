#!/bin/bash
# GPU box: collect the rocprofv3 evidence for profiles/ (kernel trace + the two PMC passes, each its own run,
# program directly after "--"), for the default bench (AUTO -> fused_split at the default B=4096) and for --impl team.
# usage: bash tools/run_profiles.sh <round-label>   (writes under gpurun_out/prof_<label>/)
set +e
label=${1:-r01}
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for impl in auto team; do
  out=$root/gpurun_out/prof_${label}_${impl}
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --impl $impl > $out/bench_trace.json 2> $out/trace.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --impl $impl > /dev/null 2> $out/fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --impl $impl > /dev/null 2> $out/write.log
  python3 $root/tools/summarize_rocprof.py $out $out/summary.txt "round ${label}, --impl ${impl}, cfg2 default B"
  cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
done
