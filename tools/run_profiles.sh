#!/bin/bash
# GPU box: collect the rocprofv3 evidence for profiles/ -- kernel trace + the two PMC passes, each its own run, program
# directly after "--" -- for the default bench line (cfg2, AUTO) and whatever else is named.
# usage: bash tools/run_profiles.sh <round-label> "<cfg>:<impl> ..."   (default "cfg2:auto")
# writes gpurun_out/prof_<label>_<cfg>_<impl>/{summary.txt,kernel_stats.csv} and updates profiles/traffic.json in the
# box's copy (copied to gpurun_out/traffic.json so that it comes back).
# FWD=1: the forward-only launch (dE = NULL, `bench.py --forward-only`) is the profiled step; directories and the
# traffic.json key get the suffix _fwd.
set +e
label=${1:-r02}
what=${2:-cfg2:auto}
extra=""; suf=""
if [ "${FWD:-0}" = "1" ]; then extra="--forward-only"; suf="_fwd"; fi
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
for item in $what; do
  cfg=${item%%:*}; impl=${item##*:}
  out=$root/gpurun_out/prof_${label}_${cfg}_${impl}${suf}
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras --impl $impl $extra > $out/bench_trace.json 2> $out/trace.log
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-extras --impl $impl $extra > /dev/null 2> $out/fetch.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --config $cfg --steps 3 --warmup 2 --no-cpu-baseline --no-extras --impl $impl $extra > /dev/null 2> $out/write.log
  python3 $root/tools/summarize_rocprof.py $out $out/summary.txt "round ${label}, --config ${cfg} --impl ${impl} ${extra}"
  cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || true
  resolved=$(python3 -c "import json,sys; print(json.loads(open('$out/bench_trace.json').read().strip().splitlines()[-1])['config']['impl'])")
  B=$(python3 -c "import json,sys; print(json.loads(open('$out/bench_trace.json').read().strip().splitlines()[-1])['config']['batches_per_launch'])")
  python3 $root/tools/record_traffic.py $out $cfg ${resolved}${suf} $B "profiles/${label}_${cfg}_${resolved}${suf}_rocprof.txt"
  rm -rf $out/trace $out/fetch $out/write
done
cp $root/profiles/traffic.json $root/gpurun_out/traffic.json
