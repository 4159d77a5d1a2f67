#!/bin/bash
# After tools/run_round_evidence.sh <label> came back: copy what is judged from gpurun_out/ into profiles/.
label=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/traffic.json profiles/traffic.json
for c in cfg1:wave cfg2:team cfg3:team cfg4:tiled cfg5:tiled; do
  cfg=${c%%:*}; im=${c##*:}
  cp gpurun_out/prof_${label}_${cfg}_auto/summary.txt profiles/${label}_${cfg}_${im}_rocprof.txt
  cp gpurun_out/prof_${label}_${cfg}_auto/kernel_stats.csv profiles/${label}_${cfg}_${im}_kernel_stats.csv
done
f=gpurun_out/${label}_fwd
cp $f/rocprof_summary.txt profiles/${label}_cfg2_team_fwd_rocprof.txt
cp $f/kernel_stats.csv profiles/${label}_cfg2_team_fwd_kernel_stats.csv
cp $f/sq_counters.txt profiles/${label}_cfg2_fwd_sq_counters.txt
cp $f/phase_stamps_wave0.txt profiles/${label}_cfg2_fwd_phase_stamps.txt
cp $f/phase_stamps_wave4.txt profiles/${label}_cfg2_fwd_phase_stamps_wave4.txt
cp $f/clock_power.txt profiles/${label}_cfg2_fwd_clock_power.txt
cat $f/clock_power.txt $f/clock_power_fwd_bwd.txt > profiles/${label}_clock_power_under_bench.txt
for x in sq_counters_cfg2_team team_cfg2_phase_stamps team_cfg2_phase_stamps_wave4 team_load_sweep soak_determinism gpu_tests; do
  cp gpurun_out/${label}_$x.txt profiles/${label}_$x.txt
done
cp gpurun_out/bench_${label}/${label}_bench_*.json profiles/ 2>/dev/null
git status --short profiles | head -40
