#!/bin/bash
# GPU box: the whole evidence set for the FORWARD-ONLY launch of the metric shape (dE = NULL: similarity + loss) --
# rocprofv3 kernel stats + the two HBM-traffic PMC passes (tools/run_profiles.sh, FWD=1), the SQ instruction-mix / wait
# counters, in-kernel phase stamps of the older and the younger wave of SIMD 0, and clock / power while the launch keeps
# the queue full.  usage: bash tools/run_fwd_profiles.sh <round-label> [impl]
# Everything lands under gpurun_out/<label>_fwd/ (copy what should be judged into profiles/).
set +e
label=${1:-r06}
impl=${2:-auto}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/${label}_fwd
mkdir -p $out
FWD=1 bash $root/tools/run_profiles.sh $label "cfg2:$impl" > $out/run_profiles.log 2>&1
cp $root/gpurun_out/prof_${label}_cfg2_${impl}_fwd/summary.txt $out/rocprof_summary.txt 2>/dev/null
cp $root/gpurun_out/prof_${label}_cfg2_${impl}_fwd/kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
CFG=cfg2 IMPLS="$impl" EXTRA="--forward-only" bash $root/tools/run_sq_counters.sh > $out/sq_counters.txt 2>&1
cd $root
python3 tools/profile_phases.py --impl team --config cfg2 --batches 4096 --forward-only > $out/phase_stamps_wave0.txt 2>&1
GE2E_EXTRA_DEFS="-DGE2E_PROF_TID=256" python3 tools/profile_phases.py --impl team --config cfg2 --batches 4096 --forward-only --lib libge2e_hip_prof4.so > $out/phase_stamps_wave4.txt 2>&1
python3 tools/clock_under_bench.py --config cfg2 --forward-only --seconds 6 > $out/clock_power.txt 2>&1
python3 tools/clock_under_bench.py --config cfg2 --seconds 6 > $out/clock_power_fwd_bwd.txt 2>&1
tail -n 30 $out/*.txt
