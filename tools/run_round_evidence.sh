#!/bin/bash
# GPU box: everything a round commits under profiles/ in ONE call -- bench lines of every BASELINE config (+ train-step), the
# rocprofv3 kernel stats + HBM-traffic PMC passes of each, the forward-only evidence set, SQ counters and phase stamps of the
# metric kernel, clock / power, the GPU test log.  usage: bash tools/run_round_evidence.sh r05
# Progress goes to gpurun_out/<label>_evidence.log (a line per step: the call must not look hung).
label=${1:-r06}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd $root
log=gpurun_out/${label}_evidence.log
mkdir -p gpurun_out
echo "start $(date)" > $log
# the PMC passes first: they write profiles/traffic.json for THIS tree's source hash, which the bench lines below then carry
# as roofline.traffic (same call, same box: the kernel averages of the profiles and the lines' ms_per_step belong together)
bash tools/run_profiles.sh $label "cfg2:auto cfg1:auto cfg3:auto cfg4:auto cfg5:auto" >> $log 2>&1; echo "profiles done $(date)" >> $log
bash tools/run_fwd_profiles.sh $label auto >> $log 2>&1; echo "fwd profiles done $(date)" >> $log
bash tools/run_round_benches.sh $label >> $log 2>&1; echo "benches done $(date)" >> $log
cd $root
CFG=cfg2 IMPLS="team" bash tools/run_sq_counters.sh > gpurun_out/${label}_sq_counters_cfg2_team.txt 2>&1; echo "sq done $(date)" >> $log
cd $root
python3 tools/profile_phases.py --impl team --config cfg2 --batches 4096 > gpurun_out/${label}_team_cfg2_phase_stamps.txt 2>&1
GE2E_EXTRA_DEFS="-DGE2E_PROF_TID=256" python3 tools/profile_phases.py --impl team --config cfg2 --batches 4096 --lib libge2e_hip_prof4.so > gpurun_out/${label}_team_cfg2_phase_stamps_wave4.txt 2>&1
echo "stamps done $(date)" >> $log
python3 tools/team_load_sweep.py > gpurun_out/${label}_team_load_sweep.txt 2>&1; echo "sweep done $(date)" >> $log
python3 tools/soak_determinism.py > gpurun_out/${label}_soak_determinism.txt 2>&1; echo "soak done $(date)" >> $log
python3 -m pytest tests -m gpu -q > gpurun_out/${label}_gpu_tests.txt 2>&1; echo "tests done $(date)" >> $log
tail -3 gpurun_out/${label}_gpu_tests.txt
cat $log
