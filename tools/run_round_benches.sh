#!/bin/bash
# GPU box: the bench lines committed under profiles/<label>_bench_*.json -- the default line as the driver runs it, the
# other BASELINE configs, and the train-step mode.  usage: bash tools/run_round_benches.sh r03
label=${1:-r03}
out=gpurun_out/bench_$label
mkdir -p $out
python bench.py > $out/${label}_bench_cfg2.json 2> $out/cfg2.err || exit 1
for c in cfg1 cfg3 cfg4 cfg5; do
  python bench.py --config $c --no-cpu-baseline > $out/${label}_bench_$c.json 2> $out/$c.err || exit 1
done
python bench.py --mode train-step --steps 200 --warmup 20 > $out/${label}_bench_train_step_1gpu.json 2> $out/train.err || exit 1
python bench.py --mode train-step --config cfg4 --steps 200 --warmup 20 > $out/${label}_bench_train_step_1gpu_cfg4.json 2> $out/train4.err || exit 1
tail -c 600 $out/${label}_bench_cfg2.json
