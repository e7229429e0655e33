#!/bin/bash
timeout 900 python3 -m pytest tests/test_gpu_prepare.py -x -q -m gpu 2>&1 | tail -30
mkdir -p gpurun_out/e4
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e4/default.json 2> gpurun_out/e4/default.err
E3K_BENCH_PREP_PIPELINE=0 $B > gpurun_out/e4/noprep.json 2> gpurun_out/e4/noprep.err
$B --config energy_force > gpurun_out/e4/force.json 2> gpurun_out/e4/force.err
$B --config diffusion > gpurun_out/e4/diffusion.json 2> gpurun_out/e4/diffusion.err
$B --batch 32 > gpurun_out/e4/b32.json 2> gpurun_out/e4/b32.err
for f in gpurun_out/e4/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['final_loss'])" 2>&1)"; done
tail -n 3 gpurun_out/e4/*.err
