#!/bin/bash
# full GPU suite + the default bench line + the ordered trace of a replayed step
timeout 2300 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -60 > gpurun_out/gpu_suite.log
cat gpurun_out/gpu_suite.log | tail -3
mkdir -p gpurun_out/e6
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e6/default.json 2> gpurun_out/e6/default.err
E3K_BENCH_PREP_PIPELINE=0 $B > gpurun_out/e6/noprep.json 2> gpurun_out/e6/noprep.err
for f in gpurun_out/e6/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['final_loss'])" 2>&1)"; done
E3K_BENCH_PREP_PIPELINE=0 TRACE_NAME=trq bash tools/r06_trace.sh
