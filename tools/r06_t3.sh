#!/bin/bash
mkdir -p gpurun_out/e5
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e5/pipe.json 2> gpurun_out/e5/pipe.err
E3K_BENCH_PREP_PIPELINE=0 $B > gpurun_out/e5/noprep.json 2> gpurun_out/e5/noprep.err
for f in gpurun_out/e5/pipe.json gpurun_out/e5/noprep.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['final_loss'])" 2>&1)"; done
TRACE_NAME=trp bash tools/r06_trace.sh
