#!/bin/bash
# quick check: operator + model tests that exercise GEMMs / layers, then the bench line and the trace
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_prepare.py tests/test_gpu_trained_parity.py -q -m gpu -x 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_gpu_model.py -q -m gpu -x -k "bench_path or stack or bucketed or thresholds or energy_forward" 2>&1 | tail -3
mkdir -p gpurun_out/e7
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e7/default.json 2> gpurun_out/e7/default.err
E3K_BENCH_PREP_PIPELINE=0 $B > gpurun_out/e7/noprep.json 2> gpurun_out/e7/noprep.err
for f in gpurun_out/e7/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['final_loss'])" 2>&1)"; done
E3K_BENCH_PREP_PIPELINE=0 TRACE_NAME=trq bash tools/r06_trace.sh
