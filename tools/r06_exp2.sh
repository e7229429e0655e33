#!/bin/bash
mkdir -p gpurun_out/e2
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e2/default.json 2> gpurun_out/e2/default.err
$B --eager > gpurun_out/e2/eager.json 2> gpurun_out/e2/eager.err
E3K_DIST_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --batch 64 > gpurun_out/e2/two_ranks.json 2> gpurun_out/e2/two_ranks.err
E3K_DIST_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --batch 64 --eager > gpurun_out/e2/two_ranks_eager.json 2> gpurun_out/e2/two_ranks_eager.err
for f in gpurun_out/e2/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['launch'][:40], d['config'].get('replica_parameter_checksums'), d['config'].get('replay_error'))" 2>&1)"; done
tail -3 gpurun_out/e2/*.err
