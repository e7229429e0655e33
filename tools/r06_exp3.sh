#!/bin/bash
mkdir -p gpurun_out/e3
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
for b in 384 512; do
$B --batch $b > gpurun_out/e3/replay_$b.json 2> gpurun_out/e3/replay_$b.err
$B --batch $b --eager > gpurun_out/e3/eager_$b.json 2> gpurun_out/e3/eager_$b.err
E3K_FWD_FORK=0 $B --batch $b --eager > gpurun_out/e3/eager1_$b.json 2> gpurun_out/e3/eager1_$b.err
done
for f in gpurun_out/e3/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'], d['host_busy_ms_per_step'], d['config']['launch'][:40])" 2>&1)"; done
