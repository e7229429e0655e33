#!/bin/bash
# bench.py --gpus 2 on ONE GPU over gloo (the N > 1 code path: replayed step + flat all-reduce + replica checksums)
mkdir -p gpurun_out/e9
E3K_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --batch 64 > gpurun_out/e9/two_ranks.json 2> gpurun_out/e9/two_ranks.err
echo "rc $?"
E3K_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 6 --warmup 3 --no-cpu-baseline --batch 64 --eager > gpurun_out/e9/two_ranks_eager.json 2> gpurun_out/e9/two_ranks_eager.err
echo "rc $?"
for f in gpurun_out/e9/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['config']; print(d['n_gpus'], d['ms_per_step'], c['ranks'], c['launch'][:30], c['all_reduce'], c['replica_parameter_checksums'], c['replay_error'], c['final_loss'])" 2>&1)"; done
grep -i "error\|Traceback" gpurun_out/e9/*.err | head -5
