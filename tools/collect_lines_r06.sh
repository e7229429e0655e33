#!/bin/bash
# Section 5 of tools/collect_profiles_r06.sh alone: the bench lines without a profiler (boxes of this pool are shared and
# noisy; re-run when a session's lines are outliers, then python3 tools/summarize_profiles_r06.py)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline"
uptime > $OUT/lines_uptime.txt
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --lmax 3 > $OUT/bench_lmax3.json 2>/dev/null
for b in 32 64 128; do $B --batch $b > $OUT/bench_b$b.json 2>/dev/null; done      # (replayed, pipelined preparation: the fixed mode of this workload)
$B --eager > $OUT/bench_eager.json 2>/dev/null                                       # rounds 1-5's default: the eager four-stream step
for b in 32 64 128; do $B --eager --batch $b > $OUT/bench_eager_b$b.json 2>/dev/null; done
E3K_BENCH_PREP_PIPELINE=0 $B > $OUT/bench_one_graph.json 2>/dev/null                 # the step with its batch preparation inside the graph
$B --batch 512 > $OUT/bench_b512.json 2>/dev/null
$B --loader > $OUT/bench_loader.json 2>/dev/null
for c in energy_force diffusion diffusion_CA; do $B --config $c > $OUT/bench_$c.json 2>/dev/null; done
$B --launch auto > $OUT/bench_launch_auto.json 2>/dev/null
$B --bonds clustered > $OUT/bench_clustered.json 2>/dev/null
$B --config energy_force --bonds clustered > $OUT/bench_energy_force_clustered.json 2>/dev/null
uptime >> $OUT/lines_uptime.txt
for f in default eager one_graph lmax3 b32 b64 b128 eager_b32 eager_b64 eager_b128 b512 loader launch_auto energy_force clustered energy_force_clustered diffusion diffusion_CA; do python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/bench_$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d.get('host_busy_ms_per_step'))"; done
cat $OUT/lines_uptime.txt
