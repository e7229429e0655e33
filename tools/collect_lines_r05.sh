#!/bin/bash
# Section 5 of tools/collect_profiles_r05.sh alone: the bench lines without a profiler (boxes of this pool are shared and
# noisy; re-run when a session's lines are outliers, then python3 tools/summarize_profiles_r05.py)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline"
uptime > $OUT/lines_uptime.txt
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --lmax 3 > $OUT/bench_lmax3.json 2>/dev/null
for b in 32 64 128; do $B --batch $b > $OUT/bench_b$b.json 2>/dev/null; done      # (eager: the fixed mode of this workload)
$B --batch 512 > $OUT/bench_b512.json 2>/dev/null
$B --loader > $OUT/bench_loader.json 2>/dev/null
$B --graph --batch 32 > $OUT/bench_graph_b32.json 2>/dev/null
for b in 32 64 128; do $B --graph-fresh --batch $b > $OUT/bench_graphfresh_b$b.json 2>/dev/null; done
for c in energy_force diffusion diffusion_CA; do $B --config $c > $OUT/bench_$c.json 2>/dev/null; done
$B --graph-fresh > $OUT/bench_graphfresh_b256.json 2>/dev/null
$B --launch auto > $OUT/bench_launch_auto.json 2>/dev/null
E3K_TP_TABLE_PACKED=0 $B > $OUT/bench_four_row_table.json 2>/dev/null      # round 4 form of the in-kernel table, same box
E3K_RADIAL_TABLE_KEYED=1 $B --config diffusion > $OUT/bench_diffusion_keyed_tables.json 2>/dev/null
$B --config energy_force --graph-fresh > $OUT/bench_energy_force_graphfresh.json 2>/dev/null
E3K_FORCE_BLOCK=0 $B --config energy_force --graph-fresh > $OUT/bench_energy_force_composed_graphfresh.json 2>/dev/null      # rounds 1-3's path, same box
$B --bonds clustered > $OUT/bench_clustered.json 2>/dev/null
$B --config energy_force --graph-fresh --bonds clustered > $OUT/bench_energy_force_clustered.json 2>/dev/null
uptime >> $OUT/lines_uptime.txt
for f in default lmax3 b32 b64 b128 b512 loader graph_b32 graphfresh_b32 graphfresh_b64 graphfresh_b128 graphfresh_b256 launch_auto four_row_table diffusion_keyed_tables energy_force energy_force_graphfresh energy_force_composed_graphfresh clustered energy_force_clustered diffusion diffusion_CA; do python3 -c "
import json,sys
d=json.loads([l for l in open('$OUT/bench_$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d.get('host_busy_ms_per_step'))"; done
cat $OUT/lines_uptime.txt
