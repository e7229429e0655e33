#!/bin/bash
# Section 1 of tools/collect_profiles_r04.sh alone (the default command under rocprofv3 --kernel-trace --stats, and one step's
# launch sequence), for a tree whose launch schedule changed after the full collection
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04
mkdir -p $OUT; rm -rf $OUT/stats
export E3K_BENCH_AUTO=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 tools/step_kernels.py $(ls -t $OUT/stats/*kernel_trace.csv | head -1) $OUT/step_kernels.txt
head -1 $OUT/step_kernels.txt
