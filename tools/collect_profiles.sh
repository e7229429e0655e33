#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: collects the rocprofv3 evidence of a round into
# gpurun_out/<tag>/ (kernel-trace stats of the default bench; separate --pmc passes; no trace domains beside --pmc).
#   tools/collect_profiles.sh r02 && python3 tools/summarize_profiles.py r02 r02     (the second step runs anywhere)
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/probe_fetch $OUT/probe_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/probe_fetch -- python3 tools/pmc_probe.py > $OUT/pmc_probe.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/probe_write -- python3 tools/pmc_probe.py > /dev/null 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 2500 $OUT/bench_default.json
