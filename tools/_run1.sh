cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_double_backward.py -x -q -m gpu -k "force_block" 2>&1 | tail -3
E3K_FORCE_MATERIALIZE=0 python -m pytest tests/test_gpu_double_backward.py -x -q -m gpu -k "force_block_equals" 2>&1 | tail -2
for m in 1 0; do E3K_FORCE_MATERIALIZE=$m python bench.py --config energy_force --graph-fresh --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('materialize $m', d['ms_per_step'], d['value'])"; done
TRACE_ARGS="--graph-fresh --config energy_force" bash tools/trace_graph.sh 2>&1 | tail -32
