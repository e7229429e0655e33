cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "knot_bins" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "bench_path" 2>&1 | tail -3
for b in uniform clustered uniform clustered; do python bench.py --no-cpu-baseline --bonds $b 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b', d['ms_per_step'], d['value'], d['config']['workload'][-90:])"; done
python bench.py --config energy_force --graph-fresh --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
