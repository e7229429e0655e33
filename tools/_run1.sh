cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "slope" 2>&1 | tail -2
for i in 1 2; do python bench.py --config energy_force --graph-fresh --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done
python bench.py --config energy_force --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['host_busy_ms_per_step'], d['config']['launch_auto'])"
