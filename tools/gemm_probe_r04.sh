#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gate or norm_act or block" 2>&1 | tail -2
for a in "" ""; do
  python3 bench.py --no-cpu-baseline --steps 40 $a 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_repeats',{}).get('median'))"
done 2>&1 | tee gpurun_out/lines_now.txt
bash tools/trace_graph.sh 2>&1 | grep "per step\|gate" | tee gpurun_out/trace_now.txt
