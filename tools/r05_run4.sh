mkdir -p gpurun_out/r05
timeout 600 python3 -m pytest tests -q -m gpu -x -k "in_kernel_knot_table or conv_block_equals or threshold" 2>&1 | tail -3
python3 tools/tp_table_bench.py 512 2>&1 | tail -3
python3 tools/tp_table_bench.py --knot-order 2>&1 | tail -2 | tee gpurun_out/r05/knot_order_walk.txt
for o in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_rec.json 2> gpurun_out/r05/bench_rec.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_rec.json'))
r=d['roofline']
print('REC', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
done
