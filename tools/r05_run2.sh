mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests -q -m gpu -x -k "in_kernel_knot_table or radial_table or bench_path or conv_block or radial_stack or threshold" > gpurun_out/r05/tests3.log 2>&1
tail -6 gpurun_out/r05/tests3.log
python3 tools/tp_table_bench.py 512 > gpurun_out/r05/tp_table_bench_packed.txt 2>&1
cat gpurun_out/r05/tp_table_bench_packed.txt
for i in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_packed.json 2> gpurun_out/r05/bench_packed.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_packed.json'))
r=d['roofline']
print('PACKED', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
E3K_TP_TABLE_PACKED=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_4row.json 2> gpurun_out/r05/bench_4row.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_4row.json'))
r=d['roofline']
print('4ROW  ', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
done
