mkdir -p gpurun_out/r05
timeout 900 python3 -m pytest tests -q -m gpu -x -k "in_kernel_knot_table or conv_block_equals or bench_path or threshold" 2>&1 | tail -4
D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
echo "== product"; python3 tools/tp_table_bench.py 512 2>&1 | tail -2
echo "== dbg ablation"; E3K_LIB=$D python3 tools/tp_table_bench.py --ablate 2>&1 | tail -6
for o in 1 2; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_rec.json 2> gpurun_out/r05/bench_rec.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_rec.json'))
r=d['roofline']
print('REC', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lmax 3 > gpurun_out/r05/bench_rec_l3.json 2> gpurun_out/r05/bench_rec_l3.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_rec_l3.json'))
r=d['roofline']
print('REC l3', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
E3K_TP_TABLE_PACKED=0 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lmax 3 > gpurun_out/r05/bench_4row_l3.json 2> gpurun_out/r05/bench_4row_l3.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_4row_l3.json'))
r=d['roofline']
print('4ROW l3', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
