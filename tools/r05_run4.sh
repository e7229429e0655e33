mkdir -p gpurun_out/r05
timeout 600 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "in_kernel_knot_table" 2>&1 | tail -3
D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
echo "== product"; python3 tools/tp_table_bench.py 512 2>&1 | tail -2
echo "== dbg order 0"; E3K_LIB=$D python3 tools/tp_table_bench.py 512 2>&1 | tail -1
echo "== dbg order 1"; E3K_LIB=$D E3K_TP_ORDER=1 python3 tools/tp_table_bench.py 512 2>&1 | tail -1
echo "== dbg order 1 correctness"; E3K_LIB=$D E3K_TP_ORDER=1 timeout 600 python3 -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "in_kernel_knot_table" 2>&1 | tail -2
for o in 0 1; do
E3K_LIB=$D E3K_TP_ORDER=$o python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_order$o.json 2> gpurun_out/r05/bench_order$o.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/bench_order$o.json'))
r=d['roofline']
print('ORDER $o', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], 'tp_fwd us', r.get('avg_launch_us'), 'frac', r['frac'], [ (k['kernel'][5:20], k.get('avg_launch_us')) for k in r['kernels']])
PY
done
