mkdir -p gpurun_out/r05
E3K_PARITY_LOG=$PWD/gpurun_out/r05/parity_diff.jsonl timeout 900 python3 -m pytest tests -q -m gpu -x -k "diffusion or knot_bins or radial_table" 2>&1 | tail -15
cat gpurun_out/r05/parity_diff.jsonl | tail -3
for i in 1 2; do
python3 bench.py --config diffusion --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/cfg_diffusion.json 2> gpurun_out/r05/cfg_diffusion.err
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/cfg_diffusion.json'))
print('diffusion', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], d['config'].get('launch','')[:30], d['config'].get('launch_auto'))
PY
done
E3K_RADIAL_TABLE=0 python3 bench.py --config diffusion --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/cfg_diffusion0.json 2>/dev/null
python3 - <<PY
import json
d=json.load(open('gpurun_out/r05/cfg_diffusion0.json'))
print('diffusion NO TABLE', d['value'], d['ms_per_step'], d['ms_per_step_repeats']['min'], 'host', d['host_busy_ms_per_step'], d['config'].get('launch','')[:30], d['config'].get('launch_auto'))
PY
