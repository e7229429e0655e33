# Run on the GPU box: cProfile of bench.py's eager step (host side) for one configuration:  bash tools/host_cprofile.sh diffusion_CA
cfg=${1:-diffusion_CA}
E3K_BENCH_AUTO=0 timeout 400 python -m cProfile -o gpurun_out/$cfg.prof bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
python - "$cfg" <<'PY'
import pstats, sys
p = pstats.Stats("gpurun_out/%s.prof" % sys.argv[1])
p.sort_stats("tottime").print_stats(45)
PY
