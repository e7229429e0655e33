#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: the rocprofv3 evidence of round 6 into gpurun_out/r06/
# (kernel-trace stats; separate --pmc passes, no trace domains beside --pmc), then
#   python3 tools/summarize_profiles_r06.py        (runs anywhere)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline"
# 1. default workload: kernel stats + one step's launch sequence, PMC traffic, calibration probe
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $B --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
python3 tools/trace_list.py $(ls -t $OUT/stats/*kernel_trace.csv | head -1) > $OUT/step_kernels.txt      # (a replayed step from the middle of the run)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/probe_fetch -o p -- python3 tools/pmc_probe.py > $OUT/pmc_probe.json 2>/dev/null
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/probe_write -o p -- python3 tools/pmc_probe.py > /dev/null 2>&1
# 1b. L2 hit rates of the edge kernels inside the step (TCC hit rate of tp_fwd / tp_bwd_x) and in isolation vs knot count
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -o p -- $B --steps 3 --warmup 1 > /dev/null 2>&1
# 2. config_energy as shipped (l_max 3): kernel stats + PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/l3_stats -o s -- $B --lmax 3 --steps 10 --warmup 3 > $OUT/l3_bench_under_rocprof.json 2> $OUT/l3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/l3_pmc_fetch -o p -- $B --lmax 3 --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/l3_pmc_write -o p -- $B --lmax 3 --steps 3 --warmup 1 > /dev/null 2>&1
# 3. MFMA busy of the GEMM kernels (default workload)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o p -- $B --steps 4 --warmup 1 > /dev/null 2>&1
# 4. the other BASELINE configurations
for c in energy_force diffusion diffusion_CA; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$c -o s -- $B --config $c --steps 10 --warmup 3 > $OUT/cfg_$c.json 2> $OUT/cfg_$c.err
done
# 4b. the serial kernel census of the replayed steps (one stream: every kernel alone): energy 256 molecules, force training 64
TRACE_ARGS="--config energy_force" bash tools/trace_graph.sh > $OUT/trace_graph_energy_force.txt 2>&1
TRACE_ARGS="--batch 256" bash tools/trace_graph.sh > $OUT/trace_graph_energy.txt 2>&1
E3K_BENCH_PREP_PIPELINE=0 TRACE_ARGS="--batch 256" bash tools/trace_graph.sh > $OUT/trace_graph_energy_one_graph.txt 2>&1
cd "$GRAFT_REPO_ROOT"
# 5. the lines themselves (no profiler; launch mode chosen by the bench unless the line's name says otherwise)
bash tools/collect_lines_r06.sh > $OUT/lines.log 2>&1
# 6. measured errors of the model-level parity tests (tests/util.py: record_measured)
rm -f $OUT/parity_measured.jsonl
E3K_PARITY_LOG=$PWD/$OUT/parity_measured.jsonl python3 -m pytest tests/test_gpu_trained_parity.py tests/test_gpu_model.py tests/test_gpu_double_backward.py -q -m gpu -k "trained or after_training or protein or diffusion or bench_path or guard or backbone or force_block or threshold or shipped_config or position_gradient" > $OUT/parity_tests.log 2>&1
# 7. round-5 probes: the packed-table kernels in isolation (+ the debug library's timing-only ablation of the packed forward), the
#    knot-order walk that would replace the g_w round trip (emulated), the guard's ratios at random init, the host's share of a step
python3 tools/tp_table_bench.py 512 > $OUT/tp_table_bench.txt 2>&1
python3 tools/postlin_bench.py > $OUT/gemm_postlin_bench.txt 2>&1
python3 tools/sample_bench.py 128 50 > $OUT/sampler.txt 2>&1
python3 tools/micro/two_graphs2.py > $OUT/two_graphs_overlap.txt 2>&1
python3 tools/micro/ext_event_torch.py > $OUT/ext_event_torch.txt 2>&1
# 8. 5 x 1000 replayed steps (the knot-table guard at work: refinement, or -- E3K_RADIAL_KNOTS_MAX=512 -- a veto), the transpose's row order
STEPS=1000 bash tools/soak.sh > $OUT/soak.txt 2>&1
python3 tools/micro/transpose_order.py > $OUT/transpose_order.txt 2>&1
ls $OUT; tail -c 600 $OUT/bench_default.json
