# Run on the GPU box (via gpurun) from the repo root: kernel trace of a short bench run + the concurrency summary
# of one step (tools/trace_gaps.py): per-queue busy time, overlap histogram, what the main queue waited for.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/tr.log 2>&1
cd $GRAFT_REPO_ROOT
cut -c120-180 gpurun_out/tr.log | tail -1
python3 tools/trace_gaps.py $(ls -t gpurun_out/tr/*kernel_trace.csv | head -1)
python3 tools/step_kernels.py $(ls -t gpurun_out/tr/*kernel_trace.csv | head -1) gpurun_out/step_kernels.txt
