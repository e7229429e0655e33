# Run on the GPU box: kernel trace of a short bench run at a given batch size + launch census / idle analysis of one step
#   bash tools/trace_small.sh 64
cd /tmp && export TMPDIR=/tmp
B=${1:-64}
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trs
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trs -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --batch $B > $GRAFT_REPO_ROOT/gpurun_out/trs.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/trace_gaps.py $(ls -t gpurun_out/trs/*kernel_trace.csv | head -1) | head -8
python3 tools/step_kernels.py $(ls -t gpurun_out/trs/*kernel_trace.csv | head -1) gpurun_out/step_kernels_b$B.txt
