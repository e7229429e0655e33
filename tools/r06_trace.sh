#!/bin/bash
# ordered kernel list + census of one replayed step (TRACE_ARGS: bench arguments, default the replayed 256-molecule step)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TRACE_NAME:-trg}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${TRACE_ARGS:---graph-fresh --batch 256} > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv 2>/dev/null | tail -1)
python3 tools/trace_list.py $f > $OUT.list.txt
tail -1 $OUT.list.txt
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv   # (tens of MB: only the summaries travel back)
