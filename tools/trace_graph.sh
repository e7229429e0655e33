cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trg
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trg -o tr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline ${TRACE_ARGS:---graph-fresh --batch 256} > $GRAFT_REPO_ROOT/gpurun_out/trg.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/trg/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 10 steps = timed region: find adam kernels
adam = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r["Kernel_Name"]]
# five whole REPLAYED steps from the middle of the run (the last steps of a bench run are the eager host-time measurement: several
# queues; a replayed step runs on one)
mid = len(adam) // 2
lo, hi = adam[mid - 3], adam[mid + 2]
sel = rows[lo + 1: hi + 1]
print("queues in the selection:", sorted({r["Queue_Id"] for r in sel}))
agg = collections.Counter(); cnt = collections.Counter()
for r in sel:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("e3k::", "")
    agg[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / 5; cnt[n] += 1 / 5
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3 / 5
print("per step: span %.0f us, kernel sum %.0f us, launches %.0f" % (span, sum(agg.values()), sum(cnt.values())))
for n, v in agg.most_common(24): print("%8.1f us %5.1f x  %s" % (v, cnt[n], n[:90]))
PY
