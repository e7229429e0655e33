# Run on the GPU box: one rocprofv3 --pmc pass over the default bench (eager layout pinned), counters averaged per TP kernel:
#   bash tools/tp_pmc_bench.sh "TCC_HIT_sum TCC_MISS_sum"
cd /tmp && export TMPDIR=/tmp
export E3K_BENCH_AUTO=0
out=$GRAFT_REPO_ROOT/gpurun_out/tp_pmc_bench
rm -rf $out
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "tp_" not in k or "<2, 2" not in k: continue
    acc[k.split("(")[0].replace("void e3k::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in cs.items()}, "(millions per launch) launches", len(next(iter(cs.values()))))
PY
