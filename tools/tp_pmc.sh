# Run on the GPU box: rocprofv3 --pmc passes over tools/tp_pmc.py (the three TP kernels of layer 3 in isolation);
# prints per kernel the mean of each counter.   bash tools/tp_pmc.sh "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY ..."
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/tp_pmc_$i
  rm -rf $out
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/tp_pmc.py > /dev/null 2>&1
  python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file in", sys.argv[1]); sys.exit()
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "tp_" not in k: continue
    acc[k.split("(")[0].replace("void e3k::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
done
