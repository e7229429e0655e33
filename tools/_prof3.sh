cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof$1 -o cfg -- python3 $GRAFT_REPO_ROOT/tools/config_bench.py 10 --only=$1 > $GRAFT_REPO_ROOT/gpurun_out/prof$1.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/prof$1/*/*kernel_stats.csv gpurun_out/prof$1/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", tot / 13 / 1e6, "launches/step", sum(int(r["Calls"]) for r in rows) / 13)
for r in rows[:32]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"])/13:7.1f} {int(r["TotalDurationNs"])/13/1e3:9.1f} us  {float(r["AverageNs"])/1e3:8.1f} us')
PY
