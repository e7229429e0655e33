# Run on the GPU box (via gpurun) from the repo root:  bash tools/profile_config.sh N   (N = 3, 4 or 5)
# rocprofv3 kernel stats of one BASELINE.json configuration's training step (tools/config_bench.py --only=N);
# prints the per-step top kernels and leaves the CSV under gpurun_out/cfgN/.
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cfg$1 -o cfg -- python3 $GRAFT_REPO_ROOT/tools/config_bench.py 10 --only=$1 > $GRAFT_REPO_ROOT/gpurun_out/cfg$1.log 2>&1
cd $GRAFT_REPO_ROOT
grep "#$1" gpurun_out/cfg$1.log
python3 - "$(ls -t gpurun_out/cfg$1/*kernel_stats.csv | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 13   # 3 warm-up + 10 timed
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time {tot / steps / 1e6:.2f} ms per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:24]:
    print(f'{r["Name"][:84]:84s} {int(r["Calls"]) / steps:6.1f} x {float(r["AverageNs"]) / 1e3:8.1f} us = {int(r["TotalDurationNs"]) / steps / 1e3:8.1f} us')
PY
