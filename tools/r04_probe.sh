#!/bin/bash
# GPU box: isolated table kernels vs knot count, L2 hit rates, serial kernel census of the energy and the force step
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; rm -rf $O; mkdir -p $O
python3 $R/tools/tp_table_bench.py 128 256 512 1024 2048 > $O/tp_table_bench.txt 2>&1
python3 $R/tools/tp_table_bench.py --clustered 512 2048 >> $O/tp_table_bench.txt 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_l2 -o p -- python3 $R/tools/tp_table_bench.py --pmc 512 2048 > /dev/null 2>&1
python3 - "$O/pmc_l2" > $O/l2_hit.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "tp_" not in k and "rtable" not in k: continue
    acc[k.split("(")[0].replace("void e3k::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    h, m = cs.get("TCC_HIT_sum", [0]), cs.get("TCC_MISS_sum", [0])
    # launches alternate between knot counts in program order: print per-launch values
    print(k, "hits", [round(v / 1e6, 2) for v in h], "misses", [round(v / 1e6, 2) for v in m])
PY
cd $R
TRACE_ARGS="--graph-fresh --config energy_force" bash tools/trace_graph.sh > $O/trace_force.txt 2>&1
cp gpurun_out/trg.log $O/trace_force.log
TRACE_ARGS="--graph-fresh --batch 256" bash tools/trace_graph.sh > $O/trace_energy.txt 2>&1
cat $O/tp_table_bench.txt $O/l2_hit.txt $O/trace_force.txt $O/trace_energy.txt
