#!/usr/bin/env python3
"""Turns gpurun_out/r01/* (tools/collect_profiles.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r01"
DST = os.path.join(ROOT, "profiles")


def counters(d, name):
    f = max(glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        key = "calibration_act_fwd" if "act_fwd" in kn else ("tp_fwd" if "tp_fwd_kernel" in kn else None)
        if key and r["Counter_Name"] == name:
            agg[key].append(float(r["Counter_Value"]))
    return agg


stats = max(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))
for name in ("bench_under_rocprof.json", "bench_default.json"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{TAG}_{name}"))
probe = json.load(open(os.path.join(SRC, "pmc_probe.json")))
pf, pw = counters("probe_fetch", "FETCH_SIZE"), counters("probe_write", "WRITE_SIZE")
cal_f = 2 ** 20 / (sum(pf["calibration_act_fwd"]) / len(pf["calibration_act_fwd"]))   # counters are in KiB
cal_w = 2 ** 20 / (sum(pw["calibration_act_fwd"]) / len(pw["calibration_act_fwd"]))
bf, bw = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
n = len(bf["tp_fwd"])
rd = sum(bf["tp_fwd"]) / n * 1024 * cal_f
wr = sum(bw["tp_fwd"]) / len(bw["tp_fwd"]) * 1024 * cal_w
under = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))
rows = list(csv.DictReader(open(stats)))
tp_rows = [r for r in rows if "tp_fwd_kernel" in r["Name"]]   # one instantiation per (max l1, max l3) of a plan
tp_calls = sum(int(r["Calls"]) for r in tp_rows)
tp = {"AverageNs": sum(float(r["TotalDurationNs"]) for r in tp_rows) / tp_calls}
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py (default workload), MI355X; "
              "calibrated on a known 1 GiB dword-per-lane stream in tools/pmc_probe.py (FETCH_SIZE reads 1/2 on gfx950)",
    "workload": under["config"]["workload"],
    "kernel": "e3k::tp_fwd_kernel (all instantiations: " + ", ".join(r["Name"].split("(")[0].replace("void ", "") + " x" + r["Calls"] for r in tp_rows) + ")",
    "launches_sampled": n,
    "calibration": {"fetch_factor": cal_f, "write_factor": cal_w},
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": under["roofline"]["avg_launch_algorithmic_MB"] * 1e6,
    "rocprof_avg_launch_us": float(tp["AverageNs"]) / 1e3,
    "bench_event_avg_launch_us": under["roofline"]["avg_launch_us"],
    "layer3_probe": {"algorithmic_bytes": probe["algorithmic_bytes_variant_A"],
                     "hbm_read_bytes": sum(pf["tp_fwd"]) / len(pf["tp_fwd"]) * 1024 * cal_f,
                     "hbm_write_bytes": sum(pw["tp_fwd"]) / len(pw["tp_fwd"]) * 1024 * cal_w},
}
json.dump(out, open(os.path.join(DST, f"{TAG}_tp_fwd_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
