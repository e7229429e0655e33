#!/usr/bin/env python3
"""Turns gpurun_out/<src>/* (tools/collect_profiles.sh) into the committed summaries under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_bench_*.json and <tag>_tp_traffic.json (HBM bytes per launch of the three edge
kernels from the FETCH_SIZE / WRITE_SIZE counters, calibrated on a known 1 GiB stream in the same session)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02")
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02"
DST = os.path.join(ROOT, "profiles")
KERNELS = {"tp_fwd": "tp_fwd_kernel", "tp_bwd_x": "tp_bwd_x_kernel", "tp_bwd_w": "tp_bwd_w_kernel"}


def counters(d, name):
    f = max(glob.glob(os.path.join(SRC, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        key = "calibration_act_fwd" if "act_fwd" in kn else next((k for k, pat in KERNELS.items() if pat in kn), None)
        if key and r["Counter_Name"] == name:
            agg[key].append(float(r["Counter_Value"]))
    return agg


stats = max(glob.glob(os.path.join(SRC, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"))
for name in ("bench_under_rocprof.json", "bench_default.json"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{TAG}_{name}"))
pf, pw = counters("probe_fetch", "FETCH_SIZE"), counters("probe_write", "WRITE_SIZE")
cal_f = 2 ** 20 / (sum(pf["calibration_act_fwd"]) / len(pf["calibration_act_fwd"]))   # counters are in KiB
cal_w = 2 ** 20 / (sum(pw["calibration_act_fwd"]) / len(pw["calibration_act_fwd"]))
bf, bw = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
under = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))
rows = list(csv.DictReader(open(stats)))
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py (default workload), MI355X; "
              "calibrated on a known 1 GiB dword-per-lane stream in tools/pmc_probe.py (FETCH_SIZE reads 1/2 on gfx950)",
    "workload": under["config"]["workload"],
    "calibration": {"fetch_factor": cal_f, "write_factor": cal_w},
}
by_name = {"tp_fwd": under["roofline"]}
by_name.update({k["kernel"].split("::")[1].replace("_kernel", ""): k for k in under["roofline"].get("kernels", []) if "::tp_" in k["kernel"]})
for key, pat in KERNELS.items():
    if not bf.get(key):
        continue
    rd = sum(bf[key]) / len(bf[key]) * 1024 * cal_f
    wr = sum(bw[key]) / len(bw[key]) * 1024 * cal_w
    k_rows = [r for r in rows if pat in r["Name"]]      # one instantiation per (max l1, max l3) of a plan
    calls = sum(int(r["Calls"]) for r in k_rows)
    out[key] = {
        "kernel": ", ".join(r["Name"].split("(")[0].replace("void ", "") + " x" + r["Calls"] for r in k_rows),
        "launches_sampled": len(bf[key]),
        "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": by_name.get(key, {}).get("avg_launch_algorithmic_MB", 0.0) * 1e6,
        "rocprof_avg_launch_us": sum(float(r["TotalDurationNs"]) for r in k_rows) / max(calls, 1) / 1e3,
        "bench_event_avg_launch_us": by_name.get(key, {}).get("avg_launch_us"),
    }
json.dump(out, open(os.path.join(DST, f"{TAG}_tp_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
