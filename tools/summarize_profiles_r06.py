#!/usr/bin/env python3
"""gpurun_out/r06/* (tools/collect_profiles_r06.sh) -> the committed summaries under profiles/r06_*:
kernel stats (default workload, l_max 3, the other BASELINE configurations), one step's launch census, HBM traffic per
launch of the three edge kernels from the FETCH_SIZE / WRITE_SIZE counters (separate passes, calibrated on a known
1 GiB stream in the same session: FETCH_SIZE reads 1/2 on gfx950) for l_max 2 and l_max 3, MFMA-busy fractions of the
GEMM kernels, and the bench lines of the session."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r06")
DST = os.path.join(ROOT, "profiles")
KERNELS = {"tp_fwd": "tp_fwd_kernel", "tp_bwd_x": "tp_bwd_x_kernel", "tp_bwd_w": "tp_bwd_w_kernel"}


def newest(pattern):
    return max(glob.glob(os.path.join(SRC, pattern), recursive=True), key=os.path.getmtime)


def counters(d, name, keymap):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{d}/**/*counter_collection.csv"))):
        kn = r["Kernel_Name"]
        key = "calibration_act_fwd" if "act_fwd" in kn else next((k for k, pat in keymap.items() if pat in kn), None)
        if key and r["Counter_Name"] == name:
            agg[key].append(float(r["Counter_Value"]))
    return agg


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


pf, pw = counters("probe_fetch", "FETCH_SIZE", {}), counters("probe_write", "WRITE_SIZE", {})
cal_f = 2 ** 20 / (sum(pf["calibration_act_fwd"]) / len(pf["calibration_act_fwd"]))   # counters are in KiB
cal_w = 2 ** 20 / (sum(pw["calibration_act_fwd"]) / len(pw["calibration_act_fwd"]))


def traffic(tag, stats_dir, fetch_dir, write_dir, under_json, out_name):
    stats = newest(f"{stats_dir}/**/*kernel_stats.csv")
    shutil.copy(stats, os.path.join(DST, f"r06_{tag}kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    bf, bw = counters(fetch_dir, "FETCH_SIZE", KERNELS), counters(write_dir, "WRITE_SIZE", KERNELS)
    under = last_json(os.path.join(SRC, under_json))
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py, MI355X; calibrated on a known "
                     "1 GiB dword-per-lane stream in tools/pmc_probe.py (FETCH_SIZE reads 1/2 on gfx950)",
           "workload": under["config"]["workload"], "calibration": {"fetch_factor": cal_f, "write_factor": cal_w}}
    by_name = {"tp_fwd": under["roofline"]}
    by_name.update({k["kernel"].split("::")[1].split(" ")[0].replace("_kernel", ""): k for k in under["roofline"].get("kernels", []) if "::tp_" in k["kernel"]})
    for key, pat in KERNELS.items():
        if not bf.get(key):
            continue
        rd = sum(bf[key]) / len(bf[key]) * 1024 * cal_f
        wr = sum(bw[key]) / len(bw[key]) * 1024 * cal_w
        k_rows = [r for r in rows if pat in r["Name"]]
        calls = sum(int(r["Calls"]) for r in k_rows)
        out[key] = {"kernel": ", ".join(r["Name"].split("(")[0].replace("void ", "") + " x" + r["Calls"] for r in k_rows),
                    "launches_sampled": len(bf[key]), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                    "traffic_bytes_per_launch": rd + wr,
                    "algorithmic_bytes_per_launch": by_name.get(key, {}).get("avg_launch_algorithmic_MB", 0.0) * 1e6,
                    "rocprof_avg_launch_us": sum(float(r["TotalDurationNs"]) for r in k_rows) / max(calls, 1) / 1e3,
                    "bench_event_avg_launch_us": by_name.get(key, {}).get("avg_launch_us")}
    json.dump(out, open(os.path.join(DST, out_name), "w"), indent=1)
    return out


t2 = traffic("bench_", "stats", "pmc_fetch", "pmc_write", "bench_under_rocprof.json", "r06_tp_traffic.json")
t3 = traffic("lmax3_", "l3_stats", "l3_pmc_fetch", "l3_pmc_write", "l3_bench_under_rocprof.json", "r06_lmax3_tp_traffic.json")
shutil.copy(os.path.join(SRC, "step_kernels.txt"), os.path.join(DST, "r06_step_kernels.txt"))
for name, tag in (("cfg_energy_force", "config3"), ("cfg_diffusion", "config4"), ("cfg_diffusion_CA", "config5")):
    shutil.copy(newest(f"{name}/**/*kernel_stats.csv"), os.path.join(DST, f"r06_{tag}_kernel_stats.csv"))
# ---- L2 hit rates of the edge kernels inside the step
l2 = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(newest("pmc_l2/**/*counter_collection.csv"))):
    kn = r["Kernel_Name"]
    if "tp_" in kn or "rtable" in kn:
        l2[kn.split("(")[0].replace("void e3k::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(DST, "r06_tp_l2_hit.txt"), "w") as f:
    f.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 (default workload, MI355X)\n"
            "# per kernel: launches, mean L2 hits / misses per launch (millions of requests), hit rate\n")
    for k, cs in sorted(l2.items()):
        h, m = cs.get("TCC_HIT_sum", []), cs.get("TCC_MISS_sum", [])
        if h and m:
            hm, mm = sum(h) / len(h), sum(m) / len(m)
            f.write(f"{k}: launches {len(h)} hits {hm / 1e6:.2f} M misses {mm / 1e6:.2f} M hit rate {hm / max(hm + mm, 1):.3f}\n")
for name in ("tp_table_bench.txt", "trace_graph_energy_force.txt", "trace_graph_energy.txt", "trace_graph_energy_one_graph.txt", "gemm_postlin_bench.txt",
             "sampler.txt", "two_graphs_overlap.txt", "ext_event_torch.txt", "soak.txt", "transpose_order.txt"):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, "r06_" + name))
# ---- MFMA busy of the GEMM kernels
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(newest("pmc_mfma/**/*kernel_trace.csv")))}
for r in csv.DictReader(open(newest("pmc_mfma/**/*counter_collection.csv"))):
    kn = r["Kernel_Name"]
    if "gemm_" not in kn and "mlp_hidden" not in kn:
        continue
    name = kn.split("(")[0].replace("void ", "")
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[name] += 1
        t = trace.get(r["Dispatch_Id"])
        if t:
            dur[name] += (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
mfma = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -- python3 bench.py "
                  "--no-cpu-baseline --steps 4 --warmup 1 (default workload, MI355X)",
        "normalisation": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over all 1024 SIMDs; busy "
                         "fraction = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)", "kernels": {}}
for name, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    mfma["kernels"][name] = {"launches": cnt[name], "total_us": round(dur[name], 1),
                             "mfma_busy_fraction": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024), 4) if gui else None}
json.dump(mfma, open(os.path.join(DST, "r06_gemm_mfma_util.json"), "w"), indent=1)
# ---- bench lines
lines = {}
for f in sorted(glob.glob(os.path.join(SRC, "bench_*.json"))):
    try:
        d = last_json(f)
        lines[os.path.basename(f)[:-5]] = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline") if k in d}
        if "cpu_baseline" in d:
            lines[os.path.basename(f)[:-5]]["cpu_baseline"] = d["cpu_baseline"]
    except Exception as exc:
        lines[os.path.basename(f)[:-5]] = {"error": str(exc)}
json.dump(lines, open(os.path.join(DST, "r06_bench_lines.json"), "w"), indent=1)
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, "r06_bench_default.json"))
shutil.copy(os.path.join(SRC, "bench_under_rocprof.json"), os.path.join(DST, "r06_bench_under_rocprof.json"))
for key in ("tp_fwd", "tp_bwd_x", "tp_bwd_w"):
    for tag, t in (("l_max 2", t2), ("l_max 3", t3)):
        if key in t:
            k = t[key]
            print(f"{tag} {key}: traffic {k['traffic_bytes_per_launch'] / 1e6:.0f} MB vs algorithmic {k['algorithmic_bytes_per_launch'] / 1e6:.0f} MB, "
                  f"rocprof {k['rocprof_avg_launch_us']:.1f} us, bench events {k['bench_event_avg_launch_us']} us")
print(json.dumps(mfma["kernels"], indent=1)[:1500])

pm = os.path.join(SRC, "parity_measured.jsonl")
if os.path.exists(pm) and sum(1 for _ in open(pm)) > 1:
    shutil.copy(pm, os.path.join(DST, "r06_parity_measured.jsonl"))
