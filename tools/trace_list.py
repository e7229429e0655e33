#!/usr/bin/env python3
"""Ordered kernel list of ONE step out of a rocprofv3 --kernel-trace CSV (python tools/trace_list.py trace.csv [which]):
start offset, duration, gap to the previous kernel's end, queue, name.  ``which``: index of the step counted in optimizer
launches (default: the middle of the run -- a replayed step when the bench ran --graph-fresh)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(adam) // 2
sel = rows[adam[k - 1] + 1: adam[k] + 1]
t0 = int(sel[0]["Start_Timestamp"])
prev = t0
tot = 0
for i, r in enumerate(sel):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("e3k::", "")
    n = n.split("(")[0][:100]
    print(f"{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {(s - prev) / 1e3:6.1f} q{r['Queue_Id']} grid {r.get('Grid_Size_X', '?'):>8} {n}")
    prev = max(prev, e)
    tot += e - s
print(f"span {(prev - t0) / 1e3:.1f} us, kernel sum {tot / 1e3:.1f} us, {len(sel)} launches")
