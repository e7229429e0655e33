#!/usr/bin/env python3
"""Kernel sequence of ONE training step from a rocprofv3 --kernel-trace CSV (python tools/step_kernels.py trace.csv [out.txt]):
every launch between two consecutive optimizer kernels in start order -- offset, duration, queue, name -- and the
launch count per kernel name.  The launch census DESIGN.md section 5 quotes comes from this."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r["Kernel_Name"]]
a, b = adam[-3], adam[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
queues = {}
nm = lambda r: r["Kernel_Name"].replace("void ", "").replace("e3k::", "").replace("at::native::", "")[:110]
print(f"# {len(step)} launches, span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us", file=out)
for r in step:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} q{q} {nm(r)}", file=out)
cnt = collections.Counter(nm(r)[:70] for r in step)
dur = collections.Counter()
for r in step:
    dur[nm(r)[:70]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("# ---- per name: launches, total us", file=out)
for k, v in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"# {v:4d} {dur[k] / 1e3:8.1f}  {k}", file=out)
