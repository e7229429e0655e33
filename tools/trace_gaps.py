#!/usr/bin/env python3
"""Concurrency summary of one training step from a rocprofv3 --kernel-trace CSV (python tools/trace_gaps.py trace.csv):
per-queue busy time, how long 0/1/2/... kernels ran at once, and the largest gaps of the main queue."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_ema_kernel" in r["Kernel_Name"]]
a, b = adam[-3], adam[-2]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels")
byq = collections.defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"queue {q}: {len(rs)} kernels, busy {busy / 1e3:.1f} us")
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in step] + [(int(r["End_Timestamp"]), -1) for r in step])
cur, last, conc = 0, None, collections.Counter()
for t, d in ev:
    if last is not None:
        conc[cur] += t - last
    cur += d
    last = t
print("time with k kernels running:", {k: round(v / 1e3, 1) for k, v in sorted(conc.items())})
main_q = max(byq.items(), key=lambda kv: len(kv[1]))[0]
main = byq[main_q]
others = [r for r in step if r["Queue_Id"] != main_q]
nm = lambda r: r["Kernel_Name"].replace("void ", "").replace("e3k::", "")[:36]
gaps = []
for i in range(len(main) - 1):
    s, e = int(main[i]["End_Timestamp"]), int(main[i + 1]["Start_Timestamp"])
    if e - s > 30000:
        running = [nm(r) + f"@q{r['Queue_Id']}" for r in others if int(r["Start_Timestamp"]) < e and int(r["End_Timestamp"]) > s]
        gaps.append((e - s, (s - t0) / 1e3, nm(main[i]), nm(main[i + 1]), running[-3:]))
print(f"main queue {main_q}: gaps > 30 us: {len(gaps)}, total {sum(g[0] for g in gaps) / 1e3:.1f} us")
for g in sorted(gaps, reverse=True)[:16]:
    print(f"  {g[0] / 1e3:7.1f} us at {g[1]:8.1f}: {g[2]} -> {g[3]} | meanwhile {g[4]}")
