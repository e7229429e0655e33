#!/usr/bin/env python3
"""A/B timing of the default bench workload under environment variants, interleaved in ONE process group session so
that box-to-box and run-to-run spread (a few percent) does not hide a small difference:
    python tools/ab_bench.py "A:" "B:E3K_CONV_BLOCK=0" "C:E3K_CF_CHAIN=0,E3K_CONV_BLOCK=0" [--rounds 4] [--steps 30]
Each variant runs bench.py --no-cpu-baseline in a child process `rounds` times, round-robin; prints the median and
minimum ms/step per variant."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:sys.argv.index("--")] if "--" in sys.argv else sys.argv[1:]
args = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] not in ("--rounds", "--steps"))]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 4
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
variants = []
for a in args:
    name, _, envs = a.partition(":")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(steps)] + extra,
                             env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-500:])
            continue
        res[name].append(json.loads(line[-1])["ms_per_step"])
for name, _ in variants:
    v = res[name]
    if v:
        print(f"{name:12s} median {statistics.median(v):.3f} ms  min {min(v):.3f}  all {v}")
