#!/bin/bash
# round-6 experiment 1: launch modes of the default workload on one box (eager / replay / forked replay / runtime graph knobs)
mkdir -p gpurun_out/e1
B="python3 bench.py --no-cpu-baseline --steps 20 --warmup 5"
$B > gpurun_out/e1/eager.json 2> gpurun_out/e1/eager.err
$B --graph-fresh > gpurun_out/e1/replay.json 2> gpurun_out/e1/replay.err
E3K_FWD_FORK=2 $B --graph-fresh > gpurun_out/e1/replay_fork.json 2> gpurun_out/e1/replay_fork.err
E3K_FWD_FORK=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 $B --graph-fresh > gpurun_out/e1/replay_fork_q8.json 2> gpurun_out/e1/replay_fork_q8.err
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B --graph-fresh > gpurun_out/e1/replay_nopkt.json 2> gpurun_out/e1/replay_nopkt.err
E3K_FWD_FORK=2 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B --graph-fresh > gpurun_out/e1/replay_fork_nopkt.json 2> gpurun_out/e1/replay_fork_nopkt.err
$B --batch 32 --graph-fresh > gpurun_out/e1/replay_b32.json 2> gpurun_out/e1/replay_b32.err
for f in gpurun_out/e1/*.json; do echo "$f: $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats'], d['host_busy_ms_per_step'])" 2>&1)"; done
