#!/bin/bash
# cache policy of the g_w round trip (tp_bwd_x writes [E, W], the table transpose reads it once): nontemporal vs default stores / loads.
# Variant libraries are built by hand: libe3k_x_{pp,pn,np}.so = (tp stores, transpose loads) plain/nt.
C=$PWD/equivariant-nn-zoo_amd/csrc
for round in 1 2; do
for v in "" _x_pp _x_pn _x_np; do
  echo -n "libe3k$v: "
  E3K_LIB=$C/libe3k$v.so python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_repeats']['min'], d['ms_per_step_repeats']['max'])"
done
done
