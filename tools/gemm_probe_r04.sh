#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
python3 tools/ab_bench.py "new:" "oldwgrad:E3K_LIB=$D,E3K_WGRAD2=0" --rounds 3 --steps 40 2>&1 | tail -3 | tee gpurun_out/ab_wgrad.txt
