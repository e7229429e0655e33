#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2
D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
python3 tools/ab_bench.py "new:" "oldwgrad:E3K_LIB=$D,E3K_WGRAD2=0" --rounds 3 --steps 40 2>&1 | tail -3 | tee gpurun_out/ab_wgrad.txt
bash tools/trace_graph.sh 2>&1 | tail -28 | tee gpurun_out/trace_wgrad2.txt
