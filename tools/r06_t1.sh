#!/bin/bash
export E3K_PARITY_LOG=$GRAFT_REPO_ROOT/gpurun_out/r06_parity_measured.jsonl
rm -f $E3K_PARITY_LOG
timeout 1500 python3 -m pytest tests/test_gpu_trained_parity.py -x -q -m gpu 2>&1 | tail -40
cat $E3K_PARITY_LOG
