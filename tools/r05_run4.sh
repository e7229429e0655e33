mkdir -p gpurun_out/r05
E3K_PARITY_LOG=$PWD/gpurun_out/r05/parity_force.jsonl timeout 1500 python3 -m pytest tests/test_gpu_double_backward.py -q -m gpu -x -k "force_block" 2>&1 | tail -6
cat gpurun_out/r05/parity_force.jsonl | tail -6
