#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export E3K_LIB=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
E3K_SK_NATURAL=1 timeout 900 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear or lin or mlp or radial" 2>&1 | tail -2
out=gpurun_out/gemm_probe.txt; : > $out
for a in 0 1; do
  echo "== SK_NATURAL $a" >> $out
  E3K_SK_NATURAL=$a timeout 120 python3 tools/postlin_bench.py 256 2>&1 | grep "radial last\|N 64 K    64" >> $out
done
cat $out
