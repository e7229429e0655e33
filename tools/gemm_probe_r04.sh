#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export E3K_LIB=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
out=gpurun_out/gemm_probe.txt; : > $out
for a in 1 2 4 8 15 30; do
  echo "== SK_CT $a" >> $out
  E3K_SK_CT=$a timeout 120 python3 tools/postlin_bench.py 256 2>&1 | grep "radial last" >> $out
done
cat $out
