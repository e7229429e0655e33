#!/bin/bash
# The node-side GEMM kernels in isolation with the dbg library's timing-only switches (section 7 of
# tools/collect_profiles_r04.sh on its own; needs `make -C equivariant-nn-zoo_amd/csrc dbg`):
#   /usr/local/graft/bin/gpurun -- bash tools/gemm_probe_r04.sh       -> gpurun_out/gemm_probe.txt
# E3K_GEMM_ABLATE   16: gemm_kernel without its MFMAs, 32: without its stores          (wrong results by design)
# E3K_WGRAD2_ABLATE 16: gemm_wgrad2_kernel without its MFMAs, 32: without its loads
# E3K_WGRAD2        0: round-3 weight-gradient kernel, 1: pipelined (shipped), 2: LDS-direct ring (E3K_WGRAD3_CFG 0 / 1 / 2)
# E3K_GEMM_PERSIST  1: the persistent forward / dgrad kernel (E3K_GEMM_PERSIST_MIN_TILES, E3K_GEMM_PERSIST_WG_PER_CU)
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
out=gpurun_out/gemm_probe.txt
{
  for a in 0 16 32 48; do echo "== gemm_kernel  E3K_GEMM_ABLATE=$a"; E3K_LIB=$D E3K_GEMM_ABLATE=$a python3 tools/postlin_bench.py 256 2>&1 | grep "post-linear fwd\|post-linear dgrad"; done
  for a in 0 16 32 48; do echo "== gemm_wgrad2_kernel  E3K_WGRAD2_ABLATE=$a"; E3K_LIB=$D E3K_WGRAD2_ABLATE=$a python3 tools/postlin_bench.py 256 2>&1 | grep "post-linear wgrad"; done
  echo "== round-3 weight gradient"; E3K_LIB=$D E3K_WGRAD2=0 python3 tools/postlin_bench.py 256 2>&1 | grep "post-linear wgrad"
  for c in 0 1 2; do echo "== LDS-direct ring, configuration $c"; E3K_LIB=$D E3K_WGRAD2=2 E3K_WGRAD3_CFG=$c python3 tools/postlin_bench.py 256 2>&1 | grep "post-linear wgrad"; done
  for w in 1 2 3; do echo "== persistent forward / dgrad, $w workgroups per CU"; E3K_LIB=$D E3K_GEMM_PERSIST=1 E3K_GEMM_PERSIST_MIN_TILES=1 E3K_GEMM_PERSIST_WG_PER_CU=$w python3 tools/postlin_bench.py 256 2>&1 | grep "post-linear fwd\|post-linear dgrad\|K  1024"; done
} > $out 2>&1
cat $out
