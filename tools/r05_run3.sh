D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
E3K_LIB=$D python3 tools/tp_table_bench.py --ablate 2>&1 | tail -8
