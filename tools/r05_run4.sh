D=$PWD/equivariant-nn-zoo_amd/csrc/libe3k_dbg.so
for o in 1 64 128 256 0; do echo "== fwd order $o"; E3K_LIB=$D E3K_TP_ORDER=$o python3 tools/tp_table_bench.py 512 2>&1 | tail -1 | cut -c60-260; done
for o in 64 128 256; do echo "== fwd order 1, bwd_x order $o"; E3K_LIB=$D E3K_TP_ORDER=1 E3K_TP_ORDER_BWD_X=$o python3 tools/tp_table_bench.py 512 2>&1 | tail -1 | cut -c60-260; done
