mkdir -p gpurun_out/r05
timeout 2400 python3 -m pytest tests -q -m gpu -k "guard or captured or bucketed or replay" > gpurun_out/r05/tests2.log 2>&1
tail -8 gpurun_out/r05/tests2.log
