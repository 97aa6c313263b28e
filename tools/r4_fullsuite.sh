mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -n 40) > gpurun_out/r4_fullsuite.log
tail -n 40 gpurun_out/r4_fullsuite.log
