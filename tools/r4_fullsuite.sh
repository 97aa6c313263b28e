mkdir -p gpurun_out
(timeout 3000 python -m pytest tests -m gpu -q --durations=40 2>&1 | tail -n 70) > gpurun_out/r4_fullsuite.log
tail -n 70 gpurun_out/r4_fullsuite.log
