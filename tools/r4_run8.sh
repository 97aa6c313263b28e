mkdir -p gpurun_out
timeout 300 tools/bin/exp_stream_probe 10 2>&1 | tail -12 > gpurun_out/r4_exp_probe.txt
cat gpurun_out/r4_exp_probe.txt
