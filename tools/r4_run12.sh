mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/test_gpu_model.py -q -k "context_filled" 2>&1 | tail -n 12) > gpurun_out/r4_t12.log
(timeout 1800 python -m pytest tests/test_gpu_fullsize.py -q -k "context_filled" 2>&1 | tail -n 12) >> gpurun_out/r4_t12.log
python bench.py --steps 128 --warmup 16 > gpurun_out/r4_bench_12.json 2> gpurun_out/r4_bench_12.err
cat gpurun_out/r4_t12.log; tail -n 1 gpurun_out/r4_bench_12.json | cut -c1-1500
