mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_model.py -q -k "sequence_region or (prefill_layers and r3) or batched or parallel or neox" 2>&1 | tail -n 6) > gpurun_out/r4_t11.log
(timeout 1500 python -m pytest tests/test_gpu_engine.py -q 2>&1 | tail -n 4) >> gpurun_out/r4_t11.log
cat gpurun_out/r4_t11.log
