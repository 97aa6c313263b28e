mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_golden_ops.py tests/test_gpu_model.py -q -m gpu -k "golden or forced_routing or moe" 2>&1 | tail -n 8) > gpurun_out/r4_t13.log
(timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -k "other_configs_logits and mixtral" 2>&1 | tail -n 8) >> gpurun_out/r4_t13.log
cat gpurun_out/r4_t13.log
