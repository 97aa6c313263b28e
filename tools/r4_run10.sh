REPS=4 bash tools/ab_libs.sh s32c32=tools/bin/libs/s32c32.so gpb=tools/bin/libs/gpb.so
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_step or flash_attn" 2>&1 | tail -3) > gpurun_out/r4_t10_ops.log
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or mega_step or layer_engine or argmax_follows" 2>&1 | tail -3) > gpurun_out/r4_t10_model.log
