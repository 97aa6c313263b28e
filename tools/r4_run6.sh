mkdir -p gpurun_out
rm -f gpurun_out/r4_exp_wide.txt
for v in base base; do
  echo "=== $v" >> gpurun_out/r4_exp_wide.txt
  timeout 300 tools/bin/exp_stream_$v 20 2>&1 | grep -E "differ|MISMATCH|identical|^qkv|^o  |^gateup|^down|^head|layer chain" >> gpurun_out/r4_exp_wide.txt
done
(timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -8) > gpurun_out/r4_t6_ops.log
(timeout 1500 python -m pytest tests/test_gpu_model.py -q -x -k "weight_stream or attn_out_one_launch or argmax_follows or mega_step or layer_engine or (prefill_layers and (8b-2l or g8 or e2048 or d128))" 2>&1 | tail -8) > gpurun_out/r4_t6_model.log
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_wide.json 2> gpurun_out/r4_bench_wide.err
