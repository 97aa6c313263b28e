mkdir -p gpurun_out
rm -f gpurun_out/r4_exp_early.txt
for v in early0 base early4 early0 base early4; do
  echo "=== $v" >> gpurun_out/r4_exp_early.txt
  timeout 300 tools/bin/exp_stream_$v 20 2>&1 | grep -E "MISMATCH|^qkv   q4k 4096\|1024\|1024 x4096 rmsnorm  |^gateup|^down  q4k 4096x14336 quant|^head|layer chain" | grep -v differ >> gpurun_out/r4_exp_early.txt
done
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_e.json 2> gpurun_out/r4_bench_e.err
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or mega_step or layer_engine or weight_stream" 2>&1 | tail -5) > gpurun_out/r4_t9_model.log
