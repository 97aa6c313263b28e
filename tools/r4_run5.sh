mkdir -p gpurun_out
rm -f gpurun_out/r4_exp_planes.txt
for v in base nomins base nomins; do
  echo "=== $v" >> gpurun_out/r4_exp_planes.txt
  timeout 300 tools/bin/exp_stream_$v 20 2>&1 | grep -E "^qkv   q4k 4096\|1024\|1024 x4096 rmsnorm  |^o     q4k 4096x4096 planes \+resid      |^gateup|^down|layer chain" | grep -v differ >> gpurun_out/r4_exp_planes.txt
done
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_step" 2>&1 | tail -5) > gpurun_out/r4_t5_ops.log
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or argmax_follows" 2>&1 | tail -30) > gpurun_out/r4_t5_model.log
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_fused4.json 2> gpurun_out/r4_bench_fused4.err
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe4.txt
