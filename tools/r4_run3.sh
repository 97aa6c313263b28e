mkdir -p gpurun_out
rm -f gpurun_out/r4_exp_nc.txt
for v in base nc14 nc12 pair nc14pair base; do
  echo "=== $v" >> gpurun_out/r4_exp_nc.txt
  timeout 300 tools/bin/exp_stream_$v 20 2>&1 | grep -E "bit-identical|differ|MISMATCH|qkv|^o |gateup|down|layer chain|lm-head|head" >> gpurun_out/r4_exp_nc.txt
done
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_step" 2>&1 | tail -5) > gpurun_out/r4_t3_ops.log
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or argmax_follows" 2>&1 | tail -8) > gpurun_out/r4_t3_model.log
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_fused2.json 2> gpurun_out/r4_bench_fused2.err
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe2.txt
MI355_NO_GRAPHS=1 MI355_AO_PROBE=1 python bench.py --steps 4 --warmup 2 --prompt 3960 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe2_long.txt
