mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_step" 2>&1 | tail -25) > gpurun_out/r4_t1_ops.log
(timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "attn_out_one_launch or argmax_follows or (prefill_layers and (8b-2l or g8 or d128))" 2>&1 | tail -25) > gpurun_out/r4_t1_model.log
python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_fused.json 2> gpurun_out/r4_bench_fused.err
MI355_ATTN_OUT_FUSED=0 python bench.py --steps 128 --warmup 16 --no-cpu-baseline > gpurun_out/r4_bench_unfused.json 2> gpurun_out/r4_bench_unfused.err
MI355_AO_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_ao_probe.txt
MI355_ATTN_OUT_FUSED=0 MI355_ATTN_PROBE=1 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-long-context > /dev/null 2> gpurun_out/r4_attn_probe_old.txt
tail -3 gpurun_out/r4_t1_ops.log gpurun_out/r4_t1_model.log
cat gpurun_out/r4_bench_fused.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused', d['value'], d.get('long_context'), d.get('prefill_tok_s'))"
cat gpurun_out/r4_bench_unfused.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused', d['value'], d.get('long_context'), d.get('prefill_tok_s'))"
