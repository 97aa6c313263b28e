mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -k "two_split_merge or long_prompt" 2>&1 | tail -n 8) > gpurun_out/r4_t16.log
(timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "flash_attn" 2>&1 | tail -n 4) >> gpurun_out/r4_t16.log
(timeout 1400 python -m pytest tests/test_gpu_tp.py -x -q -k "8-8 or 8-0" 2>&1 | tail -n 12) >> gpurun_out/r4_t16.log
cat gpurun_out/r4_t16.log
for v in 1 0 1 0; do MI355_FA_MERGE_FUSED=$v python bench.py --steps 32 --warmup 4 --no-cpu-baseline --no-long-context 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fused=$v', d['value'], d['prefill_tok_s'], d['prefill_ms'])"; done
