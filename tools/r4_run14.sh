mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_model.py -q -k "weight_stream" 2>&1 | tail -n 8) > gpurun_out/r4_t14.log
(timeout 1500 python -m pytest tests/test_gpu_ops.py -q -k "mul_mat" 2>&1 | tail -n 4) >> gpurun_out/r4_t14.log
(timeout 1500 python -m pytest tests/test_gpu_model.py -q -k "prefill_layers and (q2_k or q3_k or q4_0 or q5_0 or iq4_nl or q8_0)" 2>&1 | tail -n 4) >> gpurun_out/r4_t14.log
cat gpurun_out/r4_t14.log
for spec in "tinyllama-1.1b q2_k f16" "tinyllama-1.1b q8_0 f16" "llama-3-8b q8_0 q8_0" "llama-3-8b q2_k q8_0"; do
  set -- $spec
  python bench.py --config $1 --ftype $2 --cache-type $3 --steps 128 --warmup 16 --no-cpu-baseline --no-long-context 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$spec stream', d['value'], d['prefill_tok_s'])"
  MI355_MMVQ_STREAM=0 python bench.py --config $1 --ftype $2 --cache-type $3 --steps 128 --warmup 16 --no-cpu-baseline --no-long-context 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$spec ring  ', d['value'], d['prefill_tok_s'])"
done 2>&1 | tee gpurun_out/r4_formats_bench.txt
