mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_model.py -q -x -k "tiny-ff28k" 2>&1 | tail -n 8) > gpurun_out/r4_t17.log
cat gpurun_out/r4_t17.log
python bench.py --config llama-3-70b --ftype q4_k_m --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/r4_70b_halves.json 2>gpurun_out/r4_70b.err
MI355_DOWN_HALVES=0 python bench.py --config llama-3-70b --ftype q4_k_m --steps 64 --warmup 8 --no-cpu-baseline --no-long-context > gpurun_out/r4_70b_whole.json 2>>gpurun_out/r4_70b.err
for f in gpurun_out/r4_70b_halves.json gpurun_out/r4_70b_whole.json; do python - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "decode", d["value"], "prefill", d["prefill_tok_s"], "frac", d["roofline"]["frac"], "long", d.get("long_context", {}).get("decode_tok_s"))
PY
done
tail -n 3 gpurun_out/r4_70b.err
