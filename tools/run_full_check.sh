#!/bin/bash
# GPU box: the whole -m gpu suite, then the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/gpu_tests.txt 2>&1
cat gpurun_out/gpu_tests.txt
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "prefill", d["prefill_tok_s"], "greedy", d["decode_tok_s_device_greedy"], "long", d["long_context"], "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "cpu", d["cpu_baseline"])
PY
