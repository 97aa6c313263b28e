#!/bin/bash
# GPU box: parity of the layer engine against the per-launch path, then bench lines with and without it, then the probe's time line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "layer_engine or weight_stream or mega_step" 2>&1 | tail -15 > gpurun_out/engine_tests.txt
cat gpurun_out/engine_tests.txt
MI355_ENGINE=1 timeout 600 python bench.py > gpurun_out/bench_engine.json 2> gpurun_out/bench_engine.err
MI355_ENGINE=0 timeout 600 python bench.py > gpurun_out/bench_noengine.json 2> gpurun_out/bench_noengine.err
MI355_ENGINE=1 MI355_ENGINE_PROBE=5 MI355_ENGINE_PROBE_FILE=gpurun_out/engine_probe.bin timeout 600 python bench.py --steps 8 --warmup 4 > gpurun_out/bench_probe.json 2> gpurun_out/engine_probe.txt
python - <<'PY'
import json
for n in ("engine", "noengine"):
    try:
        d = json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("long_context", {}), d.get("prefill_tok_s"))
    except Exception as e:
        print(n, "failed", e)
PY
grep -A48 "engine probe" gpurun_out/engine_probe.txt | tail -49
