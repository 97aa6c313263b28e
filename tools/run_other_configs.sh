#!/bin/bash
# bench lines of the other BASELINE configurations on one GPU (and the reference's smoke-model type mix), one JSON line each under gpurun_out/r6cfg/
cd "$(dirname "$0")/.."
O=gpurun_out/r6cfg; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; tail -c 300 $O/$name.json | head -c 0; python - "$O/$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "decode", d["value"], "prefill", d["prefill_tok_s"], "long", d.get("long_context", {}).get("decode_tok_s"), d.get("long_context", {}).get("prefill_tok_s"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run llama-3-8b-q8_0 --config llama-3-8b --ftype q8_0
run llama-2-7b --config llama-2-7b --ftype q5_k_m --cache-type f16
run tinyllama-1.1b --config tinyllama-1.1b --ftype q8_0 --cache-type f16
run tinyllama-1.1b-q2_k --config tinyllama-1.1b --ftype q2_k --cache-type f16
run tinyllama-1.1b-q4_k_m --config tinyllama-1.1b --ftype q4_k_m --cache-type f16
run mixtral-8x7b --config mixtral-8x7b --ftype q5_k_m
run llama-3-70b --config llama-3-70b --ftype q4_k_m
# (a leased box keeps /tmp between calls: the 70B and Mixtral files are 75 GB of an 80 GB disk)
rm -f /tmp/mi355-bench-llama-3-70b* /tmp/mi355-bench-mixtral* /tmp/mi355-bench-tinyllama* /tmp/mi355-bench-llama-2-7b* /tmp/mi355-bench-llama-3-8b-q8_0*
