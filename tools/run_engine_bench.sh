#!/bin/bash
# engine-level throughput (slot loop + tokenizer + sampler + JSON around the decode): one user, then the reference's 40-user load shape
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -x -q -k "topk or device_sampling or grammar or determinism_graph_equals_eager_and_causality" 2>&1 | tail -3
python tools/bench_engine.py --users 1 --rounds 2 --max-tokens 256 --n-parallel 1 --ctx-per-seq 2048 --greedy
python tools/bench_engine.py --users 1 --rounds 2 --max-tokens 256 --n-parallel 1 --ctx-per-seq 2048 --device-sampling 1
python tools/bench_engine.py --users 1 --rounds 2 --max-tokens 256 --n-parallel 1 --ctx-per-seq 2048 --device-sampling 0
python tools/bench_engine.py --users 40 --rounds 2 --max-tokens 200 --n-parallel 32 --device-sampling 1
python tools/bench_engine.py --users 40 --rounds 2 --max-tokens 200 --n-parallel 32 --device-sampling 0
python tools/bench_engine.py --users 40 --rounds 2 --max-tokens 200 --n-parallel 32 --greedy
} 2>&1 | grep -v "UTC" | tee gpurun_out/engine_bench.txt
