#!/usr/bin/env python3
"""Decode-step breakdown on the bench workload (Llama-3-8B Q4_K_M synthetic, cache q8_0, prompt 512):
graph-mode ms/token, then eager-mode per-op-class device time from HIP events (mi355_profile_last_decode).
Usage (GPU box):  python tools/profile_decode.py [--config llama-3-8b] [--steps 64]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="llama-3-8b")
    ap.add_argument("--ftype", default="q4_k_m")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--cache-type", default="q8_0")
    args = ap.parse_args()
    pkg = ge.load_pkg()
    gs = pkg.gguf_synth
    path = f"/tmp/mi355-bench-{args.config}-{args.ftype}.gguf"
    if not os.path.exists(path):
        gs.write_synthetic_llama(path, gs.CONFIGS[args.config], args.ftype, seed=0xC0FFEE, with_vocab=False)
    KV = {"f16": 1, "q8_0": 8}[args.cache_type]
    model = pkg.Model(path)
    rng = np.random.default_rng(1234)
    prompt = rng.integers(0, model.n_vocab, args.prompt)
    for graphs in (True, False):
        ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=512, type_k=KV, type_v=KV, use_graphs=graphs, logits_to_host=False)
        assert ctx.decode(prompt, np.arange(args.prompt)) == 0
        tok, pos = ctx.argmax(), args.prompt
        for _ in range(8):
            assert ctx.decode([tok], [pos]) == 0
            tok = ctx.argmax(); pos += 1
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            assert ctx.decode([tok], [pos]) == 0
            tok = ctx.argmax(); pos += 1
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(f"{'graph' if graphs else 'eager'}: {dt * 1e3:.3f} ms/token  {1 / dt:.1f} tok/s", flush=True)
        if not graphs:
            ctx.profile(True)
            agg = {}
            n = 8
            for _ in range(n):
                assert ctx.decode([tok], [pos]) == 0
                tok = ctx.argmax(); pos += 1
                for k, v in ctx.last_profile().items():
                    agg[k] = agg.get(k, 0.0) + v
            tot = sum(agg.values()) / n
            nl = model.n_layer
            print(f"eager per-op (HIP events, includes launch gaps): total {tot:.1f} us/token")
            for k, us in sorted(agg.items(), key=lambda kv: -kv[1]):
                print(f"  {k:14s} {us / n:9.1f} us/token  {us / n / nl:7.2f} us/layer")
            us, by = ctx.weight_sweep_us(5)
            print(f"weight sweep: {us:.1f} us, {by / us / 1e3:.1f} GB/s")
        ctx.close()
    model.close()


if __name__ == "__main__":
    main()
