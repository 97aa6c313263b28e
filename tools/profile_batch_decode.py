#!/usr/bin/env python3
"""Continuous-batching decode: B sequences, one token each per step (what UpdateSlots submits with n_parallel = B).
Prints ms/step and aggregate tok/s, plus the eager per-op breakdown.  usage: tools/profile_batch_decode.py [B] [prompt]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
P = int(sys.argv[2]) if len(sys.argv) > 2 else 128
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
n_ctx = B * (P + 96)
ctx = pkg.Context(model, n_ctx=n_ctx, n_batch=2048, n_ubatch=512, n_seq_max=B, type_k=8, type_v=8, logits_to_host=False, use_graphs=os.environ.get("GRAPHS", "1") == "1")
rng = np.random.default_rng(1)
for s in range(B):
    prompt = rng.integers(0, model.n_vocab, P)
    assert ctx.decode(prompt, np.arange(P), seq=[s] * P) == 0
toks = [int(rng.integers(0, model.n_vocab)) for _ in range(B)]
def step(pos):
    assert ctx.decode(toks, [pos] * B, seq=list(range(B)), logits=[1] * B) == 0
    return [ctx.argmax(i) for i in range(B)]
for i in range(4):
    toks = step(P + i)
ctx.synchronize()
n = 32
t0 = time.perf_counter()
for i in range(n):
    toks = step(P + 4 + i)
ctx.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"B={B} prompt={P}: {dt * 1e3:.3f} ms/step, aggregate {B / dt:.0f} tok/s ({1 / dt:.1f} steps/s)")
ctx.profile(True)
toks = step(P + 4 + n)
for k, v in sorted(ctx.last_profile().items(), key=lambda kv: -kv[1]):
    print(f"  {k:14s} {v:9.1f} us")
