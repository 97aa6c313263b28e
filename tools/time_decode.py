#!/usr/bin/env python3
"""Decode tok/s of the bench model in this process (library: MI355_LLAMA_LIB or the tree's), for same-box A/B runs of two builds (tools/ab_libs.sh).
usage: time_decode.py [steps] [prompt_tokens]   -> prints "<tok/s at the prompt's end> <tok/s with the context filled to 3968>" """
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 192
n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
# (another BASELINE configuration: TD_CONFIG=llama-2-7b TD_FTYPE=q5_k_m TD_KV=f16)
cfg_name, ftype, kvt = os.environ.get("TD_CONFIG", "llama-3-8b"), os.environ.get("TD_FTYPE", "q4_k_m"), {"f16": 1, "q8_0": 8}[os.environ.get("TD_KV", "q8_0")]
path = f"/tmp/mi355-bench-{cfg_name}-{ftype}.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS[cfg_name], ftype, seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=kvt, type_v=kvt)


def run(n_p, n_steps):
    ctx.kv_clear()
    prompt = np.random.default_rng(1234).integers(0, model.n_vocab, n_p)
    for i0 in range(0, n_p, 2048):
        assert ctx.decode(prompt[i0:i0 + 2048], np.arange(i0, min(n_p, i0 + 2048))) == 0
    tok, pos = ctx.argmax(), n_p
    for _ in range(16):
        ctx.decode([tok], [pos]); ctx.logits_ready(); tok = ctx.argmax(); pos += 1
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        ctx.decode([tok], [pos]); ctx.logits_ready(); tok = ctx.argmax(); pos += 1
    ctx.synchronize()
    return n_steps / (time.perf_counter() - t0)


def prefill(n_p, reps=5):
    prompt = np.random.default_rng(99).integers(0, model.n_vocab, n_p)
    ts = []
    for _ in range(reps):
        ctx.kv_clear(); ctx.synchronize()
        t0 = time.perf_counter()
        assert ctx.decode(prompt, np.arange(n_p)) == 0
        ctx.logits_ready()
        ts.append(time.perf_counter() - t0)
    return n_p / sorted(ts)[len(ts) // 2]


a = run(n_prompt, steps)
b = run(3968, min(steps, 96))
p = prefill(512)
print(f"{a:.1f} {b:.1f} {p:.0f}")
ctx.close(); model.close()
