#!/usr/bin/env python3
"""Mixtral-shaped prompt batch for profilers: usage: moe_prefill.py [tokens] [reps] [layers]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_pkg(); pkg.Backend()
gs = pkg.gguf_synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 4
c0 = gs.CONFIGS["mixtral-8x7b"]
cfg = gs.LlamaConfig(f"mixtral-{nl}l", c0.n_embd, nl, c0.n_head, c0.n_head_kv, c0.n_ff, c0.n_vocab, c0.rope_base, c0.eps, c0.n_ctx_train, c0.n_expert, c0.n_expert_used)
path = f"/tmp/mi355-mixtral-{nl}l-q5_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, cfg, "q5_k_m", seed=0xC0FFEE, with_vocab=False)
m = pkg.Model(path)
c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8, logits_to_host=False)
p = np.random.default_rng(0).integers(0, m.n_vocab, T)
for r in range(reps):
    c.kv_clear()
    t = time.perf_counter(); c.decode(p, np.arange(T)); c.argmax(); dt = time.perf_counter() - t
    print(f"prefill T={T} layers={nl}: {dt*1e3:.2f} ms = {dt*1e3/nl:.3f} ms / layer", flush=True)
tok = c.argmax()
for s in range(8):
    c.decode([tok], [T + s]); tok = c.argmax()
c.synchronize()
t = time.perf_counter()
for s in range(32):
    c.decode([tok], [T + 8 + s]); tok = c.argmax()
c.synchronize()
dt = time.perf_counter() - t
print(f"decode: {dt/32*1e3:.3f} ms / step = {dt/32*1e6/nl:.1f} us / layer (+ lm-head)")
c.close(); m.close()
