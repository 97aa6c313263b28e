#!/usr/bin/env python3
"""Minimal decode loop for profilers: prompt 512 then N single-token steps on the bench model (no timing, no output)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=512, type_k=8, type_v=8, logits_to_host=False)
prompt = np.random.default_rng(1234).integers(0, model.n_vocab, 512)
assert ctx.decode(prompt, np.arange(512)) == 0
tok, pos = ctx.argmax(), 512
for _ in range(n):
    assert ctx.decode([tok], [pos]) == 0
    tok = ctx.argmax(); pos += 1
ctx.synchronize()
ctx.close(); model.close()
