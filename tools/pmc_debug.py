#!/usr/bin/env python3
"""Prints prompt / decode outputs of a small synthetic model (to diff a plain run against a profiled one)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
cfgname = sys.argv[1] if len(sys.argv) > 1 else "tiny-d128"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
path = f"/tmp/pmcdbg-{cfgname}.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS[cfgname], "q4_k_m", seed=7, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=1024, n_batch=512, n_ubatch=512, type_k=8, type_v=8, logits_to_host=True)
prompt = np.random.default_rng(1).integers(0, model.n_vocab, T)
assert ctx.decode(prompt, np.arange(T)) == 0
lg = ctx.logits(-1)
print("prefill argmax", ctx.argmax(), "logits[:4]", lg[:4], "nan", int(np.isnan(lg).sum()))
tok, pos = int(np.nanargmax(lg)), T
for _ in range(4):
    assert ctx.decode([tok], [pos]) == 0
    lg = ctx.logits(-1)
    print("decode argmax", ctx.argmax(), "host argmax", int(np.nanargmax(lg)), "nan", int(np.isnan(lg).sum()))
    tok = int(np.nanargmax(lg)); pos += 1
