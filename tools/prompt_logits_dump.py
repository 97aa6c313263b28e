#!/usr/bin/env python3
"""tools/prompt_logits_dump.py <out.npy> [n_prompt]: logits of a seeded n_prompt-token prompt on the bench model (llama-3-8b q4_k_m, q8_0 cache) - for bitwise A/Bs
between environment settings or library builds: run it twice, compare the files."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
tok = np.random.default_rng(1).integers(0, model.n_vocab, n)
assert ctx.decode(tok, np.arange(n)) == 0
np.save(out, ctx.logits())
print("saved", out)
