#!/usr/bin/env python3
"""Round 6: phase stamps of the attention-block launch (MI355_AO_PROBE=1, eager launches: the probe allocates) on the bench model at a given position.
usage: MI355_AO_PROBE=1 [MI355_QKV_ATTN_FUSED=0] python tools/r6_qf_probe.py [prompt_tokens]   -> the time line on stderr when the context closes"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg_name, ftype, kvt = os.environ.get("TD_CONFIG", "llama-3-8b"), os.environ.get("TD_FTYPE", "q4_k_m"), {"f16": 1, "q8_0": 8}[os.environ.get("TD_KV", "q8_0")]
path = f"/tmp/mi355-bench-{cfg_name}-{ftype}.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS[cfg_name], ftype, seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=kvt, type_v=kvt, use_graphs=False)
prompt = np.random.default_rng(1234).integers(0, model.n_vocab, n_prompt)
for i0 in range(0, n_prompt, 2048):
    assert ctx.decode(prompt[i0:i0 + 2048], np.arange(i0, min(n_prompt, i0 + 2048))) == 0
tok, pos = ctx.argmax(), n_prompt
for _ in range(48):
    ctx.decode([tok], [pos]); tok = ctx.argmax(); pos += 1
ctx.synchronize()
print("qkv_attn_launches", ctx.qkv_attn_launches())
ctx.close(); model.close()
