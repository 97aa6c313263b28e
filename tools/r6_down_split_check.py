#!/usr/bin/env python3
"""Round 6: ffn_down of a single-token step as two column halves in one launch (mmvq_stream_ffn_down_split) against the whole-row launch, on the bench model:
logits of both forms against each other, per-role kernel times of both (kernel begin / end timestamps, Context profile mode), decode tok/s of both.
usage: r6_down_split_check.py [steps]      (needs a library built with tools/r6_down_split.patch applied: the option "down_split" does not exist in the tree)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
be = pkg.Backend()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
prompt = np.random.default_rng(1234).integers(0, model.n_vocab, 512)


def run(split):
    be.set_option("down_split", 1 if split else 0)
    ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
    assert ctx.decode(prompt, np.arange(512)) == 0
    rows = [ctx.logits().copy()]
    tok, pos = int(rows[0].argmax()), 512
    toks = []
    for _ in range(24):
        assert ctx.decode([tok], [pos]) == 0
        rows.append(ctx.logits().copy()); toks.append(tok)
        tok = int(rows[-1].argmax()); pos += 1
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.decode([tok], [pos]); ctx.logits_ready(); tok = ctx.argmax(); pos += 1
    ctx.synchronize()
    rate = steps / (time.perf_counter() - t0)
    acc = {}
    ctx.profile(True)
    for _ in range(16):
        ctx.decode([tok], [pos]); tok = ctx.argmax(); pos += 1
        for k, v in ctx.last_profile().items():
            acc[k] = acc.get(k, 0.0) + v / 16
    ctx.profile(False)
    ctx.close()
    return np.stack(rows), toks, rate, acc


r1, t1, rate1, p1 = run(True)
r0, t0_, rate0, p0 = run(False)
err = max(float(np.abs(a - b).max() / max(1.0, np.abs(b).max())) for a, b in zip(r1, r0))
print(f"split vs whole rows: max rel err over 25 rows {err:.3e}; greedy tokens equal: {t1 == t0_}")
print("per row:", " ".join(f"{float(np.abs(a - b).max() / max(1.0, np.abs(b).max())):.1e}" for a, b in zip(r1, r0)))
print(f"decode tok/s: split {rate1:.1f}  whole {rate0:.1f}")
for k in sorted(set(p1) | set(p0)):
    if k.startswith("k:") or k.startswith("n:"):
        print(f"  {k:<14} split {p1.get(k, 0):8.2f}   whole {p0.get(k, 0):8.2f}")
