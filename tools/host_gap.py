#!/usr/bin/env python3
"""How much of a decode step is the HOST's (round 5)?  The same single-token step (a) as the bench runs it - launch, wait for the logits row, read the arg-max, next -
(b) with the logits kept on the device (device arg-max only), (c) enqueued back to back without waiting in between (the numbers are garbage - every step gets the
same token - but the GPU never idles: this is the step's pure device time including graph-to-graph gaps).  usage: host_gap.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 192
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)


def run(mode, to_host):
    ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8, logits_to_host=to_host)
    prompt = np.random.default_rng(1234).integers(0, model.n_vocab, 512)
    assert ctx.decode(prompt, np.arange(512)) == 0
    tok, pos = ctx.argmax(), 512
    for _ in range(16):
        ctx.decode([tok], [pos]); tok = ctx.argmax(); pos += 1
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.decode([tok], [pos])
        if mode == "sync":
            if to_host:
                ctx.logits_ready()
            tok = ctx.argmax()
        pos += 1
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.close()
    return dt / steps * 1e6


for rep in range(2):
    a, b, c = run("sync", True), run("sync", False), run("async", False)
    print(f"us per step: host-visible logits {a:.1f} | device arg-max only {b:.1f} | back to back, no wait {c:.1f}   -> host share {a - c:.1f} us ({(a - c) / a * 100:.1f} %)")
model.close()
