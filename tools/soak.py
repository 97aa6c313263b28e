#!/usr/bin/env python3
"""A few minutes of sustained work to flush out rare failures (a bounded in-kernel wait that gives up, a scheduler race): N single-token steps on the bench model,
then four parallel greedy chat streams through the engine for a while.  Prints what it did; exits non-zero on any error.  usage: soak.py [steps] [seconds]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_pkg()
gs = pkg.gguf_synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
prompt = np.random.default_rng(1).integers(0, model.n_vocab, 512)
assert ctx.decode(prompt, np.arange(512)) == 0
tok, pos, done = ctx.argmax(), 512, 0
t0 = time.time()
while done < steps:
    n = min(3000, steps - done, 4090 - pos)
    toks = ctx.greedy_steps(tok, pos, n)
    tok, pos, done = int(toks[-1]), pos + n, done + n
    if pos >= 4090:                                   # context full: start over (prompt + steps again)
        ctx.kv_clear()
        assert ctx.decode(prompt, np.arange(512)) == 0
        tok, pos = ctx.argmax(), 512
print(f"{done} single-token steps in {time.time() - t0:.1f} s: ok", flush=True)
ctx.close(); model.close()

tiny = "/tmp/mi355-soak-tiny.gguf"
gs.write_synthetic_llama(tiny, "tiny-d128", "q4_k_m", with_vocab=True)
e = pkg.Engine()
st, body = e.load_model(llama_model_path=tiny, model="t", ctx_len=1024, n_parallel=4, user_prompt="u:", ai_prompt="a:")
assert st["status_code"] == 200, (st, body)
stop = time.time() + seconds
counts, errors = [0, 0, 0, 0], []


def worker(i):
    k = 0
    while time.time() < stop:
        st, body = e.chat_completion(model="t", messages=[{"role": "user", "content": f"request {i} {k} " + "x " * (k % 37)}], max_tokens=16 + (k % 48),
                                     temperature=0.0 if k % 2 else 0.8, seed=k)[-1]
        if st["status_code"] != 200 or st["has_error"]:
            errors.append((i, k, st, body))
            return
        counts[i] += 1
        k += 1


th = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
for t in th:
    t.start()
for t in th:
    t.join()
e.close()
print(f"engine: {sum(counts)} requests over 4 slots in {seconds:.0f} s, errors: {errors[:2]}", flush=True)
sys.exit(1 if errors else 0)
