import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_pkg(); gs = pkg.gguf_synth
path = "/tmp/mi355-bench-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS["llama-3-8b"], "q4_k_m", seed=0xC0FFEE, with_vocab=False)
model = pkg.Model(path)
ctx = pkg.Context(model, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
prompt = np.random.default_rng(1234).integers(0, model.n_vocab, 512)
for rep in range(3):
    for mode in ("py", "c"):
        ctx.kv_clear()
        assert ctx.decode(prompt, np.arange(512)) == 0
        tok, pos = ctx.argmax(), 512
        toks = []
        for _ in range(16):
            ctx.decode([tok], [pos]); ctx.logits_ready(); tok = ctx.argmax(); pos += 1
        ctx.synchronize()
        t0 = time.perf_counter()
        if mode == "py":
            for _ in range(192):
                ctx.decode([tok], [pos]); ctx.logits_ready(); tok = ctx.argmax(); pos += 1; toks.append(tok)
        else:
            toks = list(ctx.greedy_steps(tok, pos, 192))
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print(mode, round(192 / dt, 1), "tok/s", "hash", hash(tuple(toks)) & 0xffff)
