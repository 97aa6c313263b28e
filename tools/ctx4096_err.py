import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge
pkg = ge.load_pkg()
g = np.load("tests/golden/fullsize_ctx4096_v1.npz")
path = "/tmp/mi355-golden-llama-3-8b-q4_k_m.gguf"
if not os.path.exists(path):
    pkg.gguf_synth.write_synthetic_llama(path, "llama-3-8b", "q4_k_m", seed=0xC0FFEE, with_vocab=False)
pkg.Backend()
m = pkg.Model(path)
c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
n = int(g["n_prompt"])
prompt = np.random.default_rng(int(g["seed"])).integers(0, m.n_vocab, n).astype(np.int32)
for i0 in range(0, n, 2048):
    c.decode(prompt[i0:i0 + 2048], np.arange(i0, min(n, i0 + 2048)))
a = c.logits()
c.decode([int(g["tok_prompt"])], [n])
b = c.logits()
for got, key in ((a, "row_prompt"), (b, "row_step")):
    ref = g[key]
    print(key, "rel err", float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())), "argmax", int(got.argmax()), int(ref.argmax()), "median abs", float(np.median(np.abs(got - ref))))
