import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_py as oq
pkg = ge.load_pkg()
path = "/tmp/dbg-e2048.gguf"
pkg.gguf_synth.write_synthetic_llama(path, "tiny-e2048", "q4_k_m", seed=11)
m = pkg.Model(path)
om = oq.OracleModel(path)
for n_prompt in (1, 4, 21, 40):
    c = pkg.Context(m, n_ctx=128, type_k=8, type_v=8, use_graphs=False)
    oc = oq.OracleContext(om, 128, 8, 8, True, 4)
    prompt = np.random.default_rng(5).integers(0, m.n_vocab, n_prompt)
    c.enable_taps(True)
    c.decode(prompt, np.arange(n_prompt))
    ref = oc.decode(prompt, np.arange(n_prompt))[0]
    for il in range(m.n_layer):
        a = c.layer_out(il, n_prompt).reshape(n_prompt, -1); b = oc.layer_out(il, n_prompt).reshape(n_prompt, -1)
        per_tok = np.abs(a - b).max(axis=1) / max(1.0, np.abs(b).max())
        print(f"n_prompt={n_prompt} layer {il}: per-token rel err", np.array2string(per_tok, precision=1, max_line_width=200))
    print("  logits err", float(np.abs(c.logits() - ref).max() / max(1.0, np.abs(ref).max())))
    c.close(); oc.close()
