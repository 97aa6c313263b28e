"""Debug: greedy decode with n_seq_max 1 vs 2 (regions + chunk lists) on the tiny model; prints where logits diverge."""
import sys, tempfile, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_pkg()
path = os.path.join(tempfile.mkdtemp(), "t.gguf")
pkg.gguf_synth.write_synthetic_llama(path, "tiny-d128", "q4_k_m", with_vocab=True)
m = pkg.Model(path)
toks = m.tokenize("s:be briefu:hello worlda:", add_special=True, parse_special=True)
print("prompt tokens", len(toks))
def run(nseq, seq, steps=20, **kw):
    c = pkg.Context(m, n_ctx=512, n_seq_max=nseq, **kw)
    assert c.decode(toks, list(range(len(toks))), seq=seq) == 0
    out, lg, pos = [], [], len(toks)
    for _ in range(steps):
        l = np.array(c.logits(-1), dtype=np.float32)
        t = int(np.argmax(l)); out.append(t); lg.append(l)
        assert c.decode([t], [pos], seq=seq) == 0
        pos += 1
    c.close()
    return out, lg
a, la = run(1, 0, steps=4)
G = os.environ.get('DBG_GRAPHS', '1') == '1'
for nseq, seq in ((2, 1),):
    b, lb = run(nseq, seq, steps=4, use_graphs=G)
    print(nseq, seq, "tokens equal:", a == b)
    for i, (x, y) in enumerate(zip(la, lb)):
        d = float(np.abs(x - y).max())
        print("  step", i, "maxdiff %.3g" % d, a[i], b[i])
        if a[i] != b[i]: break
