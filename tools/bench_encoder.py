"""Embedding throughput of the encoder graph on one GPU: nomic-embed-text-v1.5's geometry (synthetic weights), batches of whole sequences through mi355_decode
with every row flagged.  Usage: python tools/bench_encoder.py [ftype=f16] [tokens per sequence=512] [sequences per batch=4]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import __graft_entry__ as ge
pkg = ge.load_pkg()
ftype = sys.argv[1] if len(sys.argv) > 1 else "f16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nseq = int(sys.argv[3]) if len(sys.argv) > 3 else 4
path = f"/tmp/nomic-embed-{ftype}.gguf"
if not os.path.exists(path):
    pkg.gguf_synth.write_synthetic_llama(path, "nomic-embed", ftype, seed=1)
pkg.Backend()
m = pkg.Model(path)
c = pkg.Context(m, n_ctx=n * nseq, n_batch=n * nseq, n_ubatch=n * nseq, n_seq_max=max(nseq, 1), type_k=1, type_v=1)
rng = np.random.default_rng(0)
toks = rng.integers(5, m.n_vocab, n * nseq)
pos = np.tile(np.arange(n), nseq)
seq = np.repeat(np.arange(nseq), n)
flags = np.ones(n * nseq, np.int8)
for it in range(3):
    c.kv_clear(); assert c.decode(toks, pos, list(seq), flags) == 0; c.embeddings(0)
t = time.perf_counter(); K = 10
for it in range(K):                 # (the batch object of the warm-up call is reused: Python fills it token by token, which is not what is measured)
    c.kv_clear(); assert c.lib.mi355_decode(c.h, c._b) == 0; c.embeddings(0)
dt = (time.perf_counter() - t) / K
print(f"{ftype}: {nseq} x {n} tokens per batch: {dt * 1e3:.2f} ms -> {n * nseq / dt:.0f} tok/s")
