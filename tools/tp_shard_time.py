#!/usr/bin/env python3
"""One rank's share of a row-split step, without communication: the process forms a null group of N ranks (exchanges = device
copies of its own part), loads rank 0's slice and times prompt + single-token steps.  usage: tp_shard_time.py <config> <ftype> <N> [prompt]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
gs = pkg.gguf_synth
cfg, ftype, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
P = int(sys.argv[4]) if len(sys.argv) > 4 else 512
path = f"/tmp/mi355-bench-{cfg}-{ftype}.gguf"
if not os.path.exists(path):
    gs.write_synthetic_llama(path, gs.CONFIGS[cfg], ftype, seed=0xC0FFEE, with_vocab=False)
if N > 1:
    be.set_option("tp_null_group", N)
t0 = time.time()
m = pkg.Model(path, tp_rank=0, tp_size=N)
c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8, logits_to_host=False)
print(f"{cfg} {ftype} rank 0 of {N}: heads {m.n_head}/{m.n_head_kv}, {m.bytes_per_token / 1e9:.3f} GB per token per rank, load {time.time() - t0:.1f} s", flush=True)
prompt = np.random.default_rng(1).integers(0, m.n_vocab, P)
for rep in range(2):
    c.kv_clear(); t = time.perf_counter(); c.decode(prompt, np.arange(P)); tok = c.argmax(); tp = time.perf_counter() - t
for s in range(16):
    c.decode([tok], [P + s]); tok = c.argmax()
c.synchronize(); t = time.perf_counter()
for s in range(64):
    c.decode([tok], [P + 16 + s]); tok = c.argmax()
c.synchronize(); dt = (time.perf_counter() - t) / 64
print(f"  prompt {P}: {tp * 1e3:.1f} ms ({P / tp:.0f} tok/s); step {dt * 1e3:.3f} ms ({1 / dt:.0f} tok/s) compute only")
c.close(); m.close()
