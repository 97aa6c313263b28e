#!/usr/bin/env python3
"""Per-kernel-class device time of one prefill micro-batch and one decode step (HIP events between launches, eager)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_pkg(); pkg.Backend()
cfg, ftype, kv = os.environ.get("CONFIG", "llama-3-8b"), os.environ.get("FTYPE", "q4_k_m"), int(os.environ.get("KV", "8"))   # KV: 8 = q8_0, 1 = f16
path = f"/tmp/mi355-bench-{cfg}-{ftype}.gguf"
if not os.path.exists(path):
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=0xC0FFEE, with_vocab=False)
m = pkg.Model(path)
c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=int(os.environ.get("UBATCH", "2048")), type_k=kv, type_v=kv, use_graphs=False)
rng = np.random.default_rng(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 512
p = rng.integers(0, m.n_vocab, T)
c.decode(p, np.arange(T)); c.synchronize(); c.kv_clear()
c.profile(True)
t = time.perf_counter(); c.decode(p, np.arange(T)); c.synchronize(); dt = time.perf_counter() - t
prof = c.last_profile()
print(f"prefill T={T}: {dt*1e3:.1f} ms wall; device by class (us):")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]): print(f"  {k:14s} {v:10.1f}")
c.decode([1], [T]); c.synchronize()
prof = c.last_profile()
print("decode step (eager, event-to-event incl. launch gaps) by class (us):")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]): print(f"  {k:14s} {v:10.1f}")
