#!/usr/bin/env python3
"""Soak test of the prompt path: random prompt lengths through the LDS-form kernels (unsplit) against the per-lane planes kernels, bit for
bit, and the same prompt twice through the default path (run-to-run determinism, K split and attention splits included).
usage: tools/stress_prefill.py [iterations] [cfg]"""
import os, sys, pathlib, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_model import make
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = sys.argv[2] if len(sys.argv) > 2 else "tiny-8b-2l"
d = pathlib.Path(tempfile.mkdtemp())
rng = np.random.default_rng(12345)
bad = 0
for ftype in ("q4_k_m", "q5_k_m"):
    m = pkg.Model(make(pkg, d, cfg, ftype))
    def run(prompt, lds, split, ub):
        be.set_option("mmq_lds_form", lds); be.set_option("mmq_split", split)
        c = pkg.Context(m, n_ctx=2304, n_batch=2048, n_ubatch=ub, type_k=8, type_v=8)
        assert c.decode(prompt, np.arange(len(prompt))) == 0
        out = c.logits().copy(); c.close()
        be.set_option("mmq_lds_form", -1); be.set_option("mmq_split", 0)
        return out
    for it in range(iters):
        T = int(rng.integers(257, 2049))
        ub = int(rng.choice([512, 1024, 2048]))
        prompt = rng.integers(0, m.n_vocab, T)
        a, b = run(prompt, 1, 1, ub), run(prompt, 0, 0, ub)
        c1, c2 = run(prompt, -1, 0, ub), run(prompt, -1, 0, ub)
        ok1, ok2 = np.array_equal(a, b), np.array_equal(c1, c2)
        if not (ok1 and ok2 and np.isfinite(c1).all()):
            bad += 1
            print(f"MISMATCH {ftype} T={T} ubatch={ub}: lds-vs-old {ok1} ({np.abs(a - b).max():.3g}), run-to-run {ok2} ({np.abs(c1 - c2).max():.3g})", flush=True)
    m.close()
    print(f"{ftype}: {iters} prompts done", flush=True)
print("stress: %d mismatches" % bad)
sys.exit(1 if bad else 0)
