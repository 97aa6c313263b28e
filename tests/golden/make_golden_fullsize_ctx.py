#!/usr/bin/env python3
"""Generates tests/golden/fullsize_ctx4096_v1.npz: the CPU ORACLE's logits for the headline model with its context filled - the synthetic Llama-3-8B Q4_K_M file (a pure
function of config and seed: gguf_synth), q8_0 cache, a seeded 3968-token prompt and one teacher-forced step (BASELINE's ctx_len 4096).  32 layers x 3968 tokens take
the scalar restatement many minutes on 32+ threads - too long for a test run - so the answer is computed ONCE by this script and committed; the GPU suite compares the
device path with it (tests/test_gpu_fullsize.py::test_fullsize_context_filled_matches_committed_oracle_logits).  Stored: the prompt's seed and length, the two logits
rows (f32), the oracle's top token of each.
usage: python tests/golden/make_golden_fullsize_ctx.py [n_prompt] [threads] [c3 | c2]      (needs oracle/ built; writes beside this file)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle_py as oq  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

SEED, N_PROMPT = 40961, 3968


# which = "c3": BASELINE config 3 (the headline): Llama-3-8B Q4_K_M, q8_0 cache -> fullsize_ctx4096_v1.npz
#         "c2": BASELINE config 2: Llama-2-7B Q5_K_M, f16 cache, the oracle in its STOCK mode (V accumulated in fp16, as the reference's CPU path) -> fullsize_ctx4096_c2_v1.npz
CASES = {"c3": ("llama-3-8b", "q4_k_m", "Q8_0", "fullsize_ctx4096_v1.npz"), "c2": ("llama-2-7b", "q5_k_m", "F16", "fullsize_ctx4096_c2_v1.npz")}


def main():
    n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else N_PROMPT
    nth = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(os.cpu_count() or 1, 128))
    which = sys.argv[3] if len(sys.argv) > 3 else "c3"
    cfg, ftype, kvname, out_name = CASES[which]
    kv = getattr(oq, kvname)
    gs = ge.load_pkg().gguf_synth
    path = f"/tmp/mi355-golden-{cfg}-{ftype}.gguf"
    if not os.path.exists(path):
        gs.write_synthetic_llama(path, cfg, ftype, seed=0xC0FFEE, with_vocab=False)
    om = oq.OracleModel(path)
    oq.set_fa_v_acc_f32(0)
    oc = oq.OracleContext(om, 4096, kv, kv, True, nth)
    prompt = np.random.default_rng(SEED).integers(0, om.n_vocab, n_prompt).astype(np.int32)
    t0 = time.time()
    row0 = oc.decode(prompt, np.arange(n_prompt))[0]
    t1 = time.time()
    tok = int(row0.argmax())
    row1 = oc.decode([tok], [n_prompt])[0]
    print(f"prompt of {n_prompt}: {t1 - t0:.1f} s on {nth} threads; step {time.time() - t1:.1f} s; top tokens {tok}, {int(row1.argmax())}", flush=True)
    out = os.path.join(HERE, out_name if n_prompt == N_PROMPT else f"fullsize_ctx_{which}_{n_prompt}.npz")
    np.savez_compressed(out, seed=SEED, n_prompt=n_prompt, row_prompt=row0.astype(np.float32), row_step=row1.astype(np.float32), tok_prompt=tok, tok_step=int(row1.argmax()))
    print("wrote", out)


if __name__ == "__main__":
    main()
