#!/usr/bin/env python3
"""Generates tests/golden/*.npz — committed input/output vectors for the hot path.

Provenance: the reference (`/root/reference`) cannot produce vectors for this path: its arithmetic lives in an
un-vendored submodule and its own tests hold none.  The vectors are therefore computed by
tests/np_twin.py, the numpy restatement written from the block-format specification that shares no code with oracle/*.c
or the HIP kernels: activation blocks (Q8_K / Q8_0), integer partial sums per (row, super-block) and the float64 dot
products built from them.  The C oracle (CPU test) and the HIP path (GPU test) are both checked against these files.

usage: python tests/golden/make_golden.py          (deterministic: fixed seeds; rewrites the .npz files)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import np_twin as tw  # noqa: E402

Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = tw.Q8_0, tw.Q4_K, tw.Q5_K, tw.Q6_K, tw.Q8_K
BB = {Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210}
NAMES = {Q8_0: "q8_0", Q4_K: "q4_k", Q5_K: "q5_k", Q6_K: "q6_k"}


def rand_weights(rng, t, n_elems, dscale=1e-2):
    be_ = 32 if t == Q8_0 else 256
    raw = rng.integers(0, 256, n_elems // be_ * BB[t], dtype=np.uint8)
    blk = raw.view(tw.DT[t])
    blk["d"] = (rng.uniform(0.5, 1.5, blk.size) * dscale).astype("<f2")
    if t in (Q4_K, Q5_K):
        blk["dmin"] = (rng.uniform(0.5, 1.5, blk.size) * dscale).astype("<f2")
    return raw


def main():
    out = {}
    N, K, T = 6, 1024, 35                      # T >= 32 so the GPU test also goes through the MFMA contraction
    rng = np.random.default_rng(20250404)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.05, 8.0, (T, 1))).astype(np.float32)
    x[1, 256:512] = 0.0                        # an all-zero block
    x[2, :6] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5]   # rounding ties
    x[3, 10] = 5.0; x[3, 200] = -5.0           # equal magnitudes: the first one sets the scale
    out["x"] = x
    out["act_q8_k"] = np.stack([tw.quantize_q8_K(x[t]) for t in range(T)])
    out["act_q8_0"] = np.stack([tw.quantize_q8_0(x[t]) for t in range(T)])
    for t in (Q4_K, Q5_K, Q6_K, Q8_0):
        W = rand_weights(rng, t, N * K)
        rb = W.size // N
        act = out["act_q8_0"] if t == Q8_0 else out["act_q8_k"]
        nblk = K // (32 if t == Q8_0 else 256)
        isum = np.zeros((T, N, nblk), np.int32)
        msum = np.zeros((T, N, nblk), np.int32)
        y = np.zeros((T, N), np.float64)
        wb = W.view(tw.DT[t])
        for tt in range(T):
            for r in range(N):
                i_, m_ = tw.int_partials(t, W[r * rb:(r + 1) * rb], act[tt])
                isum[tt, r], msum[tt, r] = i_, m_
                blk = wb[r * nblk:(r + 1) * nblk]
                d = blk["d"].astype(np.float64)
                if t == Q8_0:
                    ad = act[tt].view(tw.DT[Q8_0])["d"].astype(np.float64)
                    y[tt, r] = float((d * ad * i_).sum())
                else:
                    ad = act[tt].view(tw.DT[Q8_K])["d"].astype(np.float64)
                    term = d * ad * i_
                    if t in (Q4_K, Q5_K):
                        term = term - blk["dmin"].astype(np.float64) * ad * m_
                    y[tt, r] = float(term.sum())
        out[f"w_{NAMES[t]}"] = W
        out[f"isum_{NAMES[t]}"] = isum
        out[f"msum_{NAMES[t]}"] = msum
        out[f"y_{NAMES[t]}"] = y
        out[f"deq_{NAMES[t]}"] = tw.dequantize(t, W)
    out["shape_N_K_T"] = np.array([N, K, T], np.int32)
    np.savez_compressed(os.path.join(HERE, "quant_dot_v1.npz"), **out)
    print("wrote quant_dot_v1.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
