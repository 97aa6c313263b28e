#!/usr/bin/env python3
"""Planes-kernel duration for chosen shapes (run under rocprofv3 --kernel-trace): (N, K, T, tile form) list below."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_ops import rand_weights
import oracle_py as oq
from oracle_py import Q4_K
rng = np.random.default_rng(1)
for N, K, T, tiles in [(14336, 4096, 512, 2), (14336, 4096, 512, 4), (4096, 4096, 512, 1), (4096, 4096, 512, 4), (4096, 14336, 512, 1), (4096, 14336, 512, 4),
                       (6144, 4096, 512, 2), (6144, 4096, 512, 4), (14336, 4096, 2048, 2), (14336, 4096, 2048, 4)]:
    W = rand_weights(rng, Q4_K, N * K)
    x = rng.standard_normal((T, K)).astype(np.float32)
    be.set_option("mmq_planes", 1); be.set_option("mmq_tiles", tiles); be.set_option("mmq_ksplit", 0)
    for _ in range(3):
        y = be.mul_mat(Q4_K, W, N, K, x)
    if N * T <= 4096 * 512 and K <= 4096:
        ref = oq.mul_mat(Q4_K, W, N, K, x)
        print("err", float(np.abs(y - ref).max() / np.abs(ref).max()))
    print("done", N, K, T, tiles, flush=True)
