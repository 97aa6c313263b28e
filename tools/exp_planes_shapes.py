#!/usr/bin/env python3
"""Planes-kernel duration for expert-batch shapes (run under rocprofv3 --kernel-trace --stats): N=14336, T=128, K and tile shape varied."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_ops import rand_weights
from oracle_py import Q5_K
rng = np.random.default_rng(1)
N = 14336
for K, T, tiles in [(4096, 128, 1), (4096, 128, 2), (2048, 128, 2), (2048, 128, 1), (4096, 512, 2), (1024, 128, 2)]:
    W = rand_weights(rng, Q5_K, N * K)
    x = rng.standard_normal((T, K)).astype(np.float32)
    be.set_option("mmq_planes", 1); be.set_option("mmq_tiles", tiles); be.set_option("mmq_ksplit", 0)
    for _ in range(3):
        be.mul_mat(Q5_K, W, N, K, x)
    print("done", K, T, tiles, flush=True)
