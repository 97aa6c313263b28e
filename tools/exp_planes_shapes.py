#!/usr/bin/env python3
"""Planes kernel vs K-split kernel duration for chosen shapes (run under rocprofv3 --kernel-trace)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MI355_KSPLIT_MAX"] = "2048"
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_ops import rand_weights
from oracle_py import Q4_K
rng = np.random.default_rng(1)
for N, K in [(4096, 4096), (4096, 14336), (14336, 4096)]:
    W = rand_weights(rng, Q4_K, N * K)
    for T in (384, 512, 768):
        x = rng.standard_normal((T, K)).astype(np.float32)
        for ks in (0, 1):
            be.set_option("mmq_planes", 1); be.set_option("mmq_tiles", 0); be.set_option("mmq_ksplit", ks)
            for _ in range(3):
                be.mul_mat(Q4_K, W, N, K, x)
            print("done", N, K, T, "ksplit" if ks else "planes", flush=True)
