#!/usr/bin/env python3
"""Prefill contraction, per-lane planes kernel (128 x 128 and 256 x 32 tiles) vs both-operands-through-LDS kernel (128 x 256):
kernel durations per shape.  Run under `rocprofv3 --kernel-trace --output-format csv`; `--parse trace.csv` prints the table."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(28672, 4096), (6144, 4096), (4096, 4096), (4096, 14336), (14336, 4096)]
TOKENS = (256, 512, 1024, 2048)
TILES = (1, 2, 4)
REPS = 3
if os.environ.get("P2_QUICK"):
    SHAPES, TOKENS, TILES = [(28672, 4096), (4096, 4096)], (512, 2048), (4,)
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    durs = []
    for row in csv.DictReader(open(sys.argv[2])):
        if "mmq_planes" in row["Kernel_Name"] and "expand" not in row["Kernel_Name"]:
            durs.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3, "planes2" in row["Kernel_Name"]))
    durs.sort()
    i = 0
    print(f"{'N':>6} {'K':>6} {'T':>5} | " + " ".join(f"tiles={t:<2d} us  TOP/s |" for t in TILES))
    for N, K in SHAPES:
        for T in TOKENS:
            cells = []
            for t in TILES:
                d = min(x[1] for x in durs[i:i + REPS]); assert all(x[2] == (t == 4) for x in durs[i:i + REPS]); i += REPS
                cells.append(f"{d:9.1f} {2.0 * N * K * T / d / 1e6:7.0f} |")
            print(f"{N:6d} {K:6d} {T:5d} | " + " ".join(cells))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_ops import rand_weights
from oracle_py import Q4_K
rng = np.random.default_rng(1)
be.set_option("mmq_planes", 1); be.set_option("mmq_ksplit", 0)
for N, K in SHAPES:
    W = rand_weights(rng, Q4_K, N * K)
    for T in TOKENS:
        x = rng.standard_normal((T, K)).astype(np.float32)
        for t in TILES:
            be.set_option("mmq_tiles", t)
            for _ in range(REPS):
                be.mul_mat(Q4_K, W, N, K, x)
        print("done", N, K, T, flush=True)
