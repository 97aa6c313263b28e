#!/usr/bin/env python3
"""Per-lane planes kernels vs the K-split kernel at expert-batch sizes (the Mixtral shapes, Q5_K): kernel durations per (shape, tokens).
Run under `rocprofv3 --kernel-trace --output-format csv`; `--parse trace.csv` prints the table."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(14336, 4096), (4096, 14336)]
TOKENS = (48, 64, 96, 128, 160, 192, 256)
MODES = ("ksplit", "planes 256x32", "planes 128x128")
REPS = 3
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv
    durs = []
    for row in csv.DictReader(open(sys.argv[2])):
        n = row["Kernel_Name"]
        if ("mmq_planes" in n or "mmq_ksplit" in n) and "expand" not in n:
            durs.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3, "ksplit" in n))
    durs.sort()
    i = 0
    print(f"{'N':>6} {'K':>6} {'T':>5} | " + " ".join(f"{m:>15s} us |" for m in MODES))
    for N, K in SHAPES:
        for T in TOKENS:
            cells = []
            for m in MODES:
                d = min(x[1] for x in durs[i:i + REPS]); assert all(x[2] == (m == "ksplit") for x in durs[i:i + REPS]), (N, K, T, m); i += REPS
                cells.append(f"{d:18.1f} |")
            print(f"{N:6d} {K:6d} {T:5d} | " + " ".join(cells))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MI355_KSPLIT_MAX"] = "2048"
import __graft_entry__ as ge
pkg = ge.load_pkg(); be = pkg.Backend()
from test_gpu_ops import rand_weights
from oracle_py import Q5_K
rng = np.random.default_rng(1)
be.set_option("mmq_planes", 1); be.set_option("mmq_lds_form", 0)
for N, K in SHAPES:
    W = rand_weights(rng, Q5_K, N * K)
    for T in TOKENS:
        x = rng.standard_normal((T, K)).astype(np.float32)
        for m in MODES:
            be.set_option("mmq_ksplit", 1 if m == "ksplit" else 0)
            be.set_option("mmq_tiles", 2 if m.endswith("128x128") else 1)
            for _ in range(REPS):
                be.mul_mat(Q5_K, W, N, K, x)
        print("done", N, K, T, flush=True)
be.set_option("mmq_lds_form", -1); be.set_option("mmq_tiles", 0); be.set_option("mmq_ksplit", 1)
