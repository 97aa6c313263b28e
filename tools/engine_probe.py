#!/usr/bin/env python3
"""Reads the raw stamps of the layer engine's probe (MI355_ENGINE_PROBE=<layer> MI355_ENGINE_PROBE_FILE=<path>: [workgroup][wave][32] u64, 10 ns ticks) and
prints per-XCD / per-workgroup views: which workgroups finish a mat-vec late, and what their loaders did meanwhile."""
import sys

import numpy as np

NW, NS = 10, 48
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, NW, NS).astype(np.int64)
ncu = t.shape[0]
base = t[:, :, 0][t[:, :, 0] > 0].min()
rel = lambda a: (a - base) * 0.01


def wg(idx, loader=False, red=np.max):
    w = slice(0, 2) if loader else slice(2, NW)
    return red(rel(t[:, w, idx]), axis=1)


names = [("wo decoded", 1, False), ("x' gather start", 2, False), ("x' in LDS", 3, False), ("gu act ready", 23, False), ("gu decoded", 4, False), ("gu loader issued", 3, True),
         ("codes in LDS", 7, False), ("dn decoded", 8, False), ("dn loader issued", 4, True), ("x'' in LDS", 10, False), ("qkv decoded", 11, False)]
print("per-workgroup (max over its waves), then by XCD (workgroup % 8): median / max")
for n, i, ld in names:
    v = wg(i, ld)
    print(f"{n:18s} all: med {np.median(v):6.2f} max {v.max():6.2f} argmax wg {int(v.argmax()):3d} |", "  ".join(f"x{x}: {np.median(v[x::8]):5.2f}/{v[x::8].max():5.2f}" for x in range(8)))
gu = wg(4)
order = np.argsort(-gu)[:12]
print("\nslowest gate|up workgroups: wg, gu decoded (max wave), its waves' decoded times, act ready, loader issued, sum wait / decode per wave (us)")
for w in order:
    print(f"wg {w:3d} xcd {w % 8}: {gu[w]:6.2f} | waves", " ".join(f"{x:5.2f}" for x in rel(t[w, 2:, 4])), "| ready", f"{rel(t[w, 2:, 23]).max():5.2f}", "| loader", " ".join(f"{x:5.2f}" for x in rel(t[w, :2, 3])),
          "| wait", " ".join(f"{x * 0.01:4.2f}" for x in t[w, 2:, 24]), "| dec", " ".join(f"{x * 0.01:4.2f}" for x in t[w, 2:, 25]))
fast = np.argsort(gu)[:4]
print("fastest:")
for w in fast:
    print(f"wg {w:3d} xcd {w % 8}: {gu[w]:6.2f} | waves", " ".join(f"{x:5.2f}" for x in rel(t[w, 2:, 4])), "| ready", f"{rel(t[w, 2:, 23]).max():5.2f}", "| loader", " ".join(f"{x:5.2f}" for x in rel(t[w, :2, 3])),
          "| wait", " ".join(f"{x * 0.01:4.2f}" for x in t[w, 2:, 24]), "| dec", " ".join(f"{x * 0.01:4.2f}" for x in t[w, 2:, 25]))
