import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import __graft_entry__ as ge
import oracle_py as oq
from oracle_py import F16, Q8_0
pkg = ge.load_pkg(); be = pkg.Backend()
D = 128
for (H, G, n_cells, T) in [(8, 2, 70, 40), (32, 8, 300, 33), (4, 4, 129, 64), (16, 2, 1000, 37)]:
    rng = np.random.default_rng(H * 1000 + n_cells + T)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = (rng.standard_normal((n_cells, G * D)) * rng.uniform(0.2, 3.0, (n_cells, 1))).astype(np.float32)
    kc = np.stack([oq.quantize(F16, r) for r in kf]); vc = np.stack([oq.quantize(F16, r) for r in vf])
    cell_pos = np.arange(n_cells, dtype=np.int32); cell_pos[rng.random(n_cells) < 0.1] = -1; cell_pos[0] = 0
    q_pos = np.sort(rng.integers(0, n_cells, T)).astype(np.int32); q_pos[0] = 0; q_pos[-1] = n_cells - 1
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    scale = 1 / np.sqrt(D)
    out = be.flash_attn(q, H, G, D, F16, kc, F16, vc, cell_pos, q_pos, scale)
    for acc in (0, 1):
        oq.set_fa_v_acc_f32(acc)
        errs = []
        for i in range(0, T, 3):
            cells = np.nonzero((cell_pos >= 0) & (cell_pos <= q_pos[i]))[0].astype(np.int32)
            ref = oq.flash_attn(q[i], H, G, D, F16, kc, F16, vc, cells, scale)
            errs.append(float(np.abs(out[i] - ref).max() / max(1.0, np.abs(ref).max())))
        print((H, G, n_cells, T), "v_acc_f32" if acc else "v_acc_f16", "max err", max(errs))
    oq.set_fa_v_acc_f32(0)
