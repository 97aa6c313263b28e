"""Committed golden vectors (tests/golden/quant_dot_v1.npz, made by tests/golden/make_golden.py from the independent numpy
twin): the C oracle must reproduce them on the CPU, the HIP path on the GPU."""
import os

import numpy as np
import pytest

import oracle_py as oq

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "quant_dot_v1.npz"))
N, K, T = [int(v) for v in G["shape_N_K_T"]]
TYPES = {"q4_k": oq.Q4_K, "q5_k": oq.Q5_K, "q6_k": oq.Q6_K, "q8_0": oq.Q8_0}


def test_golden_is_what_the_generator_writes(tmp_path):
    """The fixture is reproducible from the committed script (fixed seeds)."""
    import subprocess
    import sys
    import shutil
    d = tmp_path / "golden"
    d.mkdir()
    shutil.copy(os.path.join(HERE, "golden", "make_golden.py"), d / "make_golden.py")
    shutil.copy(os.path.join(HERE, "np_twin.py"), tmp_path / "np_twin.py")
    subprocess.check_call([sys.executable, str(d / "make_golden.py")], stdout=subprocess.DEVNULL)
    H = np.load(d / "quant_dot_v1.npz")
    assert sorted(H.files) == sorted(G.files)
    for k in G.files:
        assert np.array_equal(G[k], H[k]), k


@pytest.mark.parametrize("act,key", [(oq.Q8_K, "act_q8_k"), (oq.Q8_0, "act_q8_0")])
def test_oracle_activation_blocks_match_golden(act, key):
    for t in range(T):
        assert np.array_equal(oq.quantize(act, G["x"][t]), G[key][t]), t


@pytest.mark.parametrize("name", list(TYPES))
def test_oracle_dequant_partials_and_dot_match_golden(name):
    t = TYPES[name]
    W = G[f"w_{name}"]
    rb = W.size // N
    assert np.array_equal(oq.dequantize(t, W, N * K), G[f"deq_{name}"])
    act = G["act_q8_0"] if t == oq.Q8_0 else G["act_q8_k"]
    for tt in range(0, T, 3):
        for r in range(N):
            wi, wm = oq.vec_dot_int_partials(t, W[r * rb:(r + 1) * rb], act[tt], K)
            assert np.array_equal(wi, G[f"isum_{name}"][tt, r]) and np.array_equal(wm, G[f"msum_{name}"][tt, r]), (tt, r)
    y = oq.mul_mat(t, W, N, K, G["x"])
    ref = G[f"y_{name}"]                                   # float64 sum of the exact per-block terms
    assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(TYPES))
def test_hip_matches_golden(pkg, name):
    be = pkg.Backend()
    t = TYPES[name]
    W = G[f"w_{name}"]
    x = G["x"]
    act_t = oq.Q8_0 if t == oq.Q8_0 else oq.Q8_K
    got_act = be.quantize_act(act_t, x)
    assert np.array_equal(got_act, G["act_q8_0" if t == oq.Q8_0 else "act_q8_k"])
    y, isum, msum = be.mul_mat(t, W, N, K, x, want_ints=True)        # T = 35: MFMA contraction for the K-quants
    assert np.array_equal(isum, G[f"isum_{name}"]) and np.array_equal(msum, G[f"msum_{name}"])
    ref = G[f"y_{name}"]
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    y1 = be.mul_mat(t, W, N, K, x[:1])                               # single token: the mat-vec kernels
    assert np.abs(y1 - ref[:1]).max() <= 2e-5 * np.abs(ref).max() + 1e-6
