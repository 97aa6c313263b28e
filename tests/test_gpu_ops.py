"""GPU parity, op by op, through the C-ABI per-op entry points: HIP kernels vs the CPU oracle on the same
seeded inputs.  Integer / byte / index results are compared bit-exactly; floating-point results within the
tolerance written next to each assert."""
import os
import numpy as np
import pytest

import np_twin as tw
import oracle_py as oq
from oracle_py import F16, F32, IQ4_NL, Q2_K, Q3_K, Q4_0, Q4_K, Q5_0, Q5_K, Q6_K, Q8_0, Q8_K

pytestmark = pytest.mark.gpu

BB = {Q4_0: 18, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q2_K: 84, Q3_K: 110, Q5_0: 22, IQ4_NL: 18}


@pytest.fixture(scope="module")
def be(pkg):
    return pkg.Backend()


def rand_weights(rng, t, n_elems, dscale=1e-2):
    be_ = 32 if t in (Q8_0, Q4_0, Q5_0, IQ4_NL) else 256
    raw = rng.integers(0, 256, n_elems // be_ * BB[t], dtype=np.uint8)
    blk = raw.view(tw.DT[t])
    blk["d"] = (rng.uniform(0.5, 1.5, blk.size) * dscale).astype("<f2")
    if t in (Q4_K, Q5_K, Q2_K):
        blk["dmin"] = (rng.uniform(0.5, 1.5, blk.size) * dscale).astype("<f2")
    return raw


@pytest.mark.parametrize("act", [Q8_K, Q8_0])
def test_activation_quant_bit_exact(be, act):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((5, 4096)) * rng.uniform(0.01, 30, (5, 1))).astype(np.float32)
    x[1, 256:512] = 0.0                         # all-zero block
    x[2, :6] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5]  # rounding ties
    x[2, 7] = -127.0
    x[3, 10] = 5.0; x[3, 200] = -5.0            # equal magnitudes, first wins
    got = be.quantize_act(act, x)
    for r in range(x.shape[0]):
        want = oq.quantize(act, x[r])
        assert (got[r] == want).all(), r


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K, Q8_0])
@pytest.mark.parametrize("K,N,T", [(256, 3, 1), (2048, 37, 1), (4096, 64, 2), (5632, 10, 3), (11008, 5, 5), (14336, 8, 1),
                                   (4096, 33, 21), (14336, 6, 9), (5632, 7, 16), (8192, 4, 8),
                                   # single token, K ending inside a pass of 2048 (persistent kernel's partial last pass)
                                   (5632, 70, 1), (11008, 33, 1), (5120, 20, 1), (13824, 9, 1), (768, 11, 1), (28672, 3, 1),
                                   (2048, 130, 40), (4096, 64, 129), (5632, 37, 33)])   # T >= 32: MFMA path for K-quants
def test_mul_mat_int_partials_exact_and_value(be, t, K, N, T):
    rng = np.random.default_rng(K + N + t)
    W = rand_weights(rng, t, N * K)
    x = rng.standard_normal((T, K)).astype(np.float32)
    y, isum, msum = be.mul_mat(t, W, N, K, x, want_ints=True)
    ref = oq.mul_mat(t, W, N, K, x)
    at = oq.vec_dot_type(t)
    rb = oq.row_bytes(t, K)
    for tt in range(T):
        act = oq.quantize(at, x[tt])
        for r in range(0, N, max(1, N // 7)):
            wi, wm = oq.vec_dot_int_partials(t, W[r * rb:(r + 1) * rb], act, K)
            assert (isum[tt, r] == wi).all() and (msum[tt, r] == wm).all(), (tt, r)
    # f32 summation order differs (wave butterfly vs 8-lane scalar): tolerance 2e-5 of the output scale
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("t", [Q2_K, Q3_K])
@pytest.mark.parametrize("K,N,T", [(256, 3, 1), (2048, 37, 1), (4096, 64, 2), (5632, 10, 3), (2048, 33, 21), (5632, 7, 16), (4096, 20, 8), (5632, 70, 1), (768, 11, 1),
                                   (2048, 50, 40)])
def test_mul_mat_q2_k_q3_k_int_partials_exact_and_value(be, t, K, N, T):
    """Q2_K / Q3_K tensors (the mixes of the Q2_K and Q3_K_S/M/L files the reference publishes, .github/workflows/convert-model-all-quant.yml:106-152; its
    e2e smoke model is a TinyLlama Q2_K file, Makefile:5): integer sums per (token, row, super-block) bit-exact against ggml_vec_dot_q2_K_q8_K /
    _q3_K_q8_K as restated in oracle/, f32 result within the re-association bound; single tokens, 2..21 tokens (tiled mat-vec) and a 40-token batch."""
    rng = np.random.default_rng(K + N + t)
    W = rand_weights(rng, t, N * K)
    x = rng.standard_normal((T, K)).astype(np.float32)
    y, isum, msum = be.mul_mat(t, W, N, K, x, want_ints=True)
    ref = oq.mul_mat(t, W, N, K, x)
    rb = oq.row_bytes(t, K)
    for tt in range(T):
        act = oq.quantize(Q8_K, x[tt])
        for r in range(0, N, max(1, N // 7)):
            wi, wm = oq.vec_dot_int_partials(t, W[r * rb:(r + 1) * rb], act, K)
            assert (isum[tt, r] == wi).all() and (msum[tt, r] == wm).all(), (tt, r)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("t", [Q4_0, Q5_0, IQ4_NL])
@pytest.mark.parametrize("K,N,T", [(256, 3, 1), (2048, 37, 1), (4096, 64, 2), (5632, 10, 3), (2048, 33, 21), (5632, 7, 16), (4096, 20, 8), (11008, 9, 1), (768, 11, 1),
                                   (2048, 50, 40), (4096, 300, 129), (14336, 70, 512), (5632, 129, 33)])
def test_mul_mat_32_element_formats_int_partials_exact_and_value(be, t, K, N, T):
    """Q4_0 / Q5_0 / IQ4_NL tensors (llama-quantize's legacy types and the fallbacks it takes for rows that are not a multiple of 256):
    integer sums per (token, row, 32-block) bit-exact against ggml_vec_dot_q4_0_q8_0 / _q5_0_q8_0 / _iq4_nl_q8_0 as restated in oracle/ (activation:
    Q8_0 blocks), f32 result within the re-association bound; single tokens, 2..21 tokens (tiled mat-vec) and batches of 32 tokens and more (the Q8_0
    matrix-core kernel on an exact Q8_0-layout copy of the tensor)."""
    rng = np.random.default_rng(K + N + t)
    W = rand_weights(rng, t, N * K)
    x = rng.standard_normal((T, K)).astype(np.float32)
    y, isum, msum = be.mul_mat(t, W, N, K, x, want_ints=True)
    ref = oq.mul_mat(t, W, N, K, x)
    rb = oq.row_bytes(t, K)
    for tt in range(T):
        act = oq.quantize(Q8_0, x[tt])
        for r in range(0, N, max(1, N // 7)):
            wi, wm = oq.vec_dot_int_partials(t, W[r * rb:(r + 1) * rb], act, K)
            assert (isum[tt, r] == wi).all() and (msum[tt, r] == 0).all(), (tt, r)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("planes", [0, 1])
@pytest.mark.parametrize("tiles", [1, 2])
@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K])
@pytest.mark.parametrize("K,N,T", [(2048, 130, 40), (4096, 300, 129), (5632, 37, 33), (4096, 70, 512), (14336, 40, 96)])
def test_mul_mat_mfma_prefill_paths(be, t, K, N, T, planes, tiles):
    """Prompt-processing contraction on the matrix cores, both forms (weights expanded into int8 operand planes ahead of
    time / on the fly) and both wave tilings; ragged row and token tails.  The integer sums are exact by construction
    (scale folded into the weight, two int8 planes), so only the f32 order over super-blocks differs from the oracle."""
    rng = np.random.default_rng(K + N + T + t)
    W = rand_weights(rng, t, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 4.0, (T, 1))).astype(np.float32)
    be.set_option("mmq_planes", planes)
    be.set_option("mmq_tiles", tiles)
    be.set_option("mmq_ksplit", 0)               # T in 33..64 would otherwise take the small-batch kernel (tested below)
    try:
        y = be.mul_mat(t, W, N, K, x)
    finally:
        be.set_option("mmq_planes", 1)
        be.set_option("mmq_tiles", 0)
        be.set_option("mmq_ksplit", 1)
    ref = oq.mul_mat(t, W, N, K, x)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K, Q2_K, Q3_K])
@pytest.mark.parametrize("K,N,T", [(2048, 130, 40), (4096, 300, 129), (1024, 128, 256), (4096, 70, 512), (14336, 40, 300), (256, 257, 513)])
def test_mul_mat_prefill_both_operands_through_lds(be, t, K, N, T):
    """The 128-row x 256-token workgroup tile whose weight planes AND activation codes reach LDS by DMA (mmq_planes2_kernel):
    same integer sums, same fold and same f32 order over super-blocks as the per-lane planes kernel, so the two agree bit
    for bit; ragged row tiles, ragged token tiles and a single super-block included.  Q2_K / Q3_K tensors ride on the same kernels through
    plane sets in the Q4_K / Q6_K formats (Q2_K: sixteen 4-bit mins per super-block, one per block sum)."""
    rng = np.random.default_rng(7 * K + N + T + t)
    W = rand_weights(rng, t, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 4.0, (T, 1))).astype(np.float32)
    be.set_option("mmq_planes", 1)
    be.set_option("mmq_ksplit", 0)
    try:
        be.set_option("mmq_tiles", 4)
        be.set_option("mmq_split", 1)                          # no K split (tested below): same f32 order as the other kernel
        y = be.mul_mat(t, W, N, K, x)
        be.set_option("mmq_tiles", 2)
        y2 = be.mul_mat(t, W, N, K, x)
    finally:
        be.set_option("mmq_tiles", 0)
        be.set_option("mmq_split", 0)
        be.set_option("mmq_ksplit", 1)
    ref = oq.mul_mat(t, W, N, K, x)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K])
@pytest.mark.parametrize("K,N,T", [(1024, 64, 40), (4096, 96, 300), (2048, 224, 513), (256, 32, 257)])
def test_ffn_gate_up_swiglu_launch(be, t, K, N, T):
    """ffn_gate and ffn_up in one launch of the LDS kernel, 64 rows of each per workgroup, the up tile handed to the gate waves through
    LDS and SwiGLU in the epilogue: against silu(gate . x) * (up . x) of the oracle's mat-muls; ragged token tiles, a last workgroup
    with 32 rows of each tensor, a single super-block."""
    rng = np.random.default_rng(3 * K + N + T + t)
    Wg, Wu = rand_weights(rng, t, N * K), rand_weights(rng, t, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 2.0, (T, 1))).astype(np.float32)
    be.set_option("mmq_tiles", 4)
    try:
        y = be.ffn_gate_up(t, Wg, Wu, N, K, x)
    finally:
        be.set_option("mmq_tiles", 0)
    g, u = oq.mul_mat(t, Wg, N, K, x), oq.mul_mat(t, Wu, N, K, x)
    with np.errstate(over="ignore"):                           # exp(-g) may overflow to inf: g / inf = 0, as on the device
        ref = (g / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32) * u
    assert np.abs(y - ref).max() <= 4e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("t", [Q4_K, Q6_K])
@pytest.mark.parametrize("K,N,T,split", [(4096, 300, 129, 2), (2048, 128, 256, 3), (14336, 40, 300, 4), (1024, 260, 513, 4), (256, 36, 140, 4)])
def test_mul_mat_prefill_split_k(be, t, K, N, T, split):
    """Tensors with few rows: the 128 x 256 kernel splits K over `split` workgroups, partial sums go through a workspace and are
    added in split order by mmq_splitk_reduce_kernel.  Only the f32 order over super-blocks changes; a split wider than the
    number of super-blocks is clamped."""
    rng = np.random.default_rng(11 * K + N + T + t)
    W = rand_weights(rng, t, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 4.0, (T, 1))).astype(np.float32)
    be.set_option("mmq_planes", 1)
    be.set_option("mmq_ksplit", 0)
    try:
        be.set_option("mmq_tiles", 4)
        be.set_option("mmq_split", split)
        y = be.mul_mat(t, W, N, K, x)
        be.set_option("mmq_split", 1)
        y1 = be.mul_mat(t, W, N, K, x)
    finally:
        be.set_option("mmq_tiles", 0)
        be.set_option("mmq_split", 0)
        be.set_option("mmq_ksplit", 1)
    ref = oq.mul_mat(t, W, N, K, x)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6
    assert np.abs(y - y1).max() <= 2e-6 * np.abs(ref).max() + 1e-6
    if K > 256:
        assert not np.array_equal(y, y1)                       # the split really ran (a different f32 order)


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K])
@pytest.mark.parametrize("K,N,T", [(4096, 100, 8), (4096, 64, 32), (14336, 40, 17), (5632, 37, 33), (2048, 130, 64), (256, 5, 9),
                                   (4096, 33, 48), (4096, 70, 5), (14336, 33, 6), (4096, 64, 3), (2048, 40, 4), (4096, 70, 129), (2048, 40, 250), (14336, 33, 97)])
def test_mul_mat_small_batch_ksplit(be, t, K, N, T):
    """Continuous-batching decode steps and short prompts (3..256 tokens): MFMA contraction with K split over the waves of a workgroup;
    ragged rows / tokens, K smaller than the 8-way split."""
    rng = np.random.default_rng(K + N + T + t)
    W = rand_weights(rng, t, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 4.0, (T, 1))).astype(np.float32)
    y = be.mul_mat(t, W, N, K, x)
    ref = oq.mul_mat(t, W, N, K, x)
    assert np.abs(y - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("K,N,T", [(2048, 130, 40), (4096, 300, 129), (5632, 37, 33), (1024, 256, 512), (14336, 40, 96), (256, 5, 32)])
def test_mul_mat_q8_0_prefill_mfma_bit_exact(be, K, N, T):
    """Prompt batches against Q8_0 weights (mmq_q80.hip): one int8 MFMA per 32-block, one f32 multiply-add per block in block
    order — the arithmetic of ggml_vec_dot_q8_0_q8_0, so the result equals the CPU restatement bit for bit; ragged rows and
    tokens, K beyond 48 KB of scale staging."""
    rng = np.random.default_rng(K + N + T)
    W = rand_weights(rng, Q8_0, N * K)
    x = (rng.standard_normal((T, K)) * rng.uniform(0.1, 4.0, (T, 1))).astype(np.float32)
    y = be.mul_mat(Q8_0, W, N, K, x)
    ref = oq.mul_mat(Q8_0, W, N, K, x)
    assert np.array_equal(y, ref), float(np.abs(y - ref).max())


def test_mul_mat_f32_f16_weights(be):
    rng = np.random.default_rng(5)
    N, K, T = 8, 512, 3
    x = rng.standard_normal((T, K)).astype(np.float32)
    Wf = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    y = be.mul_mat(F32, Wf.view(np.uint8), N, K, x)
    assert np.abs(y - oq.mul_mat(F32, Wf.view(np.uint8), N, K, x)).max() <= 1e-5
    Wh = Wf.astype(np.float16)
    y = be.mul_mat(F16, Wh.view(np.uint8), N, K, x)
    assert np.abs(y - oq.mul_mat(F16, Wh.view(np.uint8), N, K, x)).max() <= 1e-5


@pytest.mark.parametrize("N,K,T", [(768, 768, 128), (2304, 768, 70), (3072, 768, 33), (768, 3072, 200), (100, 512, 9), (36, 256, 8), (4096, 4096, 129)])
def test_mul_mat_f16_weights_batched_on_the_matrix_cores(be, N, K, T):
    """ggml_mul_mat with f16 weights for a batch (mmf.hip): f16-rounded activations, exact products, f32 accumulation - against the CPU restatement's vec_dot_f16
    path; ragged sizes exercise the partial tiles.  Sums of K products of magnitude ~0.05 x 1: agreement to f32 re-association."""
    rng = np.random.default_rng(N + K + T)
    x = rng.standard_normal((T, K)).astype(np.float32)
    Wh = (rng.standard_normal((N, K)) * 0.05).astype(np.float16)
    y = be.mul_mat(F16, Wh.view(np.uint8), N, K, x)
    ref = oq.mul_mat(F16, Wh.view(np.uint8), N, K, x)
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), float(np.abs(y - ref).max())
    # and against float64 on the same rounded operands
    want = x.astype(np.float16).astype(np.float64) @ Wh.astype(np.float64).T
    assert np.abs(y - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max()))


def test_rms_norm_mul(be):
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((3, 4096)) * 4).astype(np.float32)
    w = rng.uniform(0.5, 1.5, 4096).astype(np.float32)
    y = be.rms_norm_mul(x, w, 1e-5)
    for r in range(3):
        ref = oq.rms_norm(x[r], 1e-5) * w
        # f64 tree vs sequential sum of squares: <= 1 ulp of the f32 scale factor
        assert np.abs(y[r] - ref).max() <= 2e-7 * np.abs(ref).max()


@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("base", [1e4, 5e5])
def test_rope(be, neox, base):
    rng = np.random.default_rng(3)
    H, D = 8, 128
    pos = np.array([0, 1, 17, 511, 4095], np.int32)
    x = rng.standard_normal((pos.size, H, D)).astype(np.float32)
    y = be.rope(x, H, D, pos, base, neox=neox)
    for i, p in enumerate(pos):
        ref = oq.rope(x[i], H, D, int(p), base, neox=neox)
        # identical f32 angle recurrence; device cosf/sinf vs libm differ by a few ulp
        assert np.abs(y[i] - ref).max() <= 4e-6, (p, np.abs(y[i] - ref).max())
    assert (y[0] == x[0]).all()


@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("D,n_orig,fs", [(128, 256, 0.25), (64, 32, 0.125), (128, 8192, 0.5)])
def test_rope_yarn(be, neox, D, n_orig, fs):
    """rope.scaling.type "yarn": the device angle / magnitude (attn.hip rope_angle) against the CPU restatement of rope_yarn"""
    rng = np.random.default_rng(8)
    H, base, attn = 4, 1e4, 0.8
    lo, hi = oq.yarn_corr_dims(D, n_orig, base)
    pos = np.array([0, 1, 29, 1023, 6000], np.int32)
    x = rng.standard_normal((pos.size, H, D)).astype(np.float32)
    y = be.rope_yarn(x, H, D, pos, base, fs, 1.0, attn, lo, hi, neox=neox)
    for i, p in enumerate(pos):
        ref = oq.rope_yarn(x[i], H, D, int(p), base, fs, 1.0, attn, lo, hi, neox=neox)
        assert np.abs(y[i] - ref).max() <= 4e-6, (p, np.abs(y[i] - ref).max())
    # ext_factor 0 / attn_factor 1 is the plain entry point, bit for bit
    assert (be.rope_yarn(x, H, D, pos, base, fs, 0.0, 1.0, lo, hi, neox=neox) == be.rope(x, H, D, pos, base, neox=neox, freq_scale=fs)).all()


@pytest.mark.parametrize("t", [Q4_K, Q5_K, Q6_K, Q8_0, F16, F32, Q2_K, Q3_K, Q4_0, Q5_0, IQ4_NL])
def test_get_rows_bit_exact(be, t):
    rng = np.random.default_rng(4 + t)
    K, R = 1024, 40
    if t == F32:
        table = rng.standard_normal(R * K).astype(np.float32).view(np.uint8)
    elif t == F16:
        table = rng.standard_normal(R * K).astype(np.float16).view(np.uint8)
    else:
        table = rand_weights(rng, t, R * K, 1.0)
    ids = np.array([0, 39, 7, 7, 21], np.int32)
    got = be.get_rows(t, table, K, R, ids)
    rb = oq.row_bytes(t, K)
    for i, r in enumerate(ids):
        want = oq.dequantize(t, table[r * rb:(r + 1) * rb], K)
        assert got[i].view(np.uint32).tolist() == want.view(np.uint32).tolist(), (t, r)


def test_swiglu_softmax(be):
    rng = np.random.default_rng(6)
    g = (rng.standard_normal(14336) * 3).astype(np.float32)
    u = rng.standard_normal(14336).astype(np.float32)
    ref = oq.silu(g) * u
    assert np.abs(be.swiglu(g, u) - ref).max() <= 1e-6 * max(1, np.abs(ref).max())
    x = rng.standard_normal((4, 777)).astype(np.float32) * 5
    m = np.where(rng.random((4, 777)) < 0.3, -np.inf, 0).astype(np.float32)
    y = be.soft_max(x, m, 0.25)
    for r in range(4):
        assert np.abs(y[r] - oq.soft_max(x[r], m[r], 0.25)).max() <= 1e-7


@pytest.mark.parametrize("tk,tv,tol", [(F16, F16, 3e-3), (Q8_0, Q8_0, 2e-5), (Q4_0, Q4_0, 2e-5), (Q8_0, F16, 3e-3)])
@pytest.mark.parametrize("H,G,D,n_cells", [(8, 2, 128, 70), (32, 4, 64, 33), (4, 4, 128, 300), (32, 8, 128, 1500),
                                           (32, 8, 128, 4000)])    # BASELINE's ctx_len = 4096 with the context filled: 63 chunks per kv head, merged in two request rounds
def test_flash_attn(be, tk, tv, tol, H, G, D, n_cells):
    rng = np.random.default_rng(H * 1000 + n_cells)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    kc = np.stack([oq.quantize(tk, r) for r in kf])
    vc = np.stack([oq.quantize(tv, r) for r in vf])
    cell_pos = np.arange(n_cells, dtype=np.int32)
    cell_pos[rng.random(n_cells) < 0.1] = -1            # holes
    q_pos = np.array([n_cells - 1, n_cells // 2, 3], np.int32)
    q = rng.standard_normal((q_pos.size, H, D)).astype(np.float32)
    scale = 1 / np.sqrt(D)
    out = be.flash_attn(q, H, G, D, tk, kc, tv, vc, cell_pos, q_pos, scale)
    for i, qp in enumerate(q_pos):
        cells = np.nonzero((cell_pos >= 0) & (cell_pos <= qp))[0].astype(np.int32)
        ref = oq.flash_attn(q[i], H, G, D, tk, kc, tv, vc, cells, scale)
        # q8_0/q4_0: integer dots exact, f32 softmax order differs; f16 V: the CPU path accumulates V in fp16, the
        # HIP path in f32, so the f16 case is only as tight as that fp16 accumulation (3e-3)
        assert np.abs(out[i] - ref).max() <= tol, (i, np.abs(out[i] - ref).max())


@pytest.mark.parametrize("tkv,tol", [(Q8_0, 2e-5), (F16, 2e-5)])
@pytest.mark.parametrize("H,G,n_cells,T", [(8, 2, 70, 40), (32, 8, 300, 33), (4, 4, 129, 64), (16, 2, 1000, 37), (32, 8, 700, 200), (8, 2, 520, 97), (2, 2, 333, 130), (8, 4, 500, 150), (4, 4, 900, 300)])
def test_flash_attn_prefill_matrix_cores(be, H, G, n_cells, T, tkv, tol):
    """Prompt-processing attention on the MFMA path (T >= 32 query rows, head_dim 128, q8_0 or f16 K / V): ragged query
    tile, ragged last key chunk, holes in the cache, queries that see one cell only, R = 1 / 4 / 8 heads per kv head; several
    tiles whose key halves and query sub-tiles (R = 1: four, R = 2: two per workgroup) see different chunk ranges, a last tile with one
    sub-tile only, key splits across workgroups.
    f16: the CPU path accumulates V in fp16 and is itself only good to ~1e-2 over hundreds of cells, so the tight comparison
    is against the restatement with that accumulation in f32 (oq.set_fa_v_acc_f32) — the HIP kernel agrees with it to 1e-6 —
    and the stock fp16 mode is checked at its own noise level."""
    D = 128
    rng = np.random.default_rng(H * 1000 + n_cells + T)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = (rng.standard_normal((n_cells, G * D)) * rng.uniform(0.2, 3.0, (n_cells, 1))).astype(np.float32)
    kc = np.stack([oq.quantize(tkv, r) for r in kf])
    vc = np.stack([oq.quantize(tkv, r) for r in vf])
    cell_pos = np.arange(n_cells, dtype=np.int32)
    cell_pos[rng.random(n_cells) < 0.1] = -1            # holes
    cell_pos[0] = 0
    q_pos = np.sort(rng.integers(0, n_cells, T)).astype(np.int32)
    q_pos[0] = 0                                         # sees a single cell
    q_pos[-1] = n_cells - 1
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    scale = 1 / np.sqrt(D)
    out = be.flash_attn(q, H, G, D, tkv, kc, tkv, vc, cell_pos, q_pos, scale)
    oq.set_fa_v_acc_f32(1 if tkv == F16 else 0)
    try:
        for i in range(0, T, 3):
            cells = np.nonzero((cell_pos >= 0) & (cell_pos <= q_pos[i]))[0].astype(np.int32)
            ref = oq.flash_attn(q[i], H, G, D, tkv, kc, tkv, vc, cells, scale)
            assert np.abs(out[i] - ref).max() <= tol * max(1.0, float(np.abs(ref).max())), (i, np.abs(out[i] - ref).max())
    finally:
        oq.set_fa_v_acc_f32(0)
    if tkv == F16:
        for i in range(0, T, 7):
            cells = np.nonzero((cell_pos >= 0) & (cell_pos <= q_pos[i]))[0].astype(np.int32)
            ref = oq.flash_attn(q[i], H, G, D, tkv, kc, tkv, vc, cells, scale)
            assert np.abs(out[i] - ref).max() <= 2e-2 * max(1.0, float(np.abs(ref).max())), (i, np.abs(out[i] - ref).max())


@pytest.mark.parametrize("H,G,D,n_cells,T", [(8, 2, 128, 300, 5), (32, 8, 128, 1000, 3), (4, 4, 64, 700, 4), (32, 8, 128, 4000, 2), (8, 8, 128, 4500, 2), (6, 2, 128, 130, 40)])
def test_flash_attn_f16_cache_parity_mode(be, H, G, D, n_cells, T):
    """The opt-in kernel that reads an f16 cache with the CPU path's own arithmetic (cell by cell, double-precision scores, V accumulated in FP16 - option
    "fa_v_acc_f16") against the oracle in its STOCK mode.  The parallel kernels sit ~1e-2 from that mode (test_flash_attn: 3e-3 at 300 cells, the matrix-core
    prompt kernel 2e-2); this one is held to 2e-4 of the largest output, to a tenth of the parallel kernels' distance, and to 75 % of the compared elements
    bit-identical.  Measured: 0 .. 7.8e-5 against 2.1e-3 .. 4.3e-3, 81 - 100 % identical; what differs is an ulp of the final division by the softmax sum, or an
    element where one half-precision rounding fell the other way - the device's expf against the host's.  Holes, later positions, more than
    one 4096-cell stretch (4500 cells), 40 queries, head_dim 64."""
    rng = np.random.default_rng(H * 1000 + n_cells)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = (rng.standard_normal((n_cells, G * D)) * rng.uniform(0.2, 3.0, (n_cells, 1))).astype(np.float32)
    kc = np.stack([oq.quantize(F16, r) for r in kf])
    vc = np.stack([oq.quantize(F16, r) for r in vf])
    cell_pos = np.arange(n_cells, dtype=np.int32)
    cell_pos[rng.random(n_cells) < 0.1] = -1            # holes
    cell_pos[0] = 0
    q_pos = np.sort(rng.integers(0, n_cells, T)).astype(np.int32)
    q_pos[0] = 0                                         # sees a single cell
    q_pos[-1] = n_cells - 1
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    scale = 1 / np.sqrt(D)
    be.set_option("fa_v_acc_f16", 1)
    try:
        out = be.flash_attn(q, H, G, D, F16, kc, F16, vc, cell_pos, q_pos, scale)
    finally:
        be.set_option("fa_v_acc_f16", -1)
    plain = be.flash_attn(q, H, G, D, F16, kc, F16, vc, cell_pos, q_pos, scale)
    oq.set_fa_v_acc_f32(0)
    worst, worst_plain, same = 0.0, 0.0, 0
    rows = list(range(0, T, max(1, T // 6)))
    for i in rows:
        cells = np.nonzero((cell_pos >= 0) & (cell_pos <= q_pos[i]))[0].astype(np.int32)
        ref = oq.flash_attn(q[i], H, G, D, F16, kc, F16, vc, cells, scale)
        den = max(1.0, float(np.abs(ref).max()))
        worst = max(worst, float(np.abs(out[i] - ref).max()) / den)
        worst_plain = max(worst_plain, float(np.abs(plain[i] - ref).max()) / den)
        same += int((out[i].reshape(-1) == ref.reshape(-1)).sum())
    if os.environ.get("MI355_TEST_RECORD_FLIPS"):
        with open(os.environ["MI355_TEST_RECORD_FLIPS"], "a") as f:
            f.write(f"fa_v16 H={H} G={G} D={D} cells={n_cells} T={T}: parity {worst:.3g} plain {worst_plain:.3g} identical elements {same}/{len(rows) * H * D}\n")
    assert worst <= 2e-4 and worst * 10 <= worst_plain, (worst, worst_plain, same)
    assert same >= 0.75 * len(rows) * H * D, (worst, same, len(rows) * H * D)


@pytest.mark.parametrize("tkv,tol", [(Q8_0, 2e-5), (F16, 2e-5), (Q4_0, 2e-5)])
def test_flash_attn_prefill_second_micro_batch_of_a_filled_context(be, tkv, tol):
    """The prompt attention as the SECOND micro-batch of a 3968-token prompt runs it (BASELINE's ctx_len = 4096): 200 queries whose positions start
    at cell 2048, over a cache that holds 3968 cells - every tile walks 64 .. 124 key chunks, split across workgroups, and its causal frontier lies deep
    inside the cache."""
    H, G, D, n_cells, T = 32, 8, 128, 3968, 200
    rng = np.random.default_rng(4096)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = (rng.standard_normal((n_cells, G * D)) * rng.uniform(0.2, 3.0, (n_cells, 1))).astype(np.float32)
    kc = np.stack([oq.quantize(tkv, r) for r in kf])
    vc = np.stack([oq.quantize(tkv, r) for r in vf])
    cell_pos = np.arange(n_cells, dtype=np.int32)
    cell_pos[rng.random(n_cells) < 0.05] = -1            # holes
    cell_pos[0] = 0
    q_pos = np.sort(rng.integers(2048, n_cells, T)).astype(np.int32)
    q_pos[0] = 2048
    q_pos[-1] = n_cells - 1
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    scale = 1 / np.sqrt(D)
    out = be.flash_attn(q, H, G, D, tkv, kc, tkv, vc, cell_pos, q_pos, scale)
    oq.set_fa_v_acc_f32(1 if tkv == F16 else 0)
    try:
        for i in list(range(0, T, 9)) + [T - 1]:
            cells = np.nonzero((cell_pos >= 0) & (cell_pos <= q_pos[i]))[0].astype(np.int32)
            ref = oq.flash_attn(q[i], H, G, D, tkv, kc, tkv, vc, cells, scale)
            assert np.abs(out[i] - ref).max() <= tol * max(1.0, float(np.abs(ref).max())), (i, np.abs(out[i] - ref).max())
    finally:
        oq.set_fa_v_acc_f32(0)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("tkv", [Q8_0, F16])
@pytest.mark.parametrize("H,G,n_cells,t_o", [(32, 8, 4000, Q4_K),     # BASELINE config 3's geometry with the context filled (ctx_len 4096): 128-cell items, 32 per kv head
                                             (32, 8, 2048, Q4_K),     # the last length that still takes 64-cell items (32 x 8 = 256 = one per CU)
                                             (32, 8, 1500, Q6_K), (32, 8, 561, Q5_K),
                                             (8, 2, 70, Q5_K), (4, 4, 300, Q4_K),             # one query head per kv head: two kv heads per merge ticket
                                             (16, 2, 2100, Q4_K), (8, 4, 130, Q6_K)])        # 8 and 2 query heads per kv head
def test_attn_step_decode_block(be, fused, tkv, H, G, n_cells, t_o):
    """The attention block of a single-token step as the decode path launches it - rope of q and of the token's K row, the K / V row quantised into the cache,
    flash_attn_ext over the visible cells (holes, cells of later positions), the merge of the chunk partials, Q8_K quantisation, attn_output mat-vec + residual -
    against the oracle's ops chained the same way; fused = 1 is attn_out.hip (one launch), 0 the single-launch attention + the weight-stream mat-vec.
    Up to 4000 cells: the length BASELINE's ctx_len = 4096 decodes at."""
    D, base = 128, 500000.0
    E = K = H * D
    rng = np.random.default_rng(H * 1000 + n_cells + t_o)
    kf = rng.standard_normal((n_cells, G * D)).astype(np.float32)
    vf = (rng.standard_normal((n_cells, G * D)) * rng.uniform(0.2, 3.0, (n_cells, 1))).astype(np.float32)
    kc = np.stack([oq.quantize(tkv, r) for r in kf])
    vc = np.stack([oq.quantize(tkv, r) for r in vf])
    tok_pos = n_cells + 7
    cell_pos = rng.permutation(n_cells).astype(np.int32)           # positions scattered over the cells (a cache after shifts and reuse)
    cell_pos[rng.random(n_cells) < 0.1] = -1                        # holes
    cell_pos[rng.random(n_cells) < 0.03] = tok_pos + 5              # cells of later positions: not visible
    tok_cell = int(n_cells * 0.61)
    cell_pos[tok_cell] = tok_pos
    q = rng.standard_normal((H, D)).astype(np.float32)
    k_new = rng.standard_normal(G * D).astype(np.float32)
    v_new = (rng.standard_normal(G * D) * 1.7).astype(np.float32)
    W = rand_weights(rng, t_o, E * K)
    resid = rng.standard_normal(E).astype(np.float32)
    scale = 1 / np.sqrt(D)
    rb = oq.row_bytes(tkv, G * D)
    att, out, kr, vr = be.attn_step(q, k_new, v_new, H, G, D, tkv, kc, tkv, vc, cell_pos, tok_pos, tok_cell, base, D, scale, t_o, W, E, resid, fused, rb, rb)
    # the oracle, op by op
    qr = oq.rope(q, H, D, tok_pos, base)
    kn = oq.rope(k_new, G, D, tok_pos, base).reshape(-1)
    k_row = oq.quantize(tkv, kn)
    v_row = oq.quantize(tkv, v_new)
    assert (vr == v_row).all()                                      # no arithmetic before the V row's quantisation: bit-exact
    # the K row goes through the rope first (device cosf / sinf: 4e-6): its codes may sit one step off on a rounding tie
    dk_ref, dk_got = oq.dequantize(tkv, k_row, G * D), oq.dequantize(tkv, kr, G * D)
    step = np.abs(kn).reshape(-1, 32).max(axis=1).repeat(32) / 127 if tkv == Q8_0 else np.abs(kn) * 2.0 ** -10
    assert (np.abs(dk_ref - dk_got) <= 1.01 * step + 1e-7).all()
    assert (dk_ref == dk_got).mean() >= 0.99
    kc2, vc2 = kc.copy(), vc.copy()
    kc2[tok_cell] = kr                                              # (the row the device wrote: what its attention saw)
    vc2[tok_cell] = v_row
    cells = np.nonzero((cell_pos >= 0) & (cell_pos <= tok_pos))[0].astype(np.int32)
    oq.set_fa_v_acc_f32(1 if tkv == F16 else 0)                     # (f16 cache: the CPU accumulates V in fp16, the HIP path in f32 - see test_flash_attn)
    try:
        ref_att = oq.flash_attn(qr, H, G, D, tkv, kc2, tkv, vc2, cells, scale).reshape(-1)
    finally:
        oq.set_fa_v_acc_f32(0)
    assert np.abs(att - ref_att).max() <= 2e-5 * max(1.0, float(np.abs(ref_att).max())), float(np.abs(att - ref_att).max())
    # the mat-vec on the device's own attention output: the Q8_K codes are then the oracle's (bit-exact quantiser), only the f32 order differs
    ref_out = resid + oq.mul_mat(t_o, W, E, K, att.reshape(1, -1))[0]
    assert np.abs(out - ref_out).max() <= 2e-5 * max(1.0, float(np.abs(ref_out - resid).max())) + 1e-6, float(np.abs(out - ref_out).max())
    # ... and end to end against the oracle's own chain: a rounding flip of one activation code moves an output by ~1e-2 of its scale at most
    ref_e2e = resid + oq.mul_mat(t_o, W, E, K, ref_att.reshape(1, -1))[0]
    assert np.abs(out - ref_e2e).max() <= 3e-2 * max(1.0, float(np.abs(ref_e2e - resid).max()))
