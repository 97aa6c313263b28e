"""Known-answer tests that pin the CPU oracle (oracle/*.c).

The reference's own tests hold no numeric vectors for this path ("parity
unpinned"), so the pins are (1) hand-packed blocks whose dequantised values have closed forms
straight from the block-format specification, (2) an independent numpy twin
(tests/np_twin.py), and (3) float64 restatements of the float ops.
"""
import numpy as np
import pytest

import np_twin as tw
import oracle_py as oq
from oracle_py import F16, IQ4_NL, Q2_K, Q3_K, Q4_0, Q4_K, Q5_0, Q5_K, Q6_K, Q8_0, Q8_K


def pack_k_scales(sc, mn):
    """Inverse of get_scale_min_k4: 8 six-bit scales + 8 six-bit mins -> 12 bytes."""
    s = np.zeros(12, np.uint8)
    for j in range(4):
        s[j] = (sc[j] & 63) | ((sc[j + 4] >> 4) << 6)
        s[j + 4] = (mn[j] & 63) | ((mn[j + 4] >> 4) << 6)
        s[j + 8] = (sc[j + 4] & 15) | ((mn[j + 4] & 15) << 4)
    return s


# ------------------------------------------------------------------ fp16
def test_fp16_to_fp32_all_codes():
    codes = np.arange(65536, dtype=np.uint16)
    want = codes.view(np.float16).astype(np.float32)
    got = np.array([oq.lib().oq_fp16_to_fp32(int(c)) for c in codes], dtype=np.float32)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all()


def test_fp32_to_fp16_round_to_nearest_even():
    rng = np.random.default_rng(1)
    xs = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, np.inf, -np.inf],
                 np.float32),
        # exact ties between adjacent halves
        (np.arange(1024, 2048, dtype=np.float32) + 0.5) * np.float32(2.0 ** -10),
    ])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).view(np.uint16)
    got = np.array([oq.lib().oq_fp32_to_fp16(float(x)) for x in xs], dtype=np.uint16)
    assert (got == want).all()


# ------------------------------------------------------------------ closed-form dequant
def test_q8_0_closed_form():
    b = np.zeros(1, tw.DT[Q8_0])
    b["d"] = 0.5
    b["qs"] = np.arange(-16, 16)
    y = oq.dequantize(Q8_0, b.view(np.uint8), 32)
    assert (y == np.arange(-16, 16) * 0.5).all()


def test_q4_0_closed_form():
    b = np.zeros(1, tw.DT[Q4_0])
    b["d"] = 2.0
    j = np.arange(16)
    b["qs"] = j | ((15 - j) << 4)
    y = oq.dequantize(Q4_0, b.view(np.uint8), 32)
    assert (y[:16] == (j - 8) * 2.0).all()
    assert (y[16:] == (15 - j - 8) * 2.0).all()


def test_q4_K_ramp_all_scales_one():
    b = np.zeros(1, tw.DT[Q4_K])
    b["d"] = 1.0
    b["dmin"] = 0.0
    b["scales"] = pack_k_scales([1] * 8, [0] * 8)
    l = np.arange(32)
    q = np.zeros((4, 32), np.uint8)
    for c in range(4):
        q[c] = ((l + c) & 15) | (((l + 2 * c + 1) & 15) << 4)
    b["qs"] = q.reshape(-1)
    y = oq.dequantize(Q4_K, b.view(np.uint8), 256).reshape(4, 2, 32)
    for c in range(4):
        assert (y[c, 0] == ((l + c) & 15)).all()
        assert (y[c, 1] == ((l + 2 * c + 1) & 15)).all()


def test_q4_K_scale_min_packing_edges():
    sc = [1, 2, 3, 63, 33, 47, 63, 16]
    mn = [5, 6, 7, 0, 60, 31, 17, 48]
    b = np.zeros(1, tw.DT[Q4_K])
    b["d"] = 0.25
    b["dmin"] = 0.5
    b["scales"] = pack_k_scales(sc, mn)
    b["qs"] = 0x93  # low nibble 3, high nibble 9
    y = oq.dequantize(Q4_K, b.view(np.uint8), 256).reshape(8, 32)
    for j in range(8):
        qv = 3 if j % 2 == 0 else 9
        assert (y[j] == 0.25 * sc[j] * qv - 0.5 * mn[j]).all(), j


def test_q5_K_high_bit():
    sc = [1, 1, 2, 2, 3, 3, 4, 4]
    mn = [0, 1, 0, 1, 0, 1, 0, 1]
    b = np.zeros(1, tw.DT[Q5_K])
    b["d"] = 1.0
    b["dmin"] = 1.0
    b["scales"] = pack_k_scales(sc, mn)
    b["qs"] = 0x21                      # low nibble 1, high nibble 2
    l = np.arange(32)
    b["qh"] = np.where(l % 2 == 0, 0b01100101, 0b10011010).astype(np.uint8)
    y = oq.dequantize(Q5_K, b.view(np.uint8), 256).reshape(8, 32)
    for j in range(8):
        bit = (np.where(l % 2 == 0, 0b01100101, 0b10011010) >> j) & 1
        qv = (1 if j % 2 == 0 else 2) + 16 * bit
        assert (y[j] == sc[j] * qv - mn[j]).all(), j


def test_q6_K_offset_edges():
    b = np.zeros(1, tw.DT[Q6_K])
    b["d"] = 0.5
    b["scales"] = np.array([1, -2, 3, -4, 5, -6, 7, -8, 127, -128, 11, -12, 13, -14, 15, -16], np.int8)
    # all-zero codes: q = -32 everywhere
    y = oq.dequantize(Q6_K, b.view(np.uint8), 256).reshape(2, 4, 2, 16)
    for n in range(2):
        for k in range(4):
            for h in range(2):
                s = int(b["scales"][0][8 * n + 2 * k + h])
                assert (y[n, k, h] == 0.5 * s * -32).all()
    # all-ones codes: q = 63 - 32 = 31
    b["ql"] = 0xFF
    b["qh"] = 0xFF
    y = oq.dequantize(Q6_K, b.view(np.uint8), 256).reshape(2, 4, 2, 16)
    for n in range(2):
        for k in range(4):
            for h in range(2):
                s = int(b["scales"][0][8 * n + 2 * k + h])
                assert (y[n, k, h] == 0.5 * s * 31).all()
    # one distinguishing code per quarter: ql low nibble -> quarters 0/1, high nibble -> 2/3, qh 2-bit fields
    b["ql"] = 0
    b["qh"] = 0
    ql = np.zeros(128, np.uint8)
    qh = np.zeros(64, np.uint8)
    ql[5] = 0x0A           # half 0, l=5: q1 low nibble = 10
    qh[5] = 0b11_10_01_00  # q1 hi=0, q2 hi=1, q3 hi=2, q4 hi=3
    ql[5 + 32] = 0x70      # l=5: q4 high nibble 7 (quarter 3), q2 low nibble 0
    b["ql"] = ql
    b["qh"] = qh
    y = oq.dequantize(Q6_K, b.view(np.uint8), 256)
    sc = b["scales"][0].astype(np.float32)
    assert y[5] == 0.5 * sc[0] * (10 - 32)
    assert y[5 + 32] == 0.5 * sc[2] * ((0 | (1 << 4)) - 32)
    assert y[5 + 64] == 0.5 * sc[4] * ((0 | (2 << 4)) - 32)
    assert y[5 + 96] == 0.5 * sc[6] * ((7 | (3 << 4)) - 32)


def test_q5_0_closed_form():
    """d = 0.5; element j: low nibble of qs[j] + 16 * bit j of qh, element j + 16: high nibble + 16 * bit j + 16; minus 16."""
    b = np.zeros(1, tw.DT[Q5_0])
    b["d"] = np.float16(0.5)
    qs = np.zeros(16, np.uint8)
    qs[3] = 0xA7                     # element 3: nibble 7, element 19: nibble 10
    b["qs"] = qs
    b["qh"] = np.uint32((1 << 3) | (1 << 31))     # fifth bit set for element 3 and element 31 (= bit 15 + 16)
    y = oq.dequantize(Q5_0, b.view(np.uint8), 32)
    assert y[3] == 0.5 * ((7 | 16) - 16) and y[19] == 0.5 * (10 - 16)
    assert y[31] == 0.5 * ((0 | 16) - 16) and y[0] == 0.5 * (0 - 16)


def test_iq4_nl_closed_form():
    """y = d * level[nibble] with the format's sixteen-level code book (ggml-common.h kvalues_iq4nl); low nibbles are elements 0..15, high ones 16..31.
    The dot against Q8_0: (d_x * d_y) * sum level * q8, one integer sum per block."""
    levels = [-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113]
    b = np.zeros(1, tw.DT[IQ4_NL])
    b["d"] = np.float16(0.25)
    b["qs"] = (np.arange(16, dtype=np.uint8) | ((15 - np.arange(16, dtype=np.uint8)) << 4))[None, :]     # element j: level j, element j + 16: level 15 - j
    y = oq.dequantize(IQ4_NL, b.view(np.uint8), 32)
    assert y[:16].tolist() == [0.25 * v for v in levels] and y[16:].tolist() == [0.25 * v for v in levels[::-1]]
    a = np.zeros(1, tw.DT[Q8_0])
    a["d"] = np.float16(2.0)
    a["qs"] = np.arange(32, dtype=np.int8)[None, :] - 16
    isum, msum = oq.vec_dot_int_partials(IQ4_NL, b.view(np.uint8), a.view(np.uint8), 32)
    want = sum(levels[j] * (j - 16) for j in range(16)) + sum(levels[15 - j] * j for j in range(16))
    assert isum.tolist() == [want] and msum.tolist() == [0]
    assert oq.vec_dot(IQ4_NL, b.view(np.uint8), a.view(np.uint8), 32) == 0.5 * want


def test_q4_0_dot_closed_form():
    """(nibble - 8) * q8 summed per block, then sumi * d_x * d_y left to right (ggml_vec_dot_q4_0_q8_0)."""
    b = np.zeros(1, tw.DT[Q4_0])
    b["d"] = np.float16(0.5)
    b["qs"] = np.full((1, 16), 0x3C, np.uint8)                   # elements 0..15: 12 - 8 = 4, elements 16..31: 3 - 8 = -5
    a = np.zeros(1, tw.DT[Q8_0])
    a["d"] = np.float16(0.125)
    a["qs"] = np.concatenate([np.full(16, 3, np.int8), np.full(16, -2, np.int8)])[None, :]
    isum, msum = oq.vec_dot_int_partials(Q4_0, b.view(np.uint8), a.view(np.uint8), 32)
    assert isum.tolist() == [16 * 4 * 3 + 16 * -5 * -2]
    assert oq.vec_dot(Q4_0, b.view(np.uint8), a.view(np.uint8), 32) == (16 * 12 + 16 * 10) * 0.5 * 0.125


def test_q2_K_closed_form():
    """Sub-block is = 8 n + 2 j + (l >= 16): value = d * (scales[is] & 15) * code - dmin * (scales[is] >> 4); code = bits 2 j .. 2 j + 1 of qs[32 n + l]."""
    b = np.zeros(1, tw.DT[Q2_K])
    b["d"] = np.float16(0.25); b["dmin"] = np.float16(0.5)
    sc = np.arange(16, dtype=np.uint8) | ((15 - np.arange(16, dtype=np.uint8)) << 4)      # scale = is, min = 15 - is
    b["scales"] = sc
    qs = np.zeros(64, np.uint8)
    qs[5] = 0b11_10_01_00            # n = 0, l = 5: codes 0, 1, 2, 3 for j = 0..3 (elements 5, 37, 69, 101)
    qs[32 + 20] = 0b01_00_11_10      # n = 1, l = 20: codes 2, 3, 0, 1 (elements 128 + 20, + 52, + 84, + 116)
    b["qs"] = qs
    y = oq.dequantize(Q2_K, b.view(np.uint8), 256)
    for j, code in enumerate([0, 1, 2, 3]):
        is_ = 2 * j                  # n = 0, l = 5 < 16
        assert y[32 * j + 5] == np.float32(0.25 * is_) * code - np.float32(0.5 * (15 - is_)), j
    for j, code in enumerate([2, 3, 0, 1]):
        is_ = 8 + 2 * j + 1          # n = 1, l = 20 >= 16
        assert y[128 + 32 * j + 20] == np.float32(0.25 * is_) * code - np.float32(0.5 * (15 - is_)), j
    assert y[0] == -np.float32(0.5 * 15)          # code 0, sub-block 0: just the minimum


def test_q3_K_closed_form():
    """value = d * (scale[is] - 32) * (code - (hmask bit ? 0 : 4)); scale is: low nibble from bytes 0..7 (is < 8: low, is >= 8: high nibble of byte is - 8),
    high two bits from byte 8 + is % 4 at bit pair is // 4; hmask bit of element (n, j, l) is bit 4 n + j of hmask[l]."""
    b = np.zeros(1, tw.DT[Q3_K])
    b["d"] = np.float16(0.5)
    want = np.array([0, 1, 31, 32, 33, 47, 48, 63, 5, 17, 29, 36, 44, 52, 60, 62])      # the 16 six-bit scales
    s = np.zeros(12, np.uint8)
    for j in range(16):
        if j < 8:
            s[j] |= want[j] & 15
        else:
            s[j - 8] |= (want[j] & 15) << 4
        s[8 + j % 4] |= (want[j] >> 4) << (2 * (j // 4))
    b["scales"] = s
    assert tw.q3_scales(s[None, :])[0].tolist() == want.tolist()
    qs = np.zeros(64, np.uint8); hm = np.zeros(32, np.uint8)
    qs[7] = 0b10_01_11_00            # n = 0, l = 7: codes 0, 3, 1, 2 (elements 7, 39, 71, 103)
    hm[7] = 0b0000_0101              # high bit present for j = 0 and j = 2 of half 0
    qs[32 + 18] = 0b00_11_00_01      # n = 1, l = 18: codes 1, 0, 3, 0
    hm[18] = 0b1010_0000             # high bit present for j = 1 and j = 3 of half 1 (bits 5, 7)
    b["qs"] = qs; b["hmask"] = hm
    y = oq.dequantize(Q3_K, b.view(np.uint8), 256)
    for j, (code, hb) in enumerate([(0, 1), (3, 0), (1, 1), (2, 0)]):
        assert y[32 * j + 7] == np.float32(0.5 * (want[2 * j] - 32)) * (code - (0 if hb else 4)), j
    for j, (code, hb) in enumerate([(1, 0), (0, 1), (3, 0), (0, 1)]):
        assert y[128 + 32 * j + 18] == np.float32(0.5 * (want[8 + 2 * j + 1] - 32)) * (code - (0 if hb else 4)), j
    assert y[0] == np.float32(0.5 * (want[0] - 32)) * -4        # code 0, no high bit: -4


# ------------------------------------------------------------------ twin agreement on random blocks
@pytest.mark.parametrize("t,be,bb", [(Q4_0, 32, 18), (Q8_0, 32, 34), (Q4_K, 256, 144), (Q5_K, 256, 176), (Q6_K, 256, 210), (Q5_0, 32, 22), (Q2_K, 256, 84),
                                     (Q3_K, 256, 110), (IQ4_NL, 32, 18)])
def test_dequant_matches_twin_bit_exact(t, be, bb):
    rng = np.random.default_rng(100 + t)
    nb = 64
    raw = rng.integers(0, 256, nb * bb, dtype=np.uint8)
    blk = raw.view(tw.DT[t])
    blk["d"] = rng.uniform(-2, 2, nb).astype("<f2")
    if t in (Q4_K, Q5_K, Q2_K):
        blk["dmin"] = rng.uniform(-2, 2, nb).astype("<f2")
    a = oq.dequantize(t, raw, nb * be)
    b = tw.dequantize(t, raw)
    assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist()


# ------------------------------------------------------------------ activation quantisation
def test_q8_K_closed_form_and_ties():
    x = np.arange(256, dtype=np.float32) - 128.0          # max |x| = 128 at element 0 (x = -128)
    raw = oq.quantize(Q8_K, x)
    b = raw.view(tw.DT[Q8_K])
    iscale = np.float32(-127.0) / np.float32(-128.0)
    assert b["d"][0] == np.float32(1.0) / iscale
    want = np.minimum(np.rint((iscale * x).astype(np.float32)), 127).astype(np.int8)
    assert (b["qs"][0] == want).all()
    assert b["qs"][0][0] == -127
    assert (b["bsums"][0] == want.astype(np.int32).reshape(16, 16).sum(1)).all()
    # ties go to even: make iscale exactly 1 (max = -127) and put x.5 values in
    x = np.zeros(256, np.float32)
    x[0] = -127.0
    x[1:7] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5]
    q = oq.quantize(Q8_K, x).view(tw.DT[Q8_K])["qs"][0]
    assert q[:7].tolist() == [-127, 0, 2, 2, 0, -2, -2]
    # positive maximum -> negative iscale -> max element maps to -127, d negative
    x = np.zeros(256, np.float32)
    x[3] = 4.0
    b = oq.quantize(Q8_K, x).view(tw.DT[Q8_K])
    assert b["qs"][0][3] == -127 and b["d"][0] < 0
    assert np.isclose(b["d"][0] * -127, 4.0)
    # all-zero block
    b = oq.quantize(Q8_K, np.zeros(256, np.float32)).view(tw.DT[Q8_K])
    assert b["d"][0] == 0 and not b["qs"].any() and not b["bsums"].any()


def test_q8_0_closed_form_round_half_away():
    x = np.zeros(32, np.float32)
    x[0] = 127.0                                   # d = 1
    x[1:5] = [0.5, 1.5, -0.5, -2.5]
    b = oq.quantize(Q8_0, x).view(tw.DT[Q8_0])
    assert b["d"][0] == 1.0
    assert b["qs"][0][:5].tolist() == [127, 1, 2, -1, -3]


def test_activation_quant_matches_twin():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(4096) * rng.uniform(0.1, 10)).astype(np.float32)
    assert (oq.quantize(Q8_K, x) == tw.quantize_q8_K(x)).all()
    assert (oq.quantize(Q8_0, x) == tw.quantize_q8_0(x)).all()


# ------------------------------------------------------------------ dot products
@pytest.mark.parametrize("t,bb", [(Q8_0, 34), (Q4_K, 144), (Q5_K, 176), (Q6_K, 210), (Q5_0, 22), (Q2_K, 84), (Q3_K, 110), (Q4_0, 18), (IQ4_NL, 18)])
def test_vec_dot_int_partials_and_value(t, bb):
    rng = np.random.default_rng(200 + t)
    K = 2048
    be = 32 if t in (Q8_0, Q5_0, Q4_0, IQ4_NL) else 256
    raw = rng.integers(0, 256, K // be * bb, dtype=np.uint8)
    blk = raw.view(tw.DT[t])
    blk["d"] = rng.uniform(0.5, 1.5, K // be).astype("<f2") * np.float16(1e-2)
    if t in (Q4_K, Q5_K, Q2_K):
        blk["dmin"] = rng.uniform(0.5, 1.5, K // be).astype("<f2") * np.float16(1e-2)
    x = rng.standard_normal(K).astype(np.float32)
    at = oq.vec_dot_type(t)
    act = oq.quantize(at, x)
    isum, msum = oq.vec_dot_int_partials(t, raw, act, K)
    ti, tm = tw.int_partials(t, raw, act)
    assert (isum == ti).all() and (msum == tm).all()
    got = oq.vec_dot(t, raw, act, K)
    want = float(np.dot(tw.dequantize(t, raw).astype(np.float64), tw.dequantize(at, act).astype(np.float64)))
    assert abs(got - want) <= 2e-5 * max(1.0, abs(want)) + 1e-4 * np.abs(tw.dequantize(t, raw)).max()


def test_vec_dot_closed_form_q4_K():
    b = np.zeros(2, tw.DT[Q4_K])
    b["d"] = 1.0
    b["dmin"] = 1.0
    b["scales"] = pack_k_scales([2] * 8, [3] * 8)
    b["qs"] = 0x55                                           # every weight = 2*5 - 3 = 7
    x = np.ones(512, np.float32)                             # q8_K: iscale = -127, q = -127, d = -1/127
    act = oq.quantize(Q8_K, x)
    got = oq.vec_dot(Q4_K, b.view(np.uint8), act, 512)
    assert np.isclose(got, 7.0 * 512, rtol=1e-6)
    isum, msum = oq.vec_dot_int_partials(Q4_K, b.view(np.uint8), act, 512)
    assert isum.tolist() == [2 * 5 * -127 * 256] * 2
    assert msum.tolist() == [3 * -127 * 256] * 2


# ------------------------------------------------------------------ float ops vs float64
def test_rms_norm_silu_softmax():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(4096).astype(np.float32) * 3
    y = oq.rms_norm(x, 1e-5)
    want = x.astype(np.float64) / np.sqrt((x.astype(np.float64) ** 2).mean() + 1e-5)
    assert np.allclose(y, want, rtol=2e-6, atol=1e-7)
    assert np.allclose(oq.silu(x), x.astype(np.float64) / (1 + np.exp(-x.astype(np.float64))), rtol=2e-6, atol=1e-7)
    m = np.where(rng.random(4096) < 0.2, -np.inf, 0).astype(np.float32)
    p = oq.soft_max(x, m, 0.125)
    z = x.astype(np.float64) * 0.125 + m
    w = np.exp(z - z.max())
    assert np.allclose(p, w / w.sum(), rtol=1e-5, atol=1e-9)
    assert abs(p.sum() - 1) < 1e-5


@pytest.mark.parametrize("base", [1e4, 5e5])
@pytest.mark.parametrize("pos", [0, 1, 4095])
def test_rope_norm_vs_float64(base, pos):
    rng = np.random.default_rng(5)
    H, D = 4, 128
    x = rng.standard_normal((H, D)).astype(np.float32)
    y = oq.rope(x, H, D, pos, base)
    if pos == 0:
        assert (y == x).all()
    i = np.arange(D // 2)
    th = pos * np.float64(base) ** (-2.0 * i / D)
    want = np.empty((H, D))
    want[:, 0::2] = x[:, 0::2] * np.cos(th) - x[:, 1::2] * np.sin(th)
    want[:, 1::2] = x[:, 0::2] * np.sin(th) + x[:, 1::2] * np.cos(th)
    # f32 theta recurrence: |dtheta| <~ pos * 64 ulp
    assert np.abs(y - want).max() < max(1e-6, pos * 2e-5 * 4)
    # rotation preserves pair norms
    assert np.allclose(y[:, 0::2] ** 2 + y[:, 1::2] ** 2, x[:, 0::2] ** 2 + x[:, 1::2] ** 2, rtol=1e-5)


def test_yarn_corr_dims_closed_form():
    # n_dims * ln(n_ctx_orig / (n_rot * 2 pi)) / (2 ln base): Llama-2's 128 dims, 4096 positions, base 1e4 -> floor(20.94) = 20, ceil(45.03) = 46
    assert oq.yarn_corr_dims(128, 4096, 1e4) == (20.0, 46.0)
    # clamped to [0, n_dims - 1]
    lo, hi = oq.yarn_corr_dims(64, 32, 1e4)
    assert lo == 0.0 and hi == 6.0
    assert oq.yarn_corr_dims(64, 1 << 30, 1e2)[1] == 63.0


@pytest.mark.parametrize("neox", [False, True])
@pytest.mark.parametrize("pos", [0, 3, 700])
def test_rope_yarn_vs_float64(neox, pos):
    """YaRN (llama.cpp rope_yarn, published formula): theta = interp * (1 - mix) + extrap * mix with mix = 1 - clamp((pair - lo) / (hi - lo)), magnitude
    attn_factor * (1 + 0.1 ln(1 / freq_scale)); against a float64 restatement, plus its limits (no mix -> linear scaling; low pairs -> the unscaled angle)."""
    rng = np.random.default_rng(6)
    H, D, base, fs, attn = 3, 128, 1e4, 0.25, 0.9
    lo, hi = oq.yarn_corr_dims(D, 256, base)
    assert 0 < lo < hi < D // 2
    x = rng.standard_normal((H, D)).astype(np.float32)
    y = oq.rope_yarn(x, H, D, pos, base, fs, 1.0, attn, lo, hi, neox=neox)
    i = np.arange(D // 2)
    extrap = pos * np.float64(base) ** (-2.0 * i / D)
    mix = 1.0 - np.clip((i - lo) / max(0.001, hi - lo), 0.0, 1.0)
    th = fs * extrap * (1 - mix) + extrap * mix
    mag = attn * (1.0 + 0.1 * np.log(1.0 / fs))
    a, b = (x[:, :D // 2], x[:, D // 2:]) if neox else (x[:, 0::2], x[:, 1::2])
    ya, yb = (y[:, :D // 2], y[:, D // 2:]) if neox else (y[:, 0::2], y[:, 1::2])
    assert np.abs(ya - mag * (a * np.cos(th) - b * np.sin(th))).max() < max(2e-6, pos * 8e-5)
    assert np.abs(yb - mag * (a * np.sin(th) + b * np.cos(th))).max() < max(2e-6, pos * 8e-5)
    # pairs below corr_lo keep the original angle, pairs above corr_hi take the interpolated one: both up to the magnitude
    plain = oq.rope(x, H, D, pos, base, neox=neox)
    lin = oq.rope(x, H, D, pos, base, neox=neox, freq_scale=fs)
    pa, la = (plain[:, :D // 2], lin[:, :D // 2]) if neox else (plain[:, 0::2], lin[:, 0::2])
    k0, k1 = int(lo) + 1, int(hi)
    assert np.allclose(ya[:, :k0], np.float32(mag) * pa[:, :k0], rtol=0, atol=1e-6)
    assert np.allclose(ya[:, k1:], np.float32(mag) * la[:, k1:], rtol=0, atol=1e-6)
    # ext_factor 0, attn_factor 1: exactly the linear-scaling rotation
    assert (oq.rope_yarn(x, H, D, pos, base, fs, 0.0, 1.0, lo, hi, neox=neox) == lin).all()


@pytest.mark.parametrize("tk,tv,tol", [(F16, F16, 3e-3), (Q8_0, Q8_0, 3e-5), (Q4_0, Q4_0, 3e-5)])
def test_flash_attn_vs_float64(tk, tv, tol):
    rng = np.random.default_rng(11)
    H, G, D, n_ctx = 8, 2, 64, 96
    q = rng.standard_normal((H, D)).astype(np.float32)
    kf = rng.standard_normal((n_ctx, G * D)).astype(np.float32)
    vf = rng.standard_normal((n_ctx, G * D)).astype(np.float32)
    kc = np.stack([oq.quantize(tk, r) for r in kf])
    vc = np.stack([oq.quantize(tv, r) for r in vf])
    cells = np.array([c for c in range(n_ctx) if c % 7 != 3], np.int32)
    scale = 1 / np.sqrt(D)
    out = oq.flash_attn(q, H, G, D, tk, kc, tv, vc, cells, scale)
    # float64 attention over the *stored* (quantised) K/V and the Q as the kernel sees it
    kd = np.stack([oq.dequantize(tk, r, G * D) for r in kc]).astype(np.float64).reshape(n_ctx, G, D)
    vd = np.stack([oq.dequantize(tv, r, G * D) for r in vc]).astype(np.float64).reshape(n_ctx, G, D)
    qt = oq.vec_dot_type(tk)
    qd = np.stack([oq.dequantize(qt, oq.quantize(qt, q[h]), D) for h in range(H)]).astype(np.float64)
    for h in range(H):
        g = h // (H // G)
        s = kd[cells, g] @ qd[h] * scale
        w = np.exp(s - s.max())
        want = (w[:, None] * vd[cells, g]).sum(0) / w.sum()
        assert np.abs(out[h] - want).max() < tol, (h, np.abs(out[h] - want).max())


def test_mul_mat_matches_dequantised_float64():
    rng = np.random.default_rng(21)
    N, K, T = 48, 1024, 3
    for t, bb in [(Q4_K, 144), (Q5_K, 176), (Q6_K, 210), (Q8_0, 34)]:
        be = 32 if t == Q8_0 else 256
        raw = rng.integers(0, 256, N * K // be * bb, dtype=np.uint8)
        blk = raw.view(tw.DT[t])
        blk["d"] = (rng.uniform(0.5, 1.5, blk.size) * 1e-3).astype("<f2")
        if t in (Q4_K, Q5_K):
            blk["dmin"] = (rng.uniform(0.5, 1.5, blk.size) * 1e-3).astype("<f2")
        x = rng.standard_normal((T, K)).astype(np.float32)
        y = oq.mul_mat(t, raw, N, K, x)
        Wd = tw.dequantize(t, raw).reshape(N, K).astype(np.float64)
        at = oq.vec_dot_type(t)
        xd = np.stack([tw.dequantize(at, oq.quantize(at, r)) for r in x]).astype(np.float64)
        want = xd @ Wd.T
        assert np.abs(y - want).max() <= 1e-5 * np.abs(want).max() + 1e-6
