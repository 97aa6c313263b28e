"""Independent numpy restatement of the ggml block formats, used to cross-check the
C oracle.  Written from the format specification, vectorised over blocks, sharing no code with
oracle/*.c.  Test infrastructure only."""
from __future__ import annotations

import numpy as np

F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 8, 12, 13, 14, 15
Q5_0, Q2_K, Q3_K, IQ4_NL = 6, 10, 11, 20
IQ4NL_LEVELS = np.array([-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113], np.int32)   # the format's code book

DT = {
    Q4_0: np.dtype([("d", "<f2"), ("qs", "u1", 16)]),
    Q8_0: np.dtype([("d", "<f2"), ("qs", "i1", 32)]),
    Q4_K: np.dtype([("d", "<f2"), ("dmin", "<f2"), ("scales", "u1", 12), ("qs", "u1", 128)]),
    Q5_K: np.dtype([("d", "<f2"), ("dmin", "<f2"), ("scales", "u1", 12), ("qh", "u1", 32), ("qs", "u1", 128)]),
    Q6_K: np.dtype([("ql", "u1", 128), ("qh", "u1", 64), ("scales", "i1", 16), ("d", "<f2")]),
    Q8_K: np.dtype([("d", "<f4"), ("qs", "i1", 256), ("bsums", "<i2", 16)]),
    Q5_0: np.dtype([("d", "<f2"), ("qh", "<u4"), ("qs", "u1", 16)]),
    IQ4_NL: np.dtype([("d", "<f2"), ("qs", "u1", 16)]),
    Q2_K: np.dtype([("scales", "u1", 16), ("qs", "u1", 64), ("d", "<f2"), ("dmin", "<f2")]),
    Q3_K: np.dtype([("hmask", "u1", 32), ("qs", "u1", 64), ("scales", "u1", 12), ("d", "<f2")]),
}


def q3_scales(scales12: np.ndarray) -> np.ndarray:
    """[nb,12] packed bytes -> [nb,16] 6-bit scales (0..63; the format subtracts 32).  Scale j: low 4 bits = nibble j of the first 8 bytes (bytes 0-7 low
    nibbles are scales 0-7, high nibbles scales 8-15), high 2 bits = bit pair (j // 4) of byte 8 + j % 4."""
    s = scales12.astype(np.int32)
    out = np.empty(s.shape[:-1] + (16,), np.int32)
    for j in range(16):
        low = (s[..., j] & 15) if j < 8 else (s[..., j - 8] >> 4)
        high = (s[..., 8 + j % 4] >> (2 * (j // 4))) & 3
        out[..., j] = low | (high << 4)
    return out


def k_scales(scales12: np.ndarray):
    """[nb,12] packed bytes -> (sc[nb,8], mn[nb,8]) 6-bit values."""
    s = scales12.astype(np.int32)
    sc = np.empty(s.shape[:-1] + (8,), np.int32)
    mn = np.empty_like(sc)
    sc[..., :4] = s[..., 0:4] & 63
    mn[..., :4] = s[..., 4:8] & 63
    sc[..., 4:] = (s[..., 8:12] & 0x0F) | ((s[..., 0:4] >> 6) << 4)
    mn[..., 4:] = (s[..., 8:12] >> 4) | ((s[..., 4:8] >> 6) << 4)
    return sc, mn


def unpack_ints(t: int, raw: np.ndarray) -> np.ndarray:
    """Integer weight codes per block: [nb, 256] (K-quants) / [nb, 32] (q8_0, q4_0 with -8 applied)."""
    b = raw.view(np.uint8).reshape(-1).view(DT[t])
    if t == Q8_0:
        return b["qs"].astype(np.int32)
    if t == Q4_0:
        q = b["qs"].astype(np.int32)
        return np.concatenate([(q & 15) - 8, (q >> 4) - 8], axis=1)
    if t in (Q4_K, Q5_K):
        q = b["qs"].astype(np.int32).reshape(-1, 4, 32)
        lo, hi = q & 15, q >> 4
        if t == Q5_K:
            h = b["qh"].astype(np.int32)[:, None, :]           # [nb,1,32]
            c = np.arange(4)[None, :, None]
            lo = lo + (((h >> (2 * c)) & 1) << 4)
            hi = hi + (((h >> (2 * c + 1)) & 1) << 4)
        return np.stack([lo, hi], axis=2).reshape(-1, 256)
    if t == IQ4_NL:
        q = b["qs"].astype(np.int32)
        return np.concatenate([IQ4NL_LEVELS[q & 15], IQ4NL_LEVELS[q >> 4]], axis=1)
    if t == Q5_0:
        q = b["qs"].astype(np.int32)
        h = b["qh"].astype(np.int64)[:, None]
        j = np.arange(16)[None, :]
        lo = (q & 15) | (((h >> j) & 1) << 4)
        hi = (q >> 4) | (((h >> (j + 16)) & 1) << 4)
        return (np.concatenate([lo, hi], axis=1) - 16).astype(np.int32)
    if t in (Q2_K, Q3_K):
        # byte l of 32-byte half n holds elements 128 n + 32 j + l at bits 2 j .. 2 j + 1
        q = b["qs"].astype(np.int32).reshape(-1, 2, 1, 32)
        jj = np.arange(4)[None, None, :, None]
        w = (q >> (2 * jj)) & 3                                  # [nb, 2, 4, 32]
        if t == Q3_K:
            hm = b["hmask"].astype(np.int32)[:, None, None, :]   # [nb,1,1,32]; bit 4 n + j belongs to element (n, j, l)
            bit = 4 * np.arange(2)[None, :, None, None] + jj
            w = w - np.where((hm >> bit) & 1, 0, 4)
        return w.reshape(-1, 256)
    if t == Q6_K:
        ql = b["ql"].astype(np.int32).reshape(-1, 2, 64)
        qh = b["qh"].astype(np.int32).reshape(-1, 2, 32)
        q1 = (ql[:, :, :32] & 15) | (((qh >> 0) & 3) << 4)
        q2 = (ql[:, :, 32:] & 15) | (((qh >> 2) & 3) << 4)
        q3 = (ql[:, :, :32] >> 4) | (((qh >> 4) & 3) << 4)
        q4 = (ql[:, :, 32:] >> 4) | (((qh >> 6) & 3) << 4)
        return (np.stack([q1, q2, q3, q4], axis=2) - 32).reshape(-1, 256)
    raise ValueError(t)


def dequantize(t: int, raw: np.ndarray) -> np.ndarray:
    if t == F32:
        return raw.view("<f4").astype(np.float32)
    if t == F16:
        return raw.view("<f2").astype(np.float32)
    b = raw.view(np.uint8).reshape(-1).view(DT[t])
    if t == Q8_K:
        return (b["d"][:, None] * b["qs"].astype(np.float32)).reshape(-1)
    q = unpack_ints(t, raw).astype(np.float32)
    d = b["d"].astype(np.float32)
    if t in (Q8_0, Q4_0, Q5_0, IQ4_NL):
        return (q * d[:, None]).reshape(-1)
    if t == Q2_K:
        sc = b["scales"].astype(np.int32)
        dm = b["dmin"].astype(np.float32)
        dl = d[:, None] * (sc & 15).astype(np.float32)           # one (scale, min) nibble pair per 16 elements
        ml = dm[:, None] * (sc >> 4).astype(np.float32)
        y = dl[:, :, None] * q.reshape(-1, 16, 16) - ml[:, :, None]
        return y.astype(np.float32).reshape(-1)
    if t == Q3_K:
        dl = d[:, None] * (q3_scales(b["scales"]) - 32).astype(np.float32)
        return (dl[:, :, None] * q.reshape(-1, 16, 16)).astype(np.float32).reshape(-1)
    if t in (Q4_K, Q5_K):
        sc, mn = k_scales(b["scales"])
        dm = b["dmin"].astype(np.float32)
        ds = (d[:, None] * sc.astype(np.float32))           # f32 products, as the spec evaluates them
        ms = (dm[:, None] * mn.astype(np.float32))
        y = ds[:, :, None] * q.reshape(-1, 8, 32) - ms[:, :, None]
        return y.astype(np.float32).reshape(-1)
    if t == Q6_K:
        sc = b["scales"].astype(np.float32)                 # [nb,16]
        # element e of half n, quarter k (0..3), l (0..31): scale index 8n + 2k + l//16
        qq = q.reshape(-1, 2, 4, 32)
        idx = (8 * np.arange(2)[:, None, None] + 2 * np.arange(4)[None, :, None] + (np.arange(32) // 16)[None, None, :])
        s = sc[:, idx]                                      # [nb,2,4,32]
        y = (d[:, None, None, None] * s) * qq
        return y.astype(np.float32).reshape(-1)
    raise ValueError(t)


def quantize_q8_0(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
    v = x * idv[:, None]
    q = np.where(v >= 0, np.floor(v + np.float32(0.5)), -np.floor(-v + np.float32(0.5)))   # roundf: half away from zero
    out = np.zeros(x.shape[0], DT[Q8_0])
    out["d"] = d.astype("<f2")
    out["qs"] = q.astype(np.int8)
    return out.view(np.uint8).reshape(-1)


def quantize_q8_K(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float32).reshape(-1, 256)
    out = np.zeros(x.shape[0], DT[Q8_K])
    idx = np.abs(x).argmax(axis=1)                          # first element with the largest magnitude
    vmax = x[np.arange(x.shape[0]), idx]
    nz = vmax != 0
    iscale = np.zeros_like(vmax)
    iscale[nz] = np.float32(-127.0) / vmax[nz]
    v = (iscale[:, None] * x).astype(np.float32)
    q = np.minimum(np.rint(v), 127).astype(np.int32)        # rint = round-half-even
    q[~nz] = 0
    out["qs"] = q.astype(np.int8)
    out["bsums"] = q.reshape(-1, 16, 16).sum(axis=2).astype(np.int16)
    dd = np.zeros_like(vmax)
    dd[nz] = np.float32(1.0) / iscale[nz]
    out["d"] = dd
    return out.view(np.uint8).reshape(-1)


def int_partials(t: int, w_raw: np.ndarray, a_raw: np.ndarray):
    """(isum[nb], msum[nb]) of one weight row against a q8_K / q8_0 activation row (exact ints)."""
    w = unpack_ints(t, w_raw).astype(np.int64)
    if t in (Q8_0, Q5_0, Q4_0, IQ4_NL):
        a = a_raw.view(np.uint8).reshape(-1).view(DT[Q8_0])["qs"].astype(np.int64)
        return (w * a).sum(axis=1).astype(np.int32), np.zeros(w.shape[0], np.int32)
    ab = a_raw.view(np.uint8).reshape(-1).view(DT[Q8_K])
    a = ab["qs"].astype(np.int64)
    wb = w_raw.view(np.uint8).reshape(-1).view(DT[t])
    if t in (Q4_K, Q5_K):
        sc, mn = k_scales(wb["scales"])
        sub = (w * a).reshape(-1, 8, 32).sum(axis=2)
        isum = (sub * sc).sum(axis=1)
        bs = ab["bsums"].astype(np.int64).reshape(-1, 8, 2).sum(axis=2)
        msum = (bs * mn).sum(axis=1)
        return isum.astype(np.int32), msum.astype(np.int32)
    if t == Q6_K:
        sub = (w * a).reshape(-1, 16, 16).sum(axis=2)
        isum = (sub * wb["scales"].astype(np.int64)).sum(axis=1)
        return isum.astype(np.int32), np.zeros(w.shape[0], np.int32)
    if t == Q2_K:
        sc = wb["scales"].astype(np.int64)
        sub = (w * a).reshape(-1, 16, 16).sum(axis=2)
        isum = (sub * (sc & 15)).sum(axis=1)
        msum = (ab["bsums"].astype(np.int64) * (sc >> 4)).sum(axis=1)
        return isum.astype(np.int32), msum.astype(np.int32)
    if t == Q3_K:
        sub = (w * a).reshape(-1, 16, 16).sum(axis=2)
        isum = (sub * (q3_scales(wb["scales"]).astype(np.int64) - 32)).sum(axis=1)
        return isum.astype(np.int32), np.zeros(w.shape[0], np.int32)
    raise ValueError(t)
