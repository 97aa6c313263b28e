// mmvq.hip — quantised mat-vec  y[t][r] = dot(W[r,:], act_q8[t,:])  for the decode path.
//
// Stands in for ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K / ggml_vec_dot_q8_0_q8_0 as driven by
// ggml_compute_forward_mul_mat (upstream ggml-cpu; absent from /root/reference — SURVEY.md §8a rows a8/a9),
// reached from the reference through llama_decode (src/llama_server_context.cc:1635).
//
// Design (gfx950, wave64, HBM-bound):
//  * one wave owns a PAIR of weight rows at a time and keeps their partial sums in registers;
//  * every lane issues 16-byte coalesced loads: 8 consecutive lanes cover the 128 quant bytes of one
//    256-weight super-block, 8 super-blocks per wave pass; headers (d, dmin, 6-bit scales) are one more
//    16-byte load that the 8 lanes of a super-block share through the coalescer;
//  * 2 rows x 2 passes are in flight before the first use (>= 8 x 16 B per lane outstanding);
//  * the int8 activation vector (Q8_K: int8 + f32 d per 256 + int16 sums per 16; Q8_0: int8 + f16 d per 32)
//    is staged once per workgroup into LDS and read with ds_read_b128;
//  * integer work is v_dot4_i32_i8; integer partial sums are exactly the CPU backend's, only the final
//    f32 summation order differs (lane partials are reduced with a wave butterfly).
//  * epilogues fuse the residual add (attn_output / ffn_down) and SwiGLU (ffn_gate + ffn_up).
#include <cstdlib>
#include "kernels.h"
#include "quant_dev.h"

#ifndef MMVQ_RPU
#define MMVQ_RPU 2      // weight rows per wave unit (x 2 passes): loads in flight per lane = RPU * 2 * loads-per-item
#endif
#ifndef MMVQ_WAVES
#define MMVQ_WAVES 3    // launch bound: waves per SIMD
#endif
#ifndef MMVQ_BLOCKS_PER_CU
#define MMVQ_BLOCKS_PER_CU 4
#endif

namespace mi355 {

// ------------------------------------------------------------------------------------------
// LDS view of the staged activations of one workgroup
struct ActLds {
    const int8_t *qs;      // [T][K]      q8_K codes
    const float *d;        // [T][K/256]
    const int16_t *bs;     // [T][K/16]
    const int8_t *qs0;     // [T][K]      q8_0 codes
    const uint16_t *d0;    // [T][K/32]   f16
    int K;                 // row stride of the staged codes (elements): K, or the chunk length for the tiled kernel;
                           // for a staged K-chunk the pointers are pre-biased by -k0 so absolute block indices still work
};

__device__ __forceinline__ uint4 ld16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
// weight stream: read once per token by exactly one wave -> non-temporal (does not displace the activations in L2)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16w(const void *p) {
#ifdef MMVQ_NO_NT
    return *reinterpret_cast<const uint4 *>(p);
#else
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

template <int TYPE> struct Item;

// ---- Q4_K : ggml super-block  [d f16][dmin f16][scales 12][qs 128]  (144 B, 16-B aligned) ----
template <> struct Item<T_Q4_K> {
    uint4 hdr;
    static constexpr int EPP = 2048;  // elements per wave pass
    uint4 q;
    int sc_lo, sc_hi, m_lo, m_hi;
    float d, dmin;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        sb = pass * 8 + (lane >> 3);
        valid = sb < (K >> 8);
        if (valid) {
            const uint8_t *b = row + (size_t)sb * 144;
            hdr = ld16w(b);
            q = ld16w(b + 16 + (lane & 7) * 16);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (!valid) return;
        const int v = lane & 7, c = v >> 1;
        d = h2f((uint16_t)(hdr.x & 0xffff));
        dmin = h2f((uint16_t)(hdr.x >> 16));
        const int sh = (c & 1) * 16;
        const uint32_t a16 = (hdr.y >> sh) & 0xffff, b16 = (hdr.z >> sh) & 0xffff, c16 = (hdr.w >> sh) & 0xffff;
        uint32_t sc, mn;
        if (c < 2) {
            sc = a16 & 0x3f3f;
            mn = b16 & 0x3f3f;
        } else {
            sc = (c16 & 0x0f0f) | ((a16 >> 2) & 0x3030);
            mn = ((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030);
        }
        sc_lo = sc & 0xff; sc_hi = sc >> 8;
        m_lo = mn & 0xff;  m_hi = mn >> 8;
    }
    // integer partials of this lane for token t
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, c = v >> 1, h = v & 1;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 64 * c + 16 * h;
        const uint4 lo = ld16(a), hi = ld16(a + 32);
        int dl = 0, dh = 0;
        dl = dot4(q.x & 0x0f0f0f0f, lo.x, dl); dh = dot4((q.x >> 4) & 0x0f0f0f0f, hi.x, dh);
        dl = dot4(q.y & 0x0f0f0f0f, lo.y, dl); dh = dot4((q.y >> 4) & 0x0f0f0f0f, hi.y, dh);
        dl = dot4(q.z & 0x0f0f0f0f, lo.z, dl); dh = dot4((q.z >> 4) & 0x0f0f0f0f, hi.z, dh);
        dl = dot4(q.w & 0x0f0f0f0f, lo.w, dl); dh = dot4((q.w >> 4) & 0x0f0f0f0f, hi.w, dh);
        isum = __mul24(sc_lo, dl) + __mul24(sc_hi, dh);
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 4 * c + h;
        msum = __mul24(m_lo, (int)bs[0]) + __mul24(m_hi, (int)bs[2]);
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum - (dmin * yd) * (float)msum;
    }
};

// ---- Q5_K : [d][dmin][scales 12][qh 32][qs 128]  (176 B, 16-B aligned) ----------------------
template <> struct Item<T_Q5_K> {
    uint4 hdr;
    static constexpr int EPP = 2048;
    uint4 q, qh;
    int sc_lo, sc_hi, m_lo, m_hi;
    float d, dmin;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        sb = pass * 8 + (lane >> 3);
        valid = sb < (K >> 8);
        if (valid) {
            const uint8_t *b = row + (size_t)sb * 176;
            hdr = ld16w(b);
            qh = ld16w(b + 16 + (lane & 1) * 16);
            q = ld16w(b + 48 + (lane & 7) * 16);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (!valid) return;
        const int v = lane & 7, c = v >> 1;
        d = h2f((uint16_t)(hdr.x & 0xffff));
        dmin = h2f((uint16_t)(hdr.x >> 16));
        const int sh = (c & 1) * 16;
        const uint32_t a16 = (hdr.y >> sh) & 0xffff, b16 = (hdr.z >> sh) & 0xffff, c16 = (hdr.w >> sh) & 0xffff;
        uint32_t sc, mn;
        if (c < 2) {
            sc = a16 & 0x3f3f;
            mn = b16 & 0x3f3f;
        } else {
            sc = (c16 & 0x0f0f) | ((a16 >> 2) & 0x3030);
            mn = ((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030);
        }
        sc_lo = sc & 0xff; sc_hi = sc >> 8;
        m_lo = mn & 0xff;  m_hi = mn >> 8;
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, c = v >> 1, h = v & 1;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 64 * c + 16 * h;
        const uint4 lo = ld16(a), hi = ld16(a + 32);
        const int s0 = 2 * c, s1 = 2 * c + 1;   // bit of qh holding the 5th bit of the low / high nibble plane
        int dl = 0, dh = 0;
#define Q5_LO(w, hw) (((w) & 0x0f0f0f0f) | ((((hw) >> s0) & 0x01010101u) << 4))
#define Q5_HI(w, hw) ((((w) >> 4) & 0x0f0f0f0f) | ((((hw) >> s1) & 0x01010101u) << 4))
        dl = dot4(Q5_LO(q.x, qh.x), lo.x, dl); dh = dot4(Q5_HI(q.x, qh.x), hi.x, dh);
        dl = dot4(Q5_LO(q.y, qh.y), lo.y, dl); dh = dot4(Q5_HI(q.y, qh.y), hi.y, dh);
        dl = dot4(Q5_LO(q.z, qh.z), lo.z, dl); dh = dot4(Q5_HI(q.z, qh.z), hi.z, dh);
        dl = dot4(Q5_LO(q.w, qh.w), lo.w, dl); dh = dot4(Q5_HI(q.w, qh.w), hi.w, dh);
#undef Q5_LO
#undef Q5_HI
        isum = __mul24(sc_lo, dl) + __mul24(sc_hi, dh);
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 4 * c + h;
        msum = __mul24(m_lo, (int)bs[0]) + __mul24(m_hi, (int)bs[2]);
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum - (dmin * yd) * (float)msum;
    }
};

// ---- Q6_K : device row planes [ql nb*128][qh nb*64][scales nb*16][d nb*2] ---------------------
template <> struct Item<T_Q6_K> {
    uint4 hdr;
    static constexpr int EPP = 2048;
    uint4 ql, qh;
    int sc_lo, sc_hi;
    float d;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        const int nb = K >> 8;
        sb = pass * 8 + (lane >> 3);
        valid = sb < nb;
        if (valid) {
            const int v = lane & 7, n = v >> 2, w = v & 3;
            ql = ld16w(row + (size_t)sb * 128 + v * 16);
            qh = ld16w(row + (size_t)nb * 128 + (size_t)sb * 64 + n * 32 + (w & 1) * 16);
            const int8_t *sc = reinterpret_cast<const int8_t *>(row + (size_t)nb * 192 + (size_t)sb * 16 + 8 * n + w);
            hdr.x = (uint32_t)(int)sc[0];
            hdr.y = (uint32_t)(int)sc[4];
            hdr.z = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (!valid) return;
        sc_lo = (int)hdr.x; sc_hi = (int)hdr.y;
        d = h2f((uint16_t)hdr.z);
        const int w = lane & 3;
        const int s0 = 2 * (w >> 1), s1 = s0 + 4;
        uint4 a, b;
        a.x = (ql.x & 0x0f0f0f0f) | (((qh.x >> s0) & 0x03030303u) << 4); b.x = ((ql.x >> 4) & 0x0f0f0f0f) | (((qh.x >> s1) & 0x03030303u) << 4);
        a.y = (ql.y & 0x0f0f0f0f) | (((qh.y >> s0) & 0x03030303u) << 4); b.y = ((ql.y >> 4) & 0x0f0f0f0f) | (((qh.y >> s1) & 0x03030303u) << 4);
        a.z = (ql.z & 0x0f0f0f0f) | (((qh.z >> s0) & 0x03030303u) << 4); b.z = ((ql.z >> 4) & 0x0f0f0f0f) | (((qh.z >> s1) & 0x03030303u) << 4);
        a.w = (ql.w & 0x0f0f0f0f) | (((qh.w >> s0) & 0x03030303u) << 4); b.w = ((ql.w >> 4) & 0x0f0f0f0f) | (((qh.w >> s1) & 0x03030303u) << 4);
        ql = a; qh = b;   // now: ql = low-plane 6-bit codes, qh = high-plane 6-bit codes (0..63)
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, n = v >> 2, w = v & 3;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 128 * n + 16 * w;
        const uint4 lo = ld16(a), hi = ld16(a + 64);
        int dl = 0, dh = 0;
        dl = dot4(ql.x, lo.x, dl); dh = dot4(qh.x, hi.x, dh);
        dl = dot4(ql.y, lo.y, dl); dh = dot4(qh.y, hi.y, dh);
        dl = dot4(ql.z, lo.z, dl); dh = dot4(qh.z, hi.z, dh);
        dl = dot4(ql.w, lo.w, dl); dh = dot4(qh.w, hi.w, dh);
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 8 * n + w;
        isum = __mul24(sc_lo, dl - 32 * (int)bs[0]) + __mul24(sc_hi, dh - 32 * (int)bs[4]);
        msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum;
    }
};

// ---- Q2_K : device row planes [qs nb*64][scales nb*16][d, dmin nb*4].  Lane v of a super-block: half n = v / 4, bytes l0 = 8 (v % 4) .. + 7 of that
// half's 32 code bytes; their four 2-bit fields are elements 128 n + 32 j + l0 .. + 7 (j = 0..3), which sit in sub-block is = 8 n + 2 j + (v % 4) / 2.
// ggml_vec_dot_q2_K_q8_K: isum = sum_is (scales[is] & 15) * sum16(q2 * q8), summs = sum_is (scales[is] >> 4) * bsums[is] ----------------
__device__ __forceinline__ uint2 ld8w(const void *p) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t *>(p));
    return make_uint2(v.x, v.y);
}
template <> struct Item<T_Q2_K> {
    uint4 hdr;                        // the 16 scale / min bytes of the super-block
    static constexpr int EPP = 2048;
    uint2 q;
    uint32_t dm;
    float d, dmin;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        const int nb = K >> 8;
        sb = pass * 8 + (lane >> 3);
        valid = sb < nb;
        if (valid) {
            const int v = lane & 7;
            q = ld8w(row + (size_t)sb * 64 + 32 * (v >> 2) + 8 * (v & 3));
            hdr = ld16w(row + (size_t)nb * 64 + (size_t)sb * 16);
            dm = *reinterpret_cast<const uint32_t *>(row + (size_t)nb * 80 + (size_t)sb * 4);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (!valid) return;
        d = h2f((uint16_t)(dm & 0xffff));
        dmin = h2f((uint16_t)(dm >> 16));
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, n = v >> 2, kq = v & 3, h = kq >> 1;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 128 * n + 8 * kq;
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 8 * n + h;
        const uint32_t w01 = n ? hdr.z : hdr.x, w23 = n ? hdr.w : hdr.y;      // scale bytes 8 n .. 8 n + 3 / 8 n + 4 .. 8 n + 7
        int is_ = 0, ms_ = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint2 aa = *reinterpret_cast<const uint2 *>(a + 32 * j);
            int s = dot4((q.x >> (2 * j)) & 0x03030303u, aa.x, 0);
            s = dot4((q.y >> (2 * j)) & 0x03030303u, aa.y, s);
            const uint32_t scb = ((j < 2 ? w01 : w23) >> (8 * ((2 * j + h) & 3))) & 0xffu;
            is_ += __mul24((int)(scb & 0xf), s);
            if ((v & 1) == 0) ms_ += __mul24((int)(scb >> 4), (int)bs[2 * j]);   // the two lanes of a sub-block: the even one carries its minimum
        }
        isum = is_; msum = ms_;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (yd * d) * (float)isum - (yd * dmin) * (float)msum;
    }
};

// ---- Q3_K : device row planes [hmask nb*32][qs nb*64][scales nb*12][d nb*2].  Same lane -> element map as Q2_K; the third bit of element (n, j, l) is
// bit 4 n + j of hmask[l] and the value is code - 4 where it is CLEAR; sub-block scale = 6-bit scale - 32 (ggml_vec_dot_q3_K_q8_K) ----------
template <> struct Item<T_Q3_K> {
    uint4 hdr;                        // x, y, z: the 12 packed scale bytes; w: d (f16)
    static constexpr int EPP = 2048;
    uint2 q, hm;
    int sc[4];                        // (scale - 32) of this lane's four sub-blocks
    float d;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        const int nb = K >> 8;
        sb = pass * 8 + (lane >> 3);
        valid = sb < nb;
        if (valid) {
            const int v = lane & 7;
            hm = ld8w(row + (size_t)sb * 32 + 8 * (v & 3));
            q = ld8w(row + (size_t)nb * 32 + (size_t)sb * 64 + 32 * (v >> 2) + 8 * (v & 3));
            const uint32_t *s = reinterpret_cast<const uint32_t *>(row + (size_t)nb * 96 + (size_t)sb * 12);
            hdr.x = s[0]; hdr.y = s[1]; hdr.z = s[2];
            hdr.w = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 108 + (size_t)sb * 2);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (!valid) return;
        d = h2f((uint16_t)hdr.w);
        // the kmask shuffle of the CPU code: 16 six-bit scales as 16 bytes
        const uint32_t k1 = 0x03030303u, k2 = 0x0f0f0f0fu, tmp = hdr.z;
        const uint32_t a0 = (hdr.x & k2) | (((tmp >> 0) & k1) << 4), a1 = (hdr.y & k2) | (((tmp >> 2) & k1) << 4);
        const uint32_t a2 = ((hdr.x >> 4) & k2) | (((tmp >> 4) & k1) << 4), a3 = ((hdr.y >> 4) & k2) | (((tmp >> 6) & k1) << 4);
        const int v = lane & 7, n = v >> 2, h = (v & 3) >> 1;
        const uint32_t w01 = n ? a2 : a0, w23 = n ? a3 : a1;
#pragma unroll
        for (int j = 0; j < 4; j++) sc[j] = (int)(((j < 2 ? w01 : w23) >> (8 * ((2 * j + h) & 3))) & 0xffu) - 32;
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, n = v >> 2, kq = v & 3;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 128 * n + 8 * kq;
        int is_ = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint2 aa = *reinterpret_cast<const uint2 *>(a + 32 * j);
            int s = dot4((q.x >> (2 * j)) & 0x03030303u, aa.x, 0);
            s = dot4((q.y >> (2 * j)) & 0x03030303u, aa.y, s);
            // - 4 for every element whose high bit is clear: 4 * sum of the activations under the complemented mask
            int m = dot4(((hm.x >> (4 * n + j)) & 0x01010101u) ^ 0x01010101u, aa.x, 0);
            m = dot4(((hm.y >> (4 * n + j)) & 0x01010101u) ^ 0x01010101u, aa.y, m);
            is_ += __mul24(sc[j], s - 4 * m);
        }
        isum = is_; msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum;
    }
};

// ---- Q8_0 : device row planes [qs K][d K/32 f16] -------------------------------------------------
// 32-element formats with a nibble field (Q4_0, Q5_0, IQ4_NL; activation Q8_0): two lanes per block as for Q8_0, lane half h = (e >> 4) & 1 takes
// nibble h of the block's 16 code bytes (element j in the low nibbles, j + 16 in the high ones).  Integer sums exact:
//   Q4_0:   sum (nib - 8) a      = dot(nib, a) - 8 sum a
//   Q5_0:   sum (nib | b << 4 - 16) a = dot(nib + 16 b, a) - 16 sum a
//   IQ4_NL: sum level[nib] a     (levels looked up with v_perm from the 16-byte code book)
template <int TYPE> struct ItemNib32 {
    uint4 hdr;
    static constexpr int EPP = 1024;
    uint4 q;          // the block's 16 code bytes
    uint32_t qh;
    float d;
    int e;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        e = pass * 1024 + lane * 16;
        valid = e < K;
        if (valid) {
            const int b = e >> 5;
            const size_t half = (size_t)K >> 1;
            q = ld16w(row + (size_t)b * 16);
            if (TYPE == T_Q5_0) {
                qh = *reinterpret_cast<const uint32_t *>(row + half + (size_t)b * 4);
                hdr.x = *reinterpret_cast<const uint16_t *>(row + half + (size_t)(K >> 5) * 4 + (size_t)b * 2);
            } else {
                hdr.x = *reinterpret_cast<const uint16_t *>(row + half + (size_t)b * 2);
            }
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (valid) d = h2f((uint16_t)hdr.x);
    }
    static __device__ __forceinline__ uint32_t levels(uint32_t idx) {      // four IQ4_NL levels from four nibble indices (one per byte)
        const uint32_t lo = __builtin_amdgcn_perm(0xf6eaddcfu, 0xbfad9881u, idx & 0x07070707u);
        const uint32_t hi = __builtin_amdgcn_perm(0x71594535u, 0x26190d01u, idx & 0x07070707u);
        const uint32_t m = ((idx >> 3) & 0x01010101u) * 0xffu;
        return (hi & m) | (lo & ~m);
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        int s = 0;
        if (valid) {
            const int h = (e >> 4) & 1;
            const uint4 a = ld16(A.qs0 + (size_t)t * A.K + e);
            const uint32_t qq[4] = {q.x, q.y, q.z, q.w}, aa[4] = {a.x, a.y, a.z, a.w};
            int asum = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t v = (qq[w] >> (4 * h)) & 0x0f0f0f0fu;
                if (TYPE == T_IQ4_NL) { s = dot4(levels(v), aa[w], s); continue; }
                if (TYPE == T_Q5_0) {
                    const uint32_t bits = (qh >> (16 * h + 4 * w)) & 0xfu;              // fifth bits of these four elements
                    v |= ((bits * 0x00204081u) & 0x01010101u) << 4;
                }
                s = dot4(v, aa[w], s);                                                   // codes are 0 .. 31: the same as unsigned bytes
                asum = dot4(0x01010101u, aa[w], asum);
            }
            if (TYPE == T_Q4_0) s -= 8 * asum;
            if (TYPE == T_Q5_0) s -= 16 * asum;
        }
        isum = s + __shfl_xor(s, 1, 64);   // both halves of the 32-element block
        msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        int isum, msum;
        ints(A, t, lane, isum, msum);     // shuffle executed by all lanes
        if (!valid || (lane & 1)) return 0.0f;
        const float da = h2f(A.d0[(size_t)t * (A.K >> 5) + (e >> 5)]);
        if (TYPE == T_Q4_0) return ((float)isum * d) * da;          // ggml_vec_dot_q4_0_q8_0: sumi * d_x * d_y, left to right
        return (d * da) * (float)isum;                              // q5_0 / iq4_nl: (d_x * d_y) * sumi
    }
};
template <> struct Item<T_Q4_0> : ItemNib32<T_Q4_0> {};
template <> struct Item<T_Q5_0> : ItemNib32<T_Q5_0> {};
template <> struct Item<T_IQ4_NL> : ItemNib32<T_IQ4_NL> {};

template <> struct Item<T_Q8_0> {
    uint4 hdr;
    static constexpr int EPP = 1024;
    uint4 q;
    float d;
    int e;       // element offset of this lane's 16 codes
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane) {
        e = pass * 1024 + lane * 16;
        valid = e < K;
        if (valid) {
            q = ld16w(row + e);
            hdr.x = *reinterpret_cast<const uint16_t *>(row + K + (e >> 5) * 2);
        }
    }
    __device__ __forceinline__ void prep(int lane) {
        if (valid) d = h2f((uint16_t)hdr.x);
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        int s = 0;
        if (valid) {
            const uint4 a = ld16(A.qs0 + (size_t)t * A.K + e);
            s = dot4(q.x, a.x, s); s = dot4(q.y, a.y, s); s = dot4(q.z, a.z, s); s = dot4(q.w, a.w, s);
        }
        isum = s + __shfl_xor(s, 1, 64);   // both halves of the 32-element block
        msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        int isum, msum;
        ints(A, t, lane, isum, msum);     // shuffle executed by all lanes
        if (!valid || (lane & 1)) return 0.0f;
        const float da = h2f(A.d0[(size_t)t * (A.K >> 5) + (e >> 5)]);
        return (float)isum * (d * da);
    }
};

// ------------------------------------------------------------------------------------------
// Activation staging.  mode 0: copy pre-quantised planes of NT tokens into LDS.
// mode 1 (NT == 1): fused RMSNorm * weight + quantise of a.nx[K] straight into LDS (same arithmetic, same order as
// norm_quant_kernel in act.hip, so the result is bit-identical to the unfused path).  mode 2: quantise a.nx only.
template <int NT, int BS>
__device__ __forceinline__ ActLds stage_act(const MMVQArgs &a, uint8_t *smem) {
    ActLds A;
    const int K = a.K, tid = threadIdx.x, nthr = blockDim.x;
    A.K = K;
    uint8_t *p = smem;
    A.qs = nullptr; A.d = nullptr; A.bs = nullptr; A.qs0 = nullptr; A.d0 = nullptr;
    int8_t *qs = nullptr, *qs0 = nullptr;
    float *d = nullptr;
    int16_t *bs = nullptr;
    uint16_t *d0 = nullptr;
    if (a.need_q8k) {
        qs = reinterpret_cast<int8_t *>(p);           p += (size_t)NT * K;
        d = reinterpret_cast<float *>(p);             p += (((size_t)NT * (K >> 8) * 4) + 15) & ~15;
        bs = reinterpret_cast<int16_t *>(p);          p += (((size_t)NT * (K >> 4) * 2) + 15) & ~15;
        A.qs = qs; A.d = d; A.bs = bs;
    }
    if (a.need_q80) {
        qs0 = reinterpret_cast<int8_t *>(p);          p += (size_t)NT * K;
        d0 = reinterpret_cast<uint16_t *>(p);         p += (((size_t)NT * (K >> 5) * 2) + 15) & ~15;
        A.qs0 = qs0; A.d0 = d0;
    }
    if (a.fuse_mode == 0 || NT != 1) {
        if (a.need_q8k) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.aq);
            for (int i = tid; i < NT * K / 16; i += nthr) reinterpret_cast<uint4 *>(qs)[i] = src[i];
            for (int i = tid; i < NT * (K >> 8); i += nthr) d[i] = a.ad[i];
            const uint32_t *bsrc = reinterpret_cast<const uint32_t *>(a.abs);
            for (int i = tid; i < NT * (K >> 5); i += nthr) reinterpret_cast<uint32_t *>(bs)[i] = bsrc[i];
        }
        if (a.need_q80) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.aq0);
            for (int i = tid; i < NT * K / 16; i += nthr) reinterpret_cast<uint4 *>(qs0)[i] = src[i];
            for (int i = tid; i < NT * (K >> 5); i += nthr) d0[i] = a.ad0[i];
        }
    } else {
        double *red = reinterpret_cast<double *>(smem + a.red_off);   // 16 doubles after the planes
        const int lane = tid & 63, wave = tid >> 6, nwv = nthr >> 6;
        const int sweep = nthr * 4;                      // elements covered by one sweep of the workgroup
        constexpr int MAXJ = 8192 / (BS * 4);            // K <= 8192 (host-checked)
        float4 xv[MAXJ];
        const int nj = K / sweep;                        // K % sweep == 0 host-checked
#pragma unroll
        for (int j = 0; j < MAXJ; j++)
            if (j < nj) xv[j] = *reinterpret_cast<const float4 *>(a.nx + j * sweep + tid * 4);
        float scale = 1.0f;
        if (a.fuse_mode == 1) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < MAXJ; j++)
                if (j < nj) {
                    const float4 v = xv[j];
                    s += (double)(v.x * v.x); s += (double)(v.y * v.y); s += (double)(v.z * v.z); s += (double)(v.w * v.w);
                }
            s = wave_sum(s);
            if (lane == 0) red[wave] = s;
            __syncthreads();
            double tot = 0.0;
            for (int w = 0; w < nwv; w++) tot += red[w];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + a.neps);
        }
#pragma unroll
        for (int j = 0; j < MAXJ; j++) {
            if (j >= nj) continue;
            const int b = wave + nwv * j;                 // 256-block handled by this wave
            const int e0 = b * 256 + lane * 4;
            float4 v = xv[j];
            if (a.fuse_mode == 1) {
                const float4 ww = *reinterpret_cast<const float4 *>(a.nw + e0);
                v.x = (v.x * scale) * ww.x; v.y = (v.y * scale) * ww.y; v.z = (v.z * scale) * ww.z; v.w = (v.w * scale) * ww.w;
            }
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (a.need_q8k) {
                uint32_t packed; int bsum; float dq;
                wave_quant_q8k(vv, lane, packed, bsum, dq);
                *reinterpret_cast<uint32_t *>(qs + e0) = packed;
                if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum;
                if (lane == 0) d[b] = dq;
            }
            if (a.need_q80) {
                uint32_t packed; float dd;
                wave_quant_q80(vv, packed, dd);
                *reinterpret_cast<uint32_t *>(qs0 + e0) = packed;
                if ((lane & 7) == 0) d0[b * 8 + (lane >> 3)] = f2h(dd);
            }
        }
    }
    __syncthreads();
    return A;
}

// One segment (block-uniform TYPE), one "unit" per wave and no grid-stride loop: a wave owns a PAIR of weight rows
// (swiglu: one ffn_gate row + the matching ffn_up row) and a subset of the 2-pass chunks of those rows; nck waves of
// the workgroup share a pair when the rows are long (K > 4096) and combine their partial sums through LDS in a fixed
// order.  The weight loads are issued first, then the activations are staged (barrier), then the integer dots run:
// every byte of the matrix is requested within the first microsecond of the launch.
template <int TYPE, int NT, int BS>
__device__ __forceinline__ void run_segment(const MMVQArgs &a, const MMVQSeg &sg, uint8_t *smem, int blk_in_seg) {
    using It = Item<TYPE>;
    constexpr int NW = BS / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.K;
    const bool swiglu = a.epi == EPI_SWIGLU;
    // MoE: the expert is picked on the device (no host sync): weight base = W + expert_sel[0] * expert_stride
    const uint8_t *W0 = sg.W + (sg.expert_sel ? (size_t)sg.expert_sel[0] * sg.expert_stride : 0);
    const MMVQSeg &ug = a.seg[1];
    const uint8_t *W1 = swiglu ? ug.W + (ug.expert_sel ? (size_t)ug.expert_sel[0] * ug.expert_stride : 0) : nullptr;
    const size_t rb0 = sg.row_bytes, rb1 = swiglu ? ug.row_bytes : sg.row_bytes;
    const int npass = (K + It::EPP - 1) / It::EPP;
    const int nchunk = (npass + 1) >> 1;
    const int nck = a.nck;                       // waves sharing one pair (power of two, <= NW, host-chosen)
    const int cw = wave & (nck - 1);             // first chunk of this wave
    const int pair = blk_in_seg * (NW / nck) + wave / nck;
    const int npairs = swiglu ? sg.n_rows : (sg.n_rows + 1) >> 1;
    const bool have = pair < npairs;
    const uint8_t *ra = nullptr, *rbp = nullptr;
    bool hasb = false;
    if (have) {
        if (swiglu) { ra = W0 + (size_t)pair * rb0; rbp = W1 + (size_t)pair * rb1; hasb = true; }
        else { ra = W0 + (size_t)(2 * pair) * rb0; rbp = ra + rb0; hasb = 2 * pair + 1 < sg.n_rows; }
    }
    It i0, i1, i2, i3;   // (row a, pass), (row b, pass), (row a, pass+1), (row b, pass+1)
    auto load_chunk = [&](int c) {
        const int ps = 2 * c;
        i0.load(ra, K, ps, lane);
        if (hasb) i1.load(rbp, K, ps, lane); else i1.valid = false;
        if (ps + 1 < npass) {
            i2.load(ra, K, ps + 1, lane);
            if (hasb) i3.load(rbp, K, ps + 1, lane); else i3.valid = false;
        } else {
            i2.valid = false; i3.valid = false;
        }
    };
    bool active = have && cw < nchunk;
    if (active) load_chunk(cw);
    const ActLds A = stage_act<NT, BS>(a, smem); // contains the workgroup barrier: reached by every wave

    float acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { acc[0][t] = 0.0f; acc[1][t] = 0.0f; }
    for (int c = cw; active; ) {
        // decode + dot one item at a time so only one item's unpacked scales are live
        i0.prep(lane);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[0][t] += i0.dot(A, t, lane);
        i1.prep(lane);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[1][t] += i1.dot(A, t, lane);
        i2.prep(lane);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[0][t] += i2.dot(A, t, lane);
        i3.prep(lane);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[1][t] += i3.dot(A, t, lane);
        c += nck;
        active = c < nchunk;
        if (active) load_chunk(c);
    }
    float v0[NT], v1[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { v0[t] = wave_sum(acc[0][t]); v1[t] = wave_sum(acc[1][t]); }
    if (nck > 1) {   // combine the chunk partials of the waves that share this pair (fixed order -> deterministic)
        float *red = reinterpret_cast<float *>(smem + a.red_off);
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < NT; t++) { red[(wave * 2 + 0) * NT + t] = v0[t]; red[(wave * 2 + 1) * NT + t] = v1[t]; }
        }
        __syncthreads();
        if (cw != 0) return;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float s0 = 0.0f, s1 = 0.0f;
            for (int j = 0; j < nck; j++) { s0 += red[((wave + j) * 2 + 0) * NT + t]; s1 += red[((wave + j) * 2 + 1) * NT + t]; }
            v0[t] = s0; v1[t] = s1;
        }
    }
    if (lane == 0 && have) {
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (swiglu) {
                sg.out[(size_t)t * sg.ld_out + pair] = (v0[t] / (1.0f + expf(-v0[t]))) * v1[t];
            } else {
                const int row0 = 2 * pair;
                const size_t o = (size_t)t * sg.ld_out + row0;
                if (a.epi == EPI_ADD) {
                    sg.out[o] = sg.resid[o] + v0[t];
                    if (hasb) sg.out[o + 1] = sg.resid[o + 1] + v1[t];
                } else {
                    sg.out[o] = v0[t];
                    if (hasb) sg.out[o + 1] = v1[t];
                }
            }
        }
    }
}

// Blocks [seg_block0[s], seg_block0[s+1]) work on segment s, so the weight TYPE is block-uniform and
// each type's code path keeps its own register footprint.
template <int NT, int BS>
__global__ __launch_bounds__(BS) void mmvq_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    const int bis = (int)blockIdx.x - a.seg_block0[s];
    switch (a.seg[s].type) {
        case T_Q4_K: run_segment<T_Q4_K, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q5_K: run_segment<T_Q5_K, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q6_K: run_segment<T_Q6_K, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q2_K: run_segment<T_Q2_K, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q3_K: run_segment<T_Q3_K, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q8_0: run_segment<T_Q8_0, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q4_0: run_segment<T_Q4_0, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_Q5_0: run_segment<T_Q5_0, NT, BS>(a, a.seg[s], smem, bis); break;
        case T_IQ4_NL: run_segment<T_IQ4_NL, NT, BS>(a, a.seg[s], smem, bis); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------
// Prompt-processing variant: NT tokens per launch (8/16), rows owned by a wave for the whole kernel, the K axis walked
// in chunks of 4096 elements whose Q8_K activations (NT x 4096 codes) are re-staged in LDS per chunk.  Each weight byte
// is read once per NT tokens.  Integer path identical to the single-token kernels (exact partial sums).
// (Placeholder for the MFMA mul_mat_q of DESIGN.md section 7: this one is v_dot4 / VALU bound.)
template <int TYPE, int NT>
__device__ __forceinline__ void run_tiled(const MMVQArgs &a, const MMVQSeg &sg, uint8_t *smem, int blk_in_seg) {
    using It = Item<TYPE>;
    constexpr int KC = 4096;
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = a.K;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const uint8_t *W0 = sg.W;
    const MMVQSeg &ug = a.seg[1];
    const uint8_t *W1 = swiglu ? ug.W : nullptr;
    const size_t rb0 = sg.row_bytes, rb1 = swiglu ? ug.row_bytes : sg.row_bytes;
    const int npass = (K + It::EPP - 1) / It::EPP;
    const int ppc = KC / It::EPP;                      // passes per chunk (2 for K-quants, 4 for Q8_0)
    const int nchunk = (K + KC - 1) / KC;
    const int pair = blk_in_seg * 4 + wave;
    const int npairs = swiglu ? sg.n_rows : (sg.n_rows + 1) >> 1;
    const bool have = pair < npairs;
    const uint8_t *ra = nullptr, *rbp = nullptr;
    bool hasb = false;
    if (have) {
        if (swiglu) { ra = W0 + (size_t)pair * rb0; rbp = W1 + (size_t)pair * rb1; hasb = true; }
        else { ra = W0 + (size_t)(2 * pair) * rb0; rbp = ra + rb0; hasb = 2 * pair + 1 < sg.n_rows; }
    }
    int8_t *qs = reinterpret_cast<int8_t *>(smem);
    float *dd = reinterpret_cast<float *>(smem + (size_t)NT * KC);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + (size_t)NT * KC + NT * (KC >> 8) * 4);
    uint16_t *d0 = reinterpret_cast<uint16_t *>(dd);   // Q8_0 activations reuse the scale area (KC/32 halves per token fit: 256 B <= 64+512)
    float acc[2][NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { acc[0][t] = 0.0f; acc[1][t] = 0.0f; }
    for (int c = 0; c < nchunk; c++) {
        const int k0 = c * KC, kc = (K - k0) < KC ? (K - k0) : KC;
        __syncthreads();                                   // everyone is done with the previous chunk's LDS image
        for (int pp = 0; pp < ppc; pp += 2) {
            const int ps = c * ppc + pp;
            It i0, i1, i2, i3;
            if (have && ps < npass) { i0.load(ra, K, ps, lane); if (hasb) i1.load(rbp, K, ps, lane); else i1.valid = false; }
            else { i0.valid = false; i1.valid = false; }
            if (have && ps + 1 < npass) { i2.load(ra, K, ps + 1, lane); if (hasb) i3.load(rbp, K, ps + 1, lane); else i3.valid = false; }
            else { i2.valid = false; i3.valid = false; }
            if (pp == 0) {                                 // stage this chunk's activations (all waves), loads already in flight
                if (act_is_q80(TYPE)) {
                    for (int i = tid; i < NT * (kc >> 4); i += 256) {
                        const int t = i / (kc >> 4), j = i - t * (kc >> 4);
                        reinterpret_cast<uint4 *>(qs + (size_t)t * kc)[j] = reinterpret_cast<const uint4 *>(a.aq0 + (size_t)t * K + k0)[j];
                    }
                    for (int i = tid; i < NT * (kc >> 5); i += 256) {
                        const int t = i / (kc >> 5), j = i - t * (kc >> 5);
                        d0[t * (kc >> 5) + j] = a.ad0[(size_t)t * (K >> 5) + (k0 >> 5) + j];
                    }
                } else {
                    for (int i = tid; i < NT * (kc >> 4); i += 256) {
                        const int t = i / (kc >> 4), j = i - t * (kc >> 4);
                        reinterpret_cast<uint4 *>(qs + (size_t)t * kc)[j] = reinterpret_cast<const uint4 *>(a.aq + (size_t)t * K + k0)[j];
                    }
                    for (int i = tid; i < NT * (kc >> 8); i += 256) {
                        const int t = i / (kc >> 8), j = i - t * (kc >> 8);
                        dd[t * (kc >> 8) + j] = a.ad[(size_t)t * (K >> 8) + (k0 >> 8) + j];
                    }
                    for (int i = tid; i < NT * (kc >> 4); i += 256) {
                        const int t = i / (kc >> 4), j = i - t * (kc >> 4);
                        bs[t * (kc >> 4) + j] = a.abs[(size_t)t * (K >> 4) + (k0 >> 4) + j];
                    }
                }
                __syncthreads();
            }
            ActLds A;
            A.K = kc;
            A.qs = qs - k0; A.d = dd - (k0 >> 8); A.bs = bs - (k0 >> 4); A.qs0 = qs - k0; A.d0 = d0 - (k0 >> 5);
            i0.prep(lane);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[0][t] += i0.dot(A, t, lane);
            i1.prep(lane);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[1][t] += i1.dot(A, t, lane);
            i2.prep(lane);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[0][t] += i2.dot(A, t, lane);
            i3.prep(lane);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[1][t] += i3.dot(A, t, lane);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float v0 = wave_sum(acc[0][t]), v1 = wave_sum(acc[1][t]);
        if (lane == 0 && have) {
            if (swiglu) {
                sg.out[(size_t)t * sg.ld_out + pair] = (v0 / (1.0f + expf(-v0))) * v1;
            } else {
                const int row0 = 2 * pair;
                const size_t o = (size_t)t * sg.ld_out + row0;
                if (a.epi == EPI_ADD) {
                    sg.out[o] = sg.resid[o] + v0;
                    if (hasb) sg.out[o + 1] = sg.resid[o + 1] + v1;
                } else {
                    sg.out[o] = v0;
                    if (hasb) sg.out[o + 1] = v1;
                }
            }
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void mmvq_tiled_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    const int bis = (int)blockIdx.x - a.seg_block0[s];
    switch (a.seg[s].type) {
        case T_Q4_K: run_tiled<T_Q4_K, NT>(a, a.seg[s], smem, bis); break;
        case T_Q5_K: run_tiled<T_Q5_K, NT>(a, a.seg[s], smem, bis); break;
        case T_Q6_K: run_tiled<T_Q6_K, NT>(a, a.seg[s], smem, bis); break;
        case T_Q2_K: run_tiled<T_Q2_K, NT>(a, a.seg[s], smem, bis); break;
        case T_Q3_K: run_tiled<T_Q3_K, NT>(a, a.seg[s], smem, bis); break;
        case T_Q8_0: run_tiled<T_Q8_0, NT>(a, a.seg[s], smem, bis); break;
        case T_Q4_0: run_tiled<T_Q4_0, NT>(a, a.seg[s], smem, bis); break;
        case T_Q5_0: run_tiled<T_Q5_0, NT>(a, a.seg[s], smem, bis); break;
        case T_IQ4_NL: run_tiled<T_IQ4_NL, NT>(a, a.seg[s], smem, bis); break;
        default: break;
    }
}

static hipError_t launch_tiled(MMVQArgs a, hipStream_t st) {
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return hipErrorInvalidValue;
    const int n_work_seg = a.epi == EPI_SWIGLU ? 1 : a.n_seg;
    a.seg_block0[0] = 0;
    for (int s = 0; s < n_work_seg; s++) {
        const int pairs = a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
        a.seg_block0[s + 1] = a.seg_block0[s] + (pairs + 3) / 4;
    }
    for (int s = n_work_seg; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n_work_seg];
    const int blocks = a.seg_block0[n_work_seg];
    if (a.epi == EPI_SWIGLU) a.n_seg = 1;
    const size_t lds = (size_t)a.T * 4096 + (size_t)a.T * 16 * 4 + (size_t)a.T * 256 * 2 + 64;
    if (a.T == 16) hipLaunchKernelGGL(mmvq_tiled_kernel<16>, dim3(blocks), dim3(256), lds, st, a);
    else if (a.T == 8) hipLaunchKernelGGL(mmvq_tiled_kernel<8>, dim3(blocks), dim3(256), lds, st, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

size_t mmvq_lds_bytes(const MMVQArgs &a, int NT) {
    size_t b = 0;
    const size_t K = (size_t)a.K;
    if (a.need_q8k) b += NT * K + ((NT * (K >> 8) * 4 + 15) & ~(size_t)15) + ((NT * (K >> 4) * 2 + 15) & ~(size_t)15);
    if (a.need_q80) b += NT * K + ((NT * (K >> 5) * 2 + 15) & ~(size_t)15);
    return b;
}

static int g_num_cu = 256;
void set_num_cu(int n) { if (n > 0) g_num_cu = n; }
int num_cu() { return g_num_cu; }

template <int NT>
static void launch_nt(const MMVQArgs &a, int blocks, int bs, size_t lds, hipStream_t st) {
    if (bs == 1024) hipLaunchKernelGGL((mmvq_kernel<NT, 1024>), dim3(blocks), dim3(1024), lds, st, a);
    else hipLaunchKernelGGL((mmvq_kernel<NT, 256>), dim3(blocks), dim3(256), lds, st, a);
}

// MI355_MMVQ_STREAM=0 / mi355_debug_set_option("mmvq_stream", 0): single-token mat-vecs take the register-ring kernel
// (mmvq_fast.hip) instead of the weight-stream kernel; the two are bit-identical (tests/test_gpu_model.py)
static bool g_stream_on = !(getenv("MI355_MMVQ_STREAM") && getenv("MI355_MMVQ_STREAM")[0] == '0');
void mmvq_set_stream(bool on) { g_stream_on = on; }

// host launcher: a.T tokens (1, 2 or 4 per launch; larger T is chunked by the caller).
// EPI_SWIGLU: seg[0] = ffn_gate, seg[1] = ffn_up (same type), out = silu(gate) * up into seg[0].out.
hipError_t launch_mmvq(MMVQArgs a, hipStream_t st) {
    static const bool fast_init = (mmvq_fast_set_threads(getenv("MI355_MMVQ_NT") ? atoi(getenv("MI355_MMVQ_NT")) : 0), true);
    (void)fast_init;
    static const bool fast_on = !(getenv("MI355_MMVQ_FAST") && getenv("MI355_MMVQ_FAST")[0] == '0');   // diagnosis switch
    // single-token steps: the weight-stream kernel (LDS-DMA loader / consumer waves) where it has a form, else the register ring
    if (fast_on && g_stream_on && mmvq_stream_applicable(a)) return launch_mmvq_stream(a, st);
    if (a.out_host) {
        // only the weight stream stores the second copy itself: any other kernel is followed by a copy of the finished row (same contract, one node more)
        float *oh = a.out_host;
        a.out_host = nullptr;
        if (a.n_seg != 1 || a.T != 1) return hipErrorInvalidValue;
        const hipError_t e = launch_mmvq(a, st);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(oh, a.seg[0].out, (size_t)a.seg[0].n_rows * 4, hipMemcpyDeviceToHost, st);
    }
    if (fast_on && mmvq_fast_applicable(a)) return launch_mmvq_fast(a, st);
    if ((a.T == 16 || a.T == 8) && a.fuse_mode == 0) {
        bool moe = false;
        for (int s = 0; s < a.n_seg; s++) moe |= a.seg[s].expert_sel != nullptr;
        if (!moe) return launch_tiled(a, st);
        return hipErrorInvalidValue;
    }
    a.need_q8k = 0; a.need_q80 = 0;
    for (int s = 0; s < a.n_seg; s++) {
        if (act_is_q80(a.seg[s].type)) a.need_q80 = 1; else a.need_q8k = 1;
    }
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return hipErrorInvalidValue;
    const int n_work_seg = a.epi == EPI_SWIGLU ? 1 : a.n_seg;
    // chunks of 2 passes per row; the pass size is the same for every K-quant (2048), 1024 for Q8_0
    int nchunk = 1;
    for (int s = 0; s < n_work_seg; s++) {
        const int epp = act_is_q80(a.seg[s].type) ? 1024 : 2048;
        const int npass = (a.K + epp - 1) / epp;
        nchunk = ((npass + 1) >> 1) > nchunk ? ((npass + 1) >> 1) : nchunk;
    }
    int total_pairs = 0;
    for (int s = 0; s < n_work_seg; s++) total_pairs += a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
    // waves sharing one pair: spread long rows over waves while the launch is small enough to need it
    int nck = 1;
    while (nck * 2 <= nchunk && nck < 4 && (long)total_pairs * nck < (long)g_num_cu * 16) nck *= 2;
    a.nck = nck;
    // big launches use 1024-thread workgroups (one activation staging per 16 waves)
    int bs = ((long)total_pairs * nck >= (long)g_num_cu * 32) ? 1024 : 256;
    if (a.fuse_mode != 0) {
        if (a.T != 1 || (a.K % (bs * 4)) != 0 || a.K / (bs * 4) > 8) {
            if (bs == 1024 && a.T == 1 && (a.K % 1024) == 0 && a.K / 1024 <= 8) bs = 256; else return hipErrorInvalidValue;
        }
    }
    const int nw = bs / 64;
    const int ppb = nw / nck;                    // pairs per workgroup
    a.seg_block0[0] = 0;
    for (int s = 0; s < n_work_seg; s++) {
        const int pairs = a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
        a.seg_block0[s + 1] = a.seg_block0[s] + (pairs + ppb - 1) / ppb;
    }
    for (int s = n_work_seg; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n_work_seg];
    const int blocks = a.seg_block0[n_work_seg];
    if (a.epi == EPI_SWIGLU) a.n_seg = 1;       // block->segment lookup sees only the gate segment; seg[1] read directly
    size_t lds = mmvq_lds_bytes(a, a.T);
    lds = (lds + 15) & ~(size_t)15;
    a.red_off = (int)lds;                        // f64 norm scratch (16 doubles) / f32 chunk partials share this tail
    lds += 16 * 8 + (size_t)nw * 2 * a.T * 4;
    switch (a.T) {
        case 1: launch_nt<1>(a, blocks, bs, lds, st); break;
        case 2: launch_nt<2>(a, blocks, bs, lds, st); break;
        case 4: launch_nt<4>(a, blocks, bs, lds, st); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Debug / parity kernel: integer partial sums per (token, row, block), same Item code as above.
template <int TYPE>
__global__ __launch_bounds__(256) void mmvq_ints_kernel(const MMVQArgs a, int32_t *isum_out, int32_t *msum_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ActLds A = stage_act<1, 256>(a, smem);
    using It = Item<TYPE>;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nw = gridDim.x * (blockDim.x >> 6);
    const int K = a.K;
    const MMVQSeg &sg = a.seg[0];
    const int nblk = act_is_q80(TYPE) ? (K >> 5) : (K >> 8);
    const int npass = (K + It::EPP - 1) / It::EPP;
    for (int r = gw; r < sg.n_rows; r += nw) {
        const uint8_t *row = sg.W + (size_t)r * sg.row_bytes;
        for (int p = 0; p < npass; p++) {
            It it;
            it.load(row, K, p, lane);
            it.prep(lane);
            int is = 0, ms = 0;
            if constexpr (act_is_q80(TYPE)) {
                it.ints(A, 0, lane, is, ms);
                if (it.valid && !(lane & 1)) { isum_out[(size_t)r * nblk + (it.e >> 5)] = is; msum_out[(size_t)r * nblk + (it.e >> 5)] = 0; }
            } else {
                if (it.valid) it.ints(A, 0, lane, is, ms);
                is += __shfl_xor(is, 1, 64); ms += __shfl_xor(ms, 1, 64);
                is += __shfl_xor(is, 2, 64); ms += __shfl_xor(ms, 2, 64);
                is += __shfl_xor(is, 4, 64); ms += __shfl_xor(ms, 4, 64);
                const int sb = p * 8 + (lane >> 3);
                if (sb < nblk && (lane & 7) == 0) { isum_out[(size_t)r * nblk + sb] = is; msum_out[(size_t)r * nblk + sb] = ms; }
            }
        }
    }
}

hipError_t launch_mmvq_ints(MMVQArgs a, int32_t *isum, int32_t *msum, hipStream_t st) {
    a.need_q8k = !act_is_q80(a.seg[0].type);
    a.need_q80 = act_is_q80(a.seg[0].type);
    const size_t lds = mmvq_lds_bytes(a, 1);
    const int blocks = 64;
    switch (a.seg[0].type) {
        case T_Q4_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q4_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q5_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q5_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q6_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q6_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q2_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q2_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q3_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q3_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q8_0: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q8_0>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q4_0: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q4_0>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q5_0: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q5_0>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_IQ4_NL: hipLaunchKernelGGL(mmvq_ints_kernel<T_IQ4_NL>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mi355
