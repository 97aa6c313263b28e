// mmvq_fast_dev.h — device side of the single-token quantised mat-vec (see mmvq_fast.hip for the design notes):
// the block decoders, the activation staging and run_fast, the persistent software-pipelined row-pair loop.  Shared by
// mmvq_fast.hip (one launch per mat-vec) and decode_mega.hip (every mat-vec of a decode step inside one launch).
#pragma once
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t ldw(const void *p) {   // weight stream: non-temporal
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
}
__device__ __forceinline__ u32x4_t lds16(const void *p) { return *reinterpret_cast<const u32x4_t *>(p); }
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// Two lane roles (round 4).  NARROW: 8 lanes per super-block, a lane takes one 16-byte piece of the codes (32 weights), a pass of the wave covers 8
// super-blocks = 2048 weights.  WIDE (Q4_K, Q5_K): 4 lanes per super-block, a lane takes one whole 64-weight chunk (32 bytes of qs: low nibbles = sub-block
// 2c, high nibbles = sub-block 2c + 1 - the unit ggml's own loops walk), a pass covers 16 super-blocks = 4096 weights.  The per-lane work that does not
// depend on the number of weights (d / dmin conversion, the 6-bit scale / min extraction, the integer -> f32 fold) is then paid once per 64 weights instead
// of once per 32: the decode was instruction-bound (rocprofv3 --pmc: the SIMDs of ffn_down / Q | K | V issue back to back; a build with 17 % fewer dot
// instructions ran the layer's mat-vecs 3.8 % faster), and the wide role issues ~37 % fewer instructions per weight.  Integer sums per sub-block are what
// they were; only the f32 association over the lanes changes (one lane's term now covers 64 weights).
template <int TYPE> constexpr bool role_wide() { return TYPE == T_Q4_K || TYPE == T_Q5_K; }
template <int TYPE> constexpr int role_sbp() { return role_wide<TYPE>() ? 16 : 8; }                       // super-blocks per pass of a wave
template <int TYPE, int KB> constexpr int role_passes() { return role_wide<TYPE>() ? (KB + 1) / 2 : KB; }  // passes for K <= KB * 2048

// activation slice one lane needs for one pass
struct ActSlice {
    u32x4_t lo, hi;      // narrow: 16 + 16 int8 codes; wide: codes 0..15 / 32..47 of the chunk
    u32x4_t lo1, hi1;    // wide only: codes 16..31 / 48..63
    float yd;          // Q8_K block scale
    int bs_lo, bs_hi;  // sums of the two 16-code groups (wide: of the two 32-code sub-blocks)
};

struct LaneRole {      // constants of this lane, computed once
    int sbl;           // super-block within the pass (narrow 0..7, wide 0..15)
    int v, c, h;       // piece (0..7), chunk (0..3), half (0..1)
    int sh;            // bit shift selecting this chunk's 16-bit scale pair
    bool hi_scales;    // c >= 2: 6-bit scales split across bytes
    int n, w;          // Q6_K: half (0..1) and 16-code column (0..3)
    // the 6-bit scale / min pair of this chunk without a branch (get_scale_min_k4 for sub-blocks 2c, 2c + 1): with y, z, w = the three scale words,
    //   sc = ((hi_scales ? w : y) >> sh & m1) | (y >> (sh + 2) & m2),   mn = ((hi_scales ? w : z) >> sh_mn & m1) | (z >> (sh + 2) & m2)
    uint32_t m1, m2; int sh_mn;
};
// sub-block scales and mins of chunk c from the 12 scale bytes (words y, z, w of the block header), two 6-bit values each in bits 0..5 / 8..13
__device__ __forceinline__ void k4_scales(uint32_t y, uint32_t z, uint32_t w, const LaneRole &L, uint32_t &sc, uint32_t &mn) {
    const uint32_t xs = L.hi_scales ? w : y, xm = L.hi_scales ? w : z;
    sc = ((xs >> L.sh) & L.m1) | ((y >> (L.sh + 2)) & L.m2);
    mn = ((xm >> L.sh_mn) & L.m1) | ((z >> (L.sh + 2)) & L.m2);
}

template <int TYPE> struct Raw;
// sum of the two int16 halves of a word (one v_dot2_i32_i16)
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sum2_i16(uint32_t w) {
    const i16x2_t one = {1, 1};
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2_t, w), one, 0, false);
}

// ---------------------------------------------------------------- Q4_K (wide role: lane = (super-block, chunk c); qs bytes 32 c .. 32 c + 31)
template <> struct Raw<T_Q4_K> {
    u32x4_t hdr, q, q1;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const uint8_t *b = row + (size_t)sb * 144;
        hdr = ldw(b);
        q = ldw(b + 16 + L.c * 32);
        q1 = ldw(b + 32 + L.c * 32);
    }
    __device__ __forceinline__ float probe() const { return (float)(hdr.x ^ hdr.y ^ hdr.z ^ hdr.w ^ q.x ^ q.y ^ q.z ^ q.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w); }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
        uint32_t sc, mn;
        k4_scales(hdr.y, hdr.z, hdr.w, L, sc, mn);
        int dl = 0, dh = 0;
        dl = dot4(q.x & 0x0f0f0f0f, A.lo.x, dl); dh = dot4((q.x >> 4) & 0x0f0f0f0f, A.hi.x, dh);
        dl = dot4(q.y & 0x0f0f0f0f, A.lo.y, dl); dh = dot4((q.y >> 4) & 0x0f0f0f0f, A.hi.y, dh);
        dl = dot4(q.z & 0x0f0f0f0f, A.lo.z, dl); dh = dot4((q.z >> 4) & 0x0f0f0f0f, A.hi.z, dh);
        dl = dot4(q.w & 0x0f0f0f0f, A.lo.w, dl); dh = dot4((q.w >> 4) & 0x0f0f0f0f, A.hi.w, dh);
        dl = dot4(q1.x & 0x0f0f0f0f, A.lo1.x, dl); dh = dot4((q1.x >> 4) & 0x0f0f0f0f, A.hi1.x, dh);
        dl = dot4(q1.y & 0x0f0f0f0f, A.lo1.y, dl); dh = dot4((q1.y >> 4) & 0x0f0f0f0f, A.hi1.y, dh);
        dl = dot4(q1.z & 0x0f0f0f0f, A.lo1.z, dl); dh = dot4((q1.z >> 4) & 0x0f0f0f0f, A.hi1.z, dh);
        dl = dot4(q1.w & 0x0f0f0f0f, A.lo1.w, dl); dh = dot4((q1.w >> 4) & 0x0f0f0f0f, A.hi1.w, dh);
        const int isum = mul24((int)(sc & 0xff), dl) + mul24((int)(sc >> 8), dh);
#ifdef MI355_EXP_NO_MINS       // tools/build_exp_r4.sh only: a cheaper (wrong) dot, to see how far the decode time follows the instruction count
        return (d * A.yd) * (float)isum + (float)(mn & 1);
#endif
        const int msum = mul24((int)(mn & 0xff), A.bs_lo) + mul24((int)(mn >> 8), A.bs_hi);
        return (d * A.yd) * (float)isum - (dmin * A.yd) * (float)msum;
    }
};

// ---------------------------------------------------------------- Q5_K (wide role; the chunk's fifth bits are bits 2 c and 2 c + 1 of the 32 qh bytes)
template <> struct Raw<T_Q5_K> {
    u32x4_t hdr, qh, qh1, q, q1;
    __device__ __forceinline__ float probe() const { return (float)(hdr.x ^ qh.x ^ qh1.x ^ q.x ^ q.y ^ q.z ^ q.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const uint8_t *b = row + (size_t)sb * 176;
        hdr = ldw(b);
        qh = ldw(b + 16);
        qh1 = ldw(b + 32);
        q = ldw(b + 48 + L.c * 32);
        q1 = ldw(b + 64 + L.c * 32);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
        uint32_t sc, mn;
        k4_scales(hdr.y, hdr.z, hdr.w, L, sc, mn);
        const int s0 = 2 * L.c, s1 = s0 + 1;
        int dl = 0, dh = 0;
#define Q5L(w, hw) (((w) & 0x0f0f0f0f) | ((((hw) >> s0) & 0x01010101u) << 4))
#define Q5H(w, hw) ((((w) >> 4) & 0x0f0f0f0f) | ((((hw) >> s1) & 0x01010101u) << 4))
        dl = dot4(Q5L(q.x, qh.x), A.lo.x, dl); dh = dot4(Q5H(q.x, qh.x), A.hi.x, dh);
        dl = dot4(Q5L(q.y, qh.y), A.lo.y, dl); dh = dot4(Q5H(q.y, qh.y), A.hi.y, dh);
        dl = dot4(Q5L(q.z, qh.z), A.lo.z, dl); dh = dot4(Q5H(q.z, qh.z), A.hi.z, dh);
        dl = dot4(Q5L(q.w, qh.w), A.lo.w, dl); dh = dot4(Q5H(q.w, qh.w), A.hi.w, dh);
        dl = dot4(Q5L(q1.x, qh1.x), A.lo1.x, dl); dh = dot4(Q5H(q1.x, qh1.x), A.hi1.x, dh);
        dl = dot4(Q5L(q1.y, qh1.y), A.lo1.y, dl); dh = dot4(Q5H(q1.y, qh1.y), A.hi1.y, dh);
        dl = dot4(Q5L(q1.z, qh1.z), A.lo1.z, dl); dh = dot4(Q5H(q1.z, qh1.z), A.hi1.z, dh);
        dl = dot4(Q5L(q1.w, qh1.w), A.lo1.w, dl); dh = dot4(Q5H(q1.w, qh1.w), A.hi1.w, dh);
#undef Q5L
#undef Q5H
        const int isum = mul24((int)(sc & 0xff), dl) + mul24((int)(sc >> 8), dh);
        const int msum = mul24((int)(mn & 0xff), A.bs_lo) + mul24((int)(mn >> 8), A.bs_hi);
        return (d * A.yd) * (float)isum - (dmin * A.yd) * (float)msum;
    }
};

// ---------------------------------------------------------------- Q6_K (device row planes: ql | qh | scales | d)
template <> struct Raw<T_Q6_K> {
    u32x4_t ql, qh;
    int sc_lo, sc_hi;
    uint32_t dh16;
    __device__ __forceinline__ float probe() const { return (float)(ql.x ^ ql.y ^ ql.z ^ ql.w ^ qh.x ^ qh.y ^ qh.z ^ qh.w ^ (uint32_t)sc_lo ^ (uint32_t)sc_hi ^ dh16); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        ql = ldw(row + (size_t)sb * 128 + L.v * 16);
        qh = ldw(row + (size_t)nb * 128 + (size_t)sb * 64 + L.n * 32 + (L.w & 1) * 16);
        const int8_t *sc = reinterpret_cast<const int8_t *>(row + (size_t)nb * 192 + (size_t)sb * 16 + 8 * L.n + L.w);
        sc_lo = sc[0];
        sc_hi = sc[4];
        dh16 = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)dh16);
        const int s0 = 2 * (L.w >> 1), s1 = s0 + 4;
        int dl = 0, dh = 0;
#define Q6L(l, hh) (((l) & 0x0f0f0f0f) | ((((hh) >> s0) & 0x03030303u) << 4))
#define Q6H(l, hh) ((((l) >> 4) & 0x0f0f0f0f) | ((((hh) >> s1) & 0x03030303u) << 4))
        dl = dot4(Q6L(ql.x, qh.x), A.lo.x, dl); dh = dot4(Q6H(ql.x, qh.x), A.hi.x, dh);
        dl = dot4(Q6L(ql.y, qh.y), A.lo.y, dl); dh = dot4(Q6H(ql.y, qh.y), A.hi.y, dh);
        dl = dot4(Q6L(ql.z, qh.z), A.lo.z, dl); dh = dot4(Q6H(ql.z, qh.z), A.hi.z, dh);
        dl = dot4(Q6L(ql.w, qh.w), A.lo.w, dl); dh = dot4(Q6H(ql.w, qh.w), A.hi.w, dh);
#undef Q6L
#undef Q6H
        const int isum = mul24(sc_lo, dl - 32 * A.bs_lo) + mul24(sc_hi, dh - 32 * A.bs_hi);
        return (d * A.yd) * (float)isum;
    }
};

// ---------------------------------------------------------------- Q2_K (device row planes: qs | scales | (d, dmin))
// byte l of the 32-byte half n of a super-block holds elements 128 n + 32 j + l in bits 2 j .. 2 j + 1.  Lane (c, h) takes bytes 16 h .. 16 h + 15 of half
// n = c >> 1 and the fields j = 2 (c & 1) (its low 16-run) and j + 1 (its high one): the activation slice of the Q4_K role (64 c + 16 h, + 32), sub-blocks
// 4 c + h and 4 c + h + 2 (scale in the low nibble of scales[is], min in the high one)
template <> struct Raw<T_Q2_K> {
    u32x4_t q;
    uint32_t dd, sc_lo, sc_hi;
    __device__ __forceinline__ float probe() const { return (float)(q.x ^ q.y ^ q.z ^ q.w ^ dd ^ sc_lo ^ sc_hi); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        q = ldw(row + (size_t)sb * 64 + (L.c >> 1) * 32 + L.h * 16);
        const uint8_t *sc = row + (size_t)nb * 64 + (size_t)sb * 16 + 4 * L.c + L.h;
        sc_lo = sc[0]; sc_hi = sc[2];
        dd = *reinterpret_cast<const uint32_t *>(row + (size_t)nb * 80 + (size_t)sb * 4);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)(dd & 0xffff)), dmin = h2f((uint16_t)(dd >> 16));
        const int s0 = 4 * (L.c & 1), s1 = s0 + 2;
        int dl = 0, dh = 0;
        dl = dot4((q.x >> s0) & 0x03030303u, A.lo.x, dl); dh = dot4((q.x >> s1) & 0x03030303u, A.hi.x, dh);
        dl = dot4((q.y >> s0) & 0x03030303u, A.lo.y, dl); dh = dot4((q.y >> s1) & 0x03030303u, A.hi.y, dh);
        dl = dot4((q.z >> s0) & 0x03030303u, A.lo.z, dl); dh = dot4((q.z >> s1) & 0x03030303u, A.hi.z, dh);
        dl = dot4((q.w >> s0) & 0x03030303u, A.lo.w, dl); dh = dot4((q.w >> s1) & 0x03030303u, A.hi.w, dh);
        const int isum = mul24((int)(sc_lo & 15), dl) + mul24((int)(sc_hi & 15), dh);
        const int msum = mul24((int)(sc_lo >> 4), A.bs_lo) + mul24((int)(sc_hi >> 4), A.bs_hi);
        return (d * A.yd) * (float)isum - (dmin * A.yd) * (float)msum;
    }
};

// ---------------------------------------------------------------- Q3_K (device row planes: hmask | qs | scales | d)
// the same element layout as Q2_K plus the high-bit mask: bit 4 n + j of hmask[l] SET means "do not subtract 4".  code = (2 bits | hbit << 2) - 4, so
// sum code * a = dot(2 bits | hbit << 2, a) - 4 * (sum of the 16 activation codes); 6-bit scales (minus 32) unpacked from 12 bytes
template <> struct Raw<T_Q3_K> {
    u32x4_t q, hm;
    uint32_t s0w, s1w, s2w, dh16;
    __device__ __forceinline__ float probe() const { return (float)(q.x ^ q.y ^ q.z ^ q.w ^ hm.x ^ hm.y ^ hm.z ^ hm.w ^ s0w ^ s1w ^ s2w ^ dh16); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        hm = ldw(row + (size_t)sb * 32 + L.h * 16);
        q = ldw(row + (size_t)nb * 32 + (size_t)sb * 64 + (L.c >> 1) * 32 + L.h * 16);
        const uint32_t *sc = reinterpret_cast<const uint32_t *>(row + (size_t)nb * 96 + (size_t)sb * 12);
        s0w = sc[0]; s1w = sc[1]; s2w = sc[2];
        dh16 = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 108 + (size_t)sb * 2);
    }
    __device__ __forceinline__ int scale(int is) const {       // 6-bit scale is (0..15) minus 32
        const uint32_t lw = (is & 4) ? s1w : s0w;              // bytes 0-3 / 4-7 hold the low nibbles of scales 0-7 and, in their high nibbles, of 8-15
        const uint32_t b = (lw >> (8 * (is & 3))) & 0xffu;
        const int low = (is & 8) ? (int)(b >> 4) : (int)(b & 15u);
        const int high = (int)((s2w >> (8 * (is & 3) + 2 * (is >> 2))) & 3u);
        return (low | (high << 4)) - 32;
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const int n = L.c >> 1, j0 = 2 * (L.c & 1);
        const int s0 = 2 * j0, s1 = s0 + 2, b0 = 4 * n + j0, b1 = b0 + 1;
        int dl = 0, dh = 0;
#define Q3L(w, hw) ((((w) >> s0) & 0x03030303u) | ((((hw) >> b0) & 0x01010101u) << 2))
#define Q3H(w, hw) ((((w) >> s1) & 0x03030303u) | ((((hw) >> b1) & 0x01010101u) << 2))
        dl = dot4(Q3L(q.x, hm.x), A.lo.x, dl); dh = dot4(Q3H(q.x, hm.x), A.hi.x, dh);
        dl = dot4(Q3L(q.y, hm.y), A.lo.y, dl); dh = dot4(Q3H(q.y, hm.y), A.hi.y, dh);
        dl = dot4(Q3L(q.z, hm.z), A.lo.z, dl); dh = dot4(Q3H(q.z, hm.z), A.hi.z, dh);
        dl = dot4(Q3L(q.w, hm.w), A.lo.w, dl); dh = dot4(Q3H(q.w, hm.w), A.hi.w, dh);
#undef Q3L
#undef Q3H
        const int is = 4 * L.c + L.h;
        const int isum = mul24(scale(is), dl - 4 * A.bs_lo) + mul24(scale(is + 2), dh - 4 * A.bs_hi);
        return (h2f((uint16_t)dh16) * A.yd) * (float)isum;
    }
};

// ---------------------------------------------------------------- Q8_0 (device row planes: codes K | f16 scales K/32)
// lane v of a super-block owns 32-block v: its 32 codes (two 16-byte pieces) and the block scale; the activation is
// Q8_0 too (ActSlice: lo / hi = the block's 32 codes, yd = its f16 scale), dot = (float)isum * (d_w * d_a) per block
template <> struct Raw<T_Q8_0> {
    u32x4_t q0, q1;
    uint32_t dh16;
    __device__ __forceinline__ float probe() const { return (float)(q0.x ^ q0.y ^ q0.z ^ q0.w ^ q1.x ^ q1.y ^ q1.z ^ q1.w ^ dh16); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const uint8_t *b = row + (size_t)sb * 256 + L.v * 32;
        q0 = ldw(b);
        q1 = ldw(b + 16);
        dh16 = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 256 + ((size_t)sb * 8 + L.v) * 2);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        int s = 0;
        s = dot4(q0.x, A.lo.x, s); s = dot4(q0.y, A.lo.y, s); s = dot4(q0.z, A.lo.z, s); s = dot4(q0.w, A.lo.w, s);
        s = dot4(q1.x, A.hi.x, s); s = dot4(q1.y, A.hi.y, s); s = dot4(q1.z, A.hi.z, s); s = dot4(q1.w, A.hi.w, s);
        return (float)s * (h2f((uint16_t)dh16) * A.yd);
    }
};

// ---------------------------------------------------------------- Q4_0 / Q5_0 / IQ4_NL (device rows: nibbles K/2 | [qh K/32*4] | f16 scales K/32*2)
// lane v of a super-block owns 32-block v as for Q8_0: its 16 nibble bytes (elements j in the low nibbles, j + 16 in the high ones: the two halves of
// the activation slice), the fifth bits (Q5_0) and the block scale; dot products as the formats' scalar ggml_vec_dot_*_q8_0
template <int TYPE> struct RawNib32 {
    u32x4_t q;
    uint32_t qh, dh16;
    __device__ __forceinline__ float probe() const { return (float)(q.x ^ q.y ^ q.z ^ q.w ^ dh16); }
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const size_t blk = (size_t)sb * 8 + L.v, half = (size_t)nb * 128, nblk = (size_t)nb * 8;
        q = ldw(row + blk * 16);
        if (TYPE == T_Q5_0) {
            qh = *reinterpret_cast<const uint32_t *>(row + half + blk * 4);
            dh16 = *reinterpret_cast<const uint16_t *>(row + half + nblk * 4 + blk * 2);
        } else {
            qh = 0;
            dh16 = *reinterpret_cast<const uint16_t *>(row + half + blk * 2);
        }
    }
    static __device__ __forceinline__ uint32_t levels(uint32_t idx) {
        const uint32_t lo = __builtin_amdgcn_perm(0xf6eaddcfu, 0xbfad9881u, idx & 0x07070707u);
        const uint32_t hi = __builtin_amdgcn_perm(0x71594535u, 0x26190d01u, idx & 0x07070707u);
        const uint32_t m = ((idx >> 3) & 0x01010101u) * 0xffu;
        return (hi & m) | (lo & ~m);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const uint32_t qq[4] = {q.x, q.y, q.z, q.w};
        const uint32_t al[4] = {A.lo.x, A.lo.y, A.lo.z, A.lo.w}, ah[4] = {A.hi.x, A.hi.y, A.hi.z, A.hi.w};
        int s = 0, asum = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t v0 = qq[w] & 0x0f0f0f0fu, v1 = (qq[w] >> 4) & 0x0f0f0f0fu;
            if (TYPE == T_IQ4_NL) { s = dot4(levels(v0), al[w], s); s = dot4(levels(v1), ah[w], s); continue; }
            if (TYPE == T_Q5_0) {
                v0 |= ((((qh >> (4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4;
                v1 |= ((((qh >> (16 + 4 * w)) & 0xfu) * 0x00204081u) & 0x01010101u) << 4;
            }
            s = dot4(v0, al[w], s); s = dot4(v1, ah[w], s);
            asum = dot4(0x01010101u, al[w], asum); asum = dot4(0x01010101u, ah[w], asum);
        }
        if (TYPE == T_Q4_0) { s -= 8 * asum; return ((float)s * h2f((uint16_t)dh16)) * A.yd; }      // sumi * d_x * d_y, left to right
        if (TYPE == T_Q5_0) s -= 16 * asum;
        return (h2f((uint16_t)dh16) * A.yd) * (float)s;                                                // (d_x * d_y) * sumi
    }
};
template <> struct Raw<T_Q4_0> : RawNib32<T_Q4_0> {};
template <> struct Raw<T_Q5_0> : RawNib32<T_Q5_0> {};
template <> struct Raw<T_IQ4_NL> : RawNib32<T_IQ4_NL> {};

template <int TYPE> __device__ __forceinline__ LaneRole make_role(int lane) {
    LaneRole L;
    if (role_wide<TYPE>()) { L.sbl = lane >> 2; L.c = lane & 3; L.v = 2 * L.c; L.h = 0; }
    else { L.sbl = lane >> 3; L.v = lane & 7; L.c = L.v >> 1; L.h = L.v & 1; }
    L.sh = (L.c & 1) * 16; L.hi_scales = L.c >= 2;
    L.n = L.v >> 2; L.w = L.v & 3;
    L.m1 = L.hi_scales ? 0x0f0fu : 0x3f3fu; L.m2 = L.hi_scales ? 0x3030u : 0u; L.sh_mn = L.sh + (L.hi_scales ? 4 : 0);
    return L;
}

// LDS view of the staged Q8_K activation of the token
struct ActL { const int8_t *qs; const float *d; const int16_t *bs; };

template <int TYPE>
__device__ __forceinline__ ActSlice read_slice(const ActL &A, int sb, const LaneRole &L) {
    ActSlice s;
    if (act_is_q80(TYPE)) {                                // (the block-sum region of the LDS layout holds the f32 block scales)
        const int8_t *a = A.qs + sb * 256 + 32 * L.v;
        s.lo = lds16(a); s.hi = lds16(a + 16);
        s.bs_lo = 0; s.bs_hi = 0;
        s.yd = reinterpret_cast<const float *>(A.bs)[sb * 8 + L.v];
        return s;
    }
    if (role_wide<TYPE>()) {           // the chunk's 64 codes; block sums of its two 32-weight sub-blocks
        const int8_t *a = A.qs + sb * 256 + 64 * L.c;
        s.lo = lds16(a); s.lo1 = lds16(a + 16); s.hi = lds16(a + 32); s.hi1 = lds16(a + 48);
        const uint32_t *bw = reinterpret_cast<const uint32_t *>(A.bs + sb * 16 + 4 * L.c);      // four int16 sums, 8-byte aligned
        s.bs_lo = sum2_i16(bw[0]); s.bs_hi = sum2_i16(bw[1]);
    } else if (TYPE == T_Q6_K) {
        const int8_t *a = A.qs + sb * 256 + 128 * L.n + 16 * L.w;
        s.lo = lds16(a); s.hi = lds16(a + 64);
        const int16_t *b = A.bs + sb * 16 + 8 * L.n + L.w;
        s.bs_lo = b[0]; s.bs_hi = b[4];
    } else {
        const int8_t *a = A.qs + sb * 256 + 64 * L.c + 16 * L.h;
        s.lo = lds16(a); s.hi = lds16(a + 32);
        const int16_t *b = A.bs + sb * 16 + 4 * L.c + L.h;
        s.bs_lo = b[0]; s.bs_hi = b[2];
    }
    s.yd = A.d[sb];
    return s;
}

// ---- activation of the token.  Three ways in, chosen per launch (all wave-uniform):
//   direct : fuse_mode 0 and K <= 4096 -> each lane loads its own two slices straight from the Q8_K planes in global
//            memory (L2/L1 hits) into registers: no LDS, no barrier;
//   copy   : fuse_mode 0, longer K      -> the planes are copied to LDS once per workgroup;
//   fused  : fuse_mode 1 / 2            -> RMSNorm * w and / or the Q8_K quantisation happen here, result in LDS.
// Global loads are always ISSUED before the first weight loads and CONSUMED after them: memory returns in order, so
// waiting for activations that were requested after the weights would also wait for the weights (the first version did
// exactly that, and its conditional register array went through scratch: 2-6 us per launch, tools/bench_mmvq.hip).
template <int TYPE, bool COH = false>
__device__ __forceinline__ ActSlice global_slice(const MMVQArgs &a, int sb, const LaneRole &L) {
    ActSlice s;
    if (act_is_q80(TYPE)) {
        const int qo = sb * 256 + 32 * L.v;
        s.lo = cld16<COH>(a.aq0, qo); s.hi = cld16<COH>(a.aq0, qo + 16);
        s.bs_lo = 0; s.bs_hi = 0;
        s.yd = h2f((uint16_t)(cld2s<COH>(a.ad0, (sb * 8 + L.v) * 2) & 0xffff));
        return s;
    }
    if (role_wide<TYPE>()) {
        const int qo = sb * 256 + 64 * L.c;
        s.lo = cld16<COH>(a.aq, qo); s.lo1 = cld16<COH>(a.aq, qo + 16); s.hi = cld16<COH>(a.aq, qo + 32); s.hi1 = cld16<COH>(a.aq, qo + 48);
        const int bo = (sb * 16 + 4 * L.c) * 2;
        s.bs_lo = cld2s<COH>(a.abs, bo) + cld2s<COH>(a.abs, bo + 2); s.bs_hi = cld2s<COH>(a.abs, bo + 4) + cld2s<COH>(a.abs, bo + 6);
    } else if (TYPE == T_Q6_K) {
        const int qo = sb * 256 + 128 * L.n + 16 * L.w;
        s.lo = cld16<COH>(a.aq, qo); s.hi = cld16<COH>(a.aq, qo + 64);
        const int bo = (sb * 16 + 8 * L.n + L.w) * 2;
        s.bs_lo = cld2s<COH>(a.abs, bo); s.bs_hi = cld2s<COH>(a.abs, bo + 8);
    } else {
        const int qo = sb * 256 + 64 * L.c + 16 * L.h;
        s.lo = cld16<COH>(a.aq, qo); s.hi = cld16<COH>(a.aq, qo + 32);
        const int bo = (sb * 16 + 4 * L.c + L.h) * 2;
        s.bs_lo = cld2s<COH>(a.abs, bo); s.bs_hi = cld2s<COH>(a.abs, bo + 4);
    }
    s.yd = __uint_as_float(cld4<COH>(a.ad, sb * 4));
    return s;
}

template <int KB, int NT> struct StageDims {
    static constexpr int NW = NT / 64;
    static constexpr int N16 = KB * 128, NQ = (N16 + NT - 1) / NT;      // u32x4_t words of the code plane, per thread
    static constexpr int NB32 = (KB * 64 + NT - 1) / NT;                // bsums as 32-bit words, per thread
    static constexpr int NJW = (KB * 8 + NW - 1) / NW;                  // fused: 256-blocks per wave (wave w owns blocks w, w + NW, ..)
};
// the staged values travel in plain local arrays of NATIVE vector types (statically indexed after unrolling, so they stay
// in registers; arrays of HIP's u32x4_t class under a conditional were placed in scratch by hipcc)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define STAGE_REGS_DECL(KB, NT) u32x4_t sq_[StageDims<KB, NT>::NQ]; uint32_t sb_[StageDims<KB, NT>::NB32]; float sdv_ = 0.0f; \
                                f32x4_t sxv_[StageDims<KB, NT>::NJW], swv_[StageDims<KB, NT>::NJW]
#define STAGE_REGS_ARGS sq_, sb_, sdv_, sxv_, swv_

template <int KB, int NT, int FUSE, bool COH = false>
__device__ __forceinline__ void stage_issue(const MMVQArgs &a, u32x4_t (&rq)[StageDims<KB, NT>::NQ], uint32_t (&rb)[StageDims<KB, NT>::NB32], float &rdv,
                                            f32x4_t (&rxv)[StageDims<KB, NT>::NJW], f32x4_t (&rwv)[StageDims<KB, NT>::NJW], int nx_off = 0) {
    using S = StageDims<KB, NT>;
    const int tid = tid_now();
    if (FUSE == 0) {
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(a.aq);
        const int n16 = (a.K >> 8) * 16;                       // K may end inside the last pass (K % 256 == 0 only)
#pragma unroll
        for (int j = 0; j < S::NQ; j++) { const int i = j * NT + tid; rq[j] = src[i < n16 ? i : n16 - 1]; }
        const uint32_t *bsrc = reinterpret_cast<const uint32_t *>(a.abs);
        const int nbt = a.K >> 8;
#pragma unroll
        for (int j = 0; j < S::NB32; j++) { const int i = j * NT + tid; rb[j] = bsrc[i < nbt * 8 ? i : nbt * 8 - 1]; }
        rdv = a.ad[tid < nbt ? tid : nbt - 1];
    } else {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int j = 0; j < S::NJW; j++) {
            const int b = wave + S::NW * j;
            const int nbt = a.K >> 8;
            const int bc = b < nbt ? b : nbt - 1;                  // clamped: always a valid address, result unused when b is out of range
            {
                const coh_u32x4 r = cld16<COH>(a.nx, (nx_off + bc * 256 + lane * 4) * 4);
                rxv[j].x = __uint_as_float(r.x); rxv[j].y = __uint_as_float(r.y); rxv[j].z = __uint_as_float(r.z); rxv[j].w = __uint_as_float(r.w);
            }
            if (FUSE == 1) rwv[j] = *reinterpret_cast<const f32x4_t *>(a.nw + bc * 256 + lane * 4);
        }
    }
}

template <int KB, int NT, int FUSE, bool Q80 = false>
__device__ __forceinline__ ActL stage_finish(const MMVQArgs &a, const u32x4_t (&rq)[StageDims<KB, NT>::NQ], const uint32_t (&rb)[StageDims<KB, NT>::NB32], float rdv,
                                             const f32x4_t (&rxv)[StageDims<KB, NT>::NJW], const f32x4_t (&rwv)[StageDims<KB, NT>::NJW], uint8_t *smem) {
    using S = StageDims<KB, NT>;
    constexpr int K = KB * 2048;                               // LDS layout is sized for whole passes
    const int nbt = a.K >> 8;                                  // super-blocks that exist
    const int tid = tid_now();
    int8_t *qs = reinterpret_cast<int8_t *>(smem);
    float *d = reinterpret_cast<float *>(smem + K);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + K + (((K >> 8) * 4 + 15) & ~15));
    if (FUSE == 0) {
#pragma unroll
        for (int j = 0; j < S::NQ; j++) { const int i = j * NT + tid; if (i < nbt * 16) reinterpret_cast<u32x4_t *>(qs)[i] = rq[j]; }
#pragma unroll
        for (int j = 0; j < S::NB32; j++) { const int i = j * NT + tid; if (i < nbt * 8) reinterpret_cast<uint32_t *>(bs)[i] = rb[j]; }
        if (tid < nbt) d[tid] = rdv;
    } else {
        double *red = reinterpret_cast<double *>(smem + a.red_off);
        const int lane = tid & 63, wave = tid >> 6;
        float scale = 1.0f;
        if (FUSE == 1) {
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < S::NJW; j++) {
                const f32x4_t v = rxv[j];
                double t = 0.0;
                t += (double)(v.x * v.x); t += (double)(v.y * v.y); t += (double)(v.z * v.z); t += (double)(v.w * v.w);
                if (wave + S::NW * j < nbt) sum += t;
            }
            sum = wave_sum(sum);
            if (lane == 0) red[wave] = sum;
            __syncthreads();
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < S::NW; w++) tot += red[w];
            const float mean = (float)(tot / (double)a.K);
            scale = 1.0f / sqrtf(mean + a.neps);
        }
#pragma unroll
        for (int j = 0; j < S::NJW; j++) {
            const int b = wave + S::NW * j;
            if (b >= nbt) continue;                                // wave-uniform
            const int e0 = b * 256 + lane * 4;
            f32x4_t v = rxv[j];
            if (FUSE == 1) {
                const f32x4_t ww = rwv[j];
                v.x = (v.x * scale) * ww.x; v.y = (v.y * scale) * ww.y; v.z = (v.z * scale) * ww.z; v.w = (v.w * scale) * ww.w;
            }
            const float vv[4] = {v.x, v.y, v.z, v.w};
            if (Q80) {                                             // quantize_row_q8_0: the scale is stored as f16
                uint32_t packed; float dd;
                wave_quant_q80(vv, packed, dd);
                *reinterpret_cast<uint32_t *>(qs + e0) = packed;
                if ((lane & 7) == 0) reinterpret_cast<float *>(bs)[b * 8 + (lane >> 3)] = h2f(f2h(dd));
                continue;
            }
            uint32_t packed; int bsum; float dq;
            wave_quant_q8k(vv, lane, packed, bsum, dq);
            *reinterpret_cast<uint32_t *>(qs + e0) = packed;
            if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum;
            if (lane == 0) d[b] = dq;
        }
    }
    __syncthreads();
    ActL A{qs, d, bs};
    return A;
}

// slice of super-block sb, or a zero-scale slice past the end of the row (partial last pass): its dot contributes 0
template <int TYPE, bool COH = false>
__device__ __forceinline__ ActSlice global_slice_t(const MMVQArgs &a, int sb, int nb, const LaneRole &L) {
    ActSlice s = global_slice<TYPE, COH>(a, sb < nb ? sb : nb - 1, L);
    if (sb >= nb) s.yd = 0.0f;
    return s;
}
template <int TYPE>
__device__ __forceinline__ ActSlice read_slice_t(const ActL &A, int sb, int nb, const LaneRole &L) {
    ActSlice s = read_slice<TYPE>(A, sb < nb ? sb : nb - 1, L);
    if (sb >= nb) s.yd = 0.0f;
    return s;
}

// One segment, persistent waves.  A unit is (row pair, PPU passes): PPU = 2 for an even number of passes, 1 otherwise,
// so that every unit has the same shape and its loads need no condition.  NSETS units (8 Raw blocks in all) are always
// in flight ahead of the one being decoded.
//
// Control flow matters for the memory pipeline: hipcc derives each s_waitcnt vmcnt(N) from the loads issued since, and
// where a load sits under a condition it must assume the path WITHOUT it (N = 0: wait for everything, including what
// was just requested).  So everything from the first activation load to the end of the steady-state loop is straight-
// line: the staging mode is a template parameter, the first NSETS units are loaded unconditionally (waves with fewer
// units read one shared dummy block that stays in L2), and only the drain after the loop loads under conditions.
//
// PRE (decode_mega.hip): the mat-vec is one phase of a longer kernel and its activation only exists once every workgroup
// has finished the previous phase.  The weights do not depend on that, so the first NSETS units are requested FIRST,
// then sync() waits for the other workgroups (the requests are in flight meanwhile), then the activation is fetched.
struct NoSync { __device__ __forceinline__ void operator()() const {} };
template <int TYPE, int KB, int NT, int FUSE, bool PRE = false, class Sync = NoSync>
__device__ __forceinline__ void run_fast(const MMVQArgs &a, const MMVQSeg &sg, uint8_t *smem, int gw, int nw, Sync sync = Sync(), int sel_j = 0) {
    using R = Raw<TYPE>;
    constexpr int NP = role_passes<TYPE, KB>(), SBP = role_sbp<TYPE>();     // passes of the wave over a row, super-blocks per pass (lane role: narrow / wide)
    constexpr bool ACT_REGS = NP <= 2;
    constexpr bool DIRECT = ACT_REGS && FUSE == 0;
    constexpr bool COH = PRE;                            // one phase of a longer kernel: what other workgroups wrote / will read is accessed device-coherently
    const int nb = a.K >> 8;                            // super-blocks per row (the last pass may be partial: K % 256 == 0)
    constexpr int PPU = (NP % 2 == 0) ? 2 : 1;          // passes per unit
    constexpr int NCH = NP / PPU;                       // units per row pair
    constexpr int NSETS = (role_wide<TYPE>() ? 2 : 4) / PPU;   // register sets in the ring (2 rows x PPU passes each; a wide block holds twice the bytes)
    const int lane = tid_now() & 63;
    const LaneRole L = make_role<TYPE>(lane);
    const bool swiglu = a.epi == EPI_SWIGLU;
    // sel_j: which of the token's selected experts this workgroup serves (mixture-of-experts step, MMVQArgs::n_sel)
    const uint8_t *W0 = sg.W + (sg.expert_sel ? (size_t)sg.expert_sel[sel_j] * sg.expert_stride : 0);
    const MMVQSeg &ug = a.seg[1];
    const uint8_t *W1 = swiglu ? ug.W + (ug.expert_sel ? (size_t)ug.expert_sel[sel_j] * ug.expert_stride : 0) : W0;
    float *const outp = sg.out + (size_t)sel_j * (size_t)a.sel_out_stride;
    const int nx_off = sel_j * a.sel_nx_stride;
    const size_t rb0 = sg.row_bytes, rb1 = swiglu ? ug.row_bytes : sg.row_bytes;
    const int npairs = swiglu ? sg.n_rows : (sg.n_rows + 1) >> 1;
    const int my_pairs = gw < npairs ? (npairs - gw + nw - 1) / nw : 0;
    const int n_units = my_pairs * NCH;                 // unit u: pair = gw + (u / NCH) * nw, chunk = u % NCH

    // ---- 1. activation loads first
    ActSlice S0, S1;                                    // K <= 4096: this lane's slices of pass 0 / pass 1, kept in registers
    STAGE_REGS_DECL(KB, NT);
    auto issue_act = [&]() {
        if (DIRECT) {
            S0 = global_slice_t<TYPE, COH>(a, L.sbl, nb, L);
            if (NP > 1) S1 = global_slice_t<TYPE, COH>(a, SBP + L.sbl, nb, L);
        } else {
            stage_issue<KB, NT, FUSE, COH>(a, STAGE_REGS_ARGS, nx_off);
        }
    };
    if (!PRE) issue_act();

    // ---- 2. weights
    R W[NSETS][2 * PPU];                                // statically indexed after unrolling
    auto load_unit = [&](int u, R (&w)[2 * PPU], bool real) {
        const int pi = u / NCH, ch = u - pi * NCH;
        const int pair = gw + pi * nw;
        const uint8_t *ra, *rbp;
        if (swiglu) { ra = W0 + (size_t)pair * rb0; rbp = W1 + (size_t)pair * rb1; }
        else {
            ra = W0 + (size_t)(2 * pair) * rb0;
            rbp = (2 * pair + 1 < sg.n_rows) ? ra + rb0 : ra;       // odd tail: row b re-reads row a, result discarded
        }
        int sb0 = ch * PPU * SBP + L.sbl;
        if (!real) { ra = W0; rbp = W0; sb0 = L.sbl; }              // nothing to fetch: everybody's dummy is row 0, pass 0
#pragma unroll
        for (int p = 0; p < PPU; p++) {
            int sb = sb0 + SBP * p;
            if (sb >= nb) sb = nb - 1;                  // tail of a partial last pass: any valid block, its slice scale is zero
            w[2 * p].load(ra, nb, sb, L);
            w[2 * p + 1].load(rbp, nb, sb, L);
        }
    };
#pragma unroll
    for (int s = 0; s < NSETS; s++) load_unit(s, W[s], s < n_units);
    if (PRE) { sync(); issue_act(); }

    // ---- 3. activations into place (workgroup barrier inside the LDS paths: reached by every wave)
    ActL AL{nullptr, nullptr, nullptr};
    if (!DIRECT) {
        AL = stage_finish<KB, NT, FUSE, act_is_q80(TYPE)>(a, STAGE_REGS_ARGS, smem);
        if (ACT_REGS) {
            S0 = read_slice_t<TYPE>(AL, L.sbl, nb, L);
            if (NP > 1) S1 = read_slice_t<TYPE>(AL, SBP + L.sbl, nb, L);
        }
    }

    float acc0 = 0.0f, acc1 = 0.0f;
    auto compute_unit = [&](int u, const R (&w)[2 * PPU]) {
        const int pi = u / NCH, ch = u - pi * NCH;
#pragma unroll
        for (int p = 0; p < PPU; p++) {
            const ActSlice sl = ACT_REGS ? (ch * PPU + p == 0 ? S0 : S1) : read_slice_t<TYPE>(AL, (ch * PPU + p) * SBP + L.sbl, nb, L);
            acc0 += w[2 * p].dot(sl, L);
            acc1 += w[2 * p + 1].dot(sl, L);
        }
        if (ch == NCH - 1) {                            // pair finished
            const float v0 = wave_sum(acc0), v1 = wave_sum(acc1);
            acc0 = 0.0f; acc1 = 0.0f;
            if (lane == 0) {
                const int pair = gw + pi * nw;
                if (swiglu) {
                    cstf<COH>(outp + pair, (v0 / (1.0f + expf(-v0))) * v1);
                } else {
                    const int row0 = 2 * pair;
                    const bool has1 = row0 + 1 < sg.n_rows;
                    if (a.epi == EPI_ADD) {
                        cstf<COH>(outp + row0, cldf<COH>(sg.resid + row0) + v0);
                        if (has1) cstf<COH>(outp + row0 + 1, cldf<COH>(sg.resid + row0 + 1) + v1);
                    } else {
                        cstf<COH>(outp + row0, v0);
                        if (has1) cstf<COH>(outp + row0 + 1, v1);
                    }
                }
            }
        }
    };
    int u = 0;
#pragma unroll 1
    for (; u + 2 * NSETS <= n_units; u += NSETS) {       // steady state: every refill exists
#pragma unroll
        for (int s = 0; s < NSETS; s++) {
            compute_unit(u + s, W[s]);
            load_unit(u + NSETS + s, W[s], true);
        }
    }
    // drain: units u .. n_units-1 (fewer than 2 * NSETS); sets 0 .. NSETS-1 hold u .. u+NSETS-1
#pragma unroll
    for (int s = 0; s < NSETS; s++) {
        if (u + s < n_units) {
            compute_unit(u + s, W[s]);
            if (u + NSETS + s < n_units) load_unit(u + NSETS + s, W[s], true);
        }
    }
#pragma unroll
    for (int s = 0; s < NSETS; s++)
        if (u + NSETS + s < n_units) compute_unit(u + NSETS + s, W[s]);
}

}  // namespace

}  // namespace mi355
