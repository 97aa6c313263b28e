// quant_dev.h — wave-level activation quantisation shared by act.hip (stand-alone kernels) and mmvq.hip
// (fused RMSNorm + quantise prologue).  One wave quantises one 256-element block: lane l holds elements 4l..4l+3.
// Bit-identical to quantize_row_q8_K / quantize_row_q8_0 of the CPU backend.
#pragma once

#include "dev_common.h"

namespace mi355 {

// Q8_K: iscale = -127 / (signed value of the FIRST element with the largest magnitude); codes = min(127, rint(iscale*x));
// d = 1/iscale; bsum16 = sum of the 16-element group this lane belongs to (valid on every lane).
__device__ __forceinline__ void wave_quant_q8k(const float (&vv)[4], int lane, uint32_t &packed, int &bsum16, float &d) {
    // largest magnitude of the block (f32 max over the wave), then the FIRST element holding it: lowest lane whose own
    // first match exists (ballot + find-first-set), its value fetched with one v_readlane
    const float a0 = fabsf(vv[0]), a1 = fabsf(vv[1]), a2 = fabsf(vv[2]), a3 = fabsf(vv[3]);
    const float amax = wave_max(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
    const float vsrc = m0 ? vv[0] : m1 ? vv[1] : m2 ? vv[2] : vv[3];
    const unsigned long long hit = __ballot(m0 || m1 || m2 || m3);
    const int src_lane = hit ? __builtin_ctzll(hit) : 0;
    const float vmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vsrc), src_lane));
    (void)lane;
    int qi[4] = {0, 0, 0, 0};
    d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.0f / vmax;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int t = __float2int_rn(iscale * vv[i]);
            qi[i] = t > 127 ? 127 : t;
        }
        d = 1.0f / iscale;
    }
    packed = (uint32_t)(qi[0] & 0xff) | ((uint32_t)(qi[1] & 0xff) << 8) | ((uint32_t)(qi[2] & 0xff) << 16) | ((uint32_t)(qi[3] & 0xff) << 24);
    int bs = qi[0] + qi[1] + qi[2] + qi[3];
    bs += dpp_i<DPP_QP_1032>(bs);       // 16 codes = 4 lanes = one quad
    bs += dpp_i<DPP_QP_2301>(bs);
    bsum16 = bs;
}

// N blocks of one wave at once (the weight stream's quantising prologue: up to ten 256-blocks per wave).  The same arithmetic block by block; the two IEEE
// divisions per block (-127 / max, 1 / iscale: ~10 instructions each, wave-uniform operands) are done ONCE for all blocks, block i's operand in lane i, and
// the quotients come back through v_readlane - the prologue is instruction-bound (8 waves x 7 blocks x ~80 instructions on an ffn_down launch).
template <int N>
__device__ __forceinline__ void wave_quant_q8k_batch(const float (&vv)[N][4], int lane, uint32_t (&packed)[N], int (&bsum16)[N], float (&d)[N]) {
    static_assert(N >= 1 && N <= 64, "one lane per block");
    bool nz[N];
    float vmax_l = 1.0f;                                       // lane i: block i's first element of the largest magnitude (1 where there is none)
#pragma unroll
    for (int i = 0; i < N; i++) {
        const float a0 = fabsf(vv[i][0]), a1 = fabsf(vv[i][1]), a2 = fabsf(vv[i][2]), a3 = fabsf(vv[i][3]);
        const float amax = wave_max(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
        const bool m0 = a0 == amax, m1 = a1 == amax, m2 = a2 == amax, m3 = a3 == amax;
        const float vsrc = m0 ? vv[i][0] : m1 ? vv[i][1] : m2 ? vv[i][2] : vv[i][3];
        const unsigned long long hit = __ballot(m0 || m1 || m2 || m3);
        const int src_lane = hit ? __builtin_ctzll(hit) : 0;
        const float vmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vsrc), src_lane));
        nz[i] = amax != 0.0f;
        if (lane == i && nz[i]) vmax_l = vmax;
    }
    const float iscale_l = -127.0f / vmax_l;
    const float d_l = 1.0f / iscale_l;
#pragma unroll
    for (int i = 0; i < N; i++) {
        int qi[4] = {0, 0, 0, 0};
        d[i] = 0.0f;
        if (nz[i]) {                                           // (wave-uniform)
            const float iscale = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(iscale_l), i));
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = __float2int_rn(iscale * vv[i][k]);
                qi[k] = t > 127 ? 127 : t;
            }
            d[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d_l), i));
        }
        packed[i] = (uint32_t)(qi[0] & 0xff) | ((uint32_t)(qi[1] & 0xff) << 8) | ((uint32_t)(qi[2] & 0xff) << 16) | ((uint32_t)(qi[3] & 0xff) << 24);
        int bs = qi[0] + qi[1] + qi[2] + qi[3];
        bs += dpp_i<DPP_QP_1032>(bs);
        bs += dpp_i<DPP_QP_2301>(bs);
        bsum16[i] = bs;
    }
}

// Q8_0: per 32 elements (8 lanes): d = amax/127, codes = roundf(x/d); d returned as f32 (store as f16)
__device__ __forceinline__ void wave_quant_q80(const float (&vv)[4], uint32_t &packed, float &d) {
    float am = fmaxf(fmaxf(fabsf(vv[0]), fabsf(vv[1])), fmaxf(fabsf(vv[2]), fabsf(vv[3])));
    am = fmaxf(am, dpp_f<DPP_QP_1032>(am));   // 32 values = 8 lanes
    am = fmaxf(am, dpp_f<DPP_QP_2301>(am));
    am = fmaxf(am, dpp_f<DPP_HALF_MIRROR>(am));
    d = am / 127.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    int qi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) qi[i] = (int)roundf(vv[i] * id);
    packed = (uint32_t)(qi[0] & 0xff) | ((uint32_t)(qi[1] & 0xff) << 8) | ((uint32_t)(qi[2] & 0xff) << 16) | ((uint32_t)(qi[3] & 0xff) << 24);
}

// ---------------------------------------------------------------- dequantise one element of a device row
__device__ __forceinline__ void k4_scale_min(int j, const uint8_t *p, int &sc, int &mn) {
    if (j < 4) { sc = p[j] & 63; mn = p[j + 4] & 63; }
    else { sc = (p[j + 4] & 0x0f) | ((p[j - 4] >> 6) << 4); mn = (p[j + 4] >> 4) | ((p[j] >> 6) << 4); }
}

// IQ4_NL's code book (ggml-common.h kvalues_iq4nl): sixteen int8 levels, denser around zero
__device__ __forceinline__ int iq4nl_value(int nib) {
    // packed little-endian: levels 0-3, 4-7, 8-11, 12-15
    const uint32_t w = nib < 8 ? (nib < 4 ? 0xbfad9881u : 0xf6eaddcfu) : (nib < 12 ? 0x26190d01u : 0x71594535u);
    return (int)(int8_t)((w >> (8 * (nib & 3))) & 0xffu);
}

__device__ __forceinline__ float dequant_elem(int type, const uint8_t *row, int K, int e) {
    switch (type) {
        case T_F32: return reinterpret_cast<const float *>(row)[e];
        case T_F16: return h2f(reinterpret_cast<const uint16_t *>(row)[e]);
        case T_Q8_0: {
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + K + (e >> 5) * 2));
            return __fmul_rn((float)(int8_t)row[e], d);
        }
        case T_Q4_0: case T_IQ4_NL: case T_Q5_0: {   // element r of block b: nibble r / 16 of qs[16 b + r % 16]; Q5_0: bit r of qh is the fifth bit
            const int b = e >> 5, r = e & 31, nblk = K >> 5;
            const int nib = (row[(size_t)b * 16 + (r & 15)] >> (4 * (r >> 4))) & 0x0f;
            const size_t half = (size_t)K >> 1;
            if (type == T_Q5_0) {
                const uint32_t qh = *reinterpret_cast<const uint32_t *>(row + half + (size_t)b * 4);
                const float d = h2f(*reinterpret_cast<const uint16_t *>(row + half + (size_t)nblk * 4 + (size_t)b * 2));
                return __fmul_rn((float)((nib | (int)(((qh >> r) & 1u) << 4)) - 16), d);
            }
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + half + (size_t)b * 2));
            if (type == T_Q4_0) return __fmul_rn((float)(nib - 8), d);
            return __fmul_rn(d, (float)iq4nl_value(nib));
        }
        case T_Q4_K: case T_Q5_K: {
            const int bsz = type == T_Q4_K ? 144 : 176;
            const uint8_t *b = row + (size_t)(e >> 8) * bsz;
            const int r = e & 255, j = r >> 5, l = r & 31, c = j >> 1;
            const float d = h2f(*reinterpret_cast<const uint16_t *>(b)), dm = h2f(*reinterpret_cast<const uint16_t *>(b + 2));
            int sc, mn;
            k4_scale_min(j, b + 4, sc, mn);
            const uint8_t *qs = b + (type == T_Q4_K ? 16 : 48);
            int q = (j & 1) ? (qs[32 * c + l] >> 4) : (qs[32 * c + l] & 0x0f);
            if (type == T_Q5_K && ((b[16 + l] >> j) & 1)) q += 16;
            return __fsub_rn(__fmul_rn(__fmul_rn(d, (float)sc), (float)q), __fmul_rn(dm, (float)mn));
        }
        case T_Q2_K: {   // element (n, j, l) of block sb: bits 2 j .. 2 j + 1 of qs[32 n + l]; sub-block is = 8 n + 2 j + l / 16
            const int nb = K >> 8, sb = e >> 8, r = e & 255, n = r >> 7, j = (r >> 5) & 3, l = r & 31, is = 8 * n + 2 * j + (l >> 4);
            const uint8_t q = row[(size_t)sb * 64 + 32 * n + l], sc = row[(size_t)nb * 64 + (size_t)sb * 16 + is];
            const uint16_t *dm = reinterpret_cast<const uint16_t *>(row + (size_t)nb * 80 + (size_t)sb * 4);
            const float dl = __fmul_rn(h2f(dm[0]), (float)(sc & 0xf)), ml = __fmul_rn(h2f(dm[1]), (float)(sc >> 4));
            return __fsub_rn(__fmul_rn(dl, (float)((q >> (2 * j)) & 3)), ml);
        }
        case T_Q3_K: {
            const int nb = K >> 8, sb = e >> 8, r = e & 255, n = r >> 7, j = (r >> 5) & 3, l = r & 31, is = 8 * n + 2 * j + (l >> 4);
            const uint8_t q = row[(size_t)nb * 32 + (size_t)sb * 64 + 32 * n + l], hm = row[(size_t)sb * 32 + l];
            const uint8_t *s12 = row + (size_t)nb * 96 + (size_t)sb * 12;
            const int low = is < 8 ? (s12[is] & 0xf) : (s12[is - 8] >> 4), high = (s12[8 + (is & 3)] >> (2 * (is >> 2))) & 3;
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + (size_t)nb * 108 + (size_t)sb * 2));
            const int code = (int)((q >> (2 * j)) & 3) - (((hm >> (4 * n + j)) & 1) ? 0 : 4);
            return __fmul_rn(__fmul_rn(d, (float)((low | (high << 4)) - 32)), (float)code);
        }
        case T_Q6_K: {
            const int nb = K >> 8, sb = e >> 8, r = e & 255, n = r >> 7, rr = r & 127, k = rr >> 5, l = rr & 31;
            const uint8_t *ql = row + (size_t)sb * 128 + n * 64, *qh = row + (size_t)nb * 128 + (size_t)sb * 64 + n * 32;
            const int8_t *sc = reinterpret_cast<const int8_t *>(row + (size_t)nb * 192 + (size_t)sb * 16 + n * 8);
            const float d = h2f(*reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2));
            const int lo = (k & 1) ? ql[l + 32] : ql[l];
            const int nib = (k & 2) ? (lo >> 4) : (lo & 0x0f);
            const int q = (nib | (((qh[l] >> (2 * k)) & 3) << 4)) - 32;
            return __fmul_rn(__fmul_rn(d, (float)sc[2 * k + (l >> 4)]), (float)q);
        }
    }
    return 0.0f;
}


}  // namespace mi355
