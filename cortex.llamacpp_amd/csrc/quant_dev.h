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

}  // namespace mi355
