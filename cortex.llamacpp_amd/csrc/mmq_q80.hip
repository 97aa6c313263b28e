// mmq_q80.hip — prompt batches against Q8_0 weights on the matrix cores (mul_mat_q for Q8_0).
//
//   out[t][r] = sum over 32-blocks b of  (float)isum_b * (d_w[r][b] * d_a[t][b]),   isum_b = sum_{k in b} w[r][k] * a[t][k]
//
// exactly ggml_vec_dot_q8_0_q8_0: integer block sums, one f32 multiply-add per block, blocks added in order — so unlike
// the K-quant kernels (whose super-block sums are re-associated) this one reproduces the CPU result bit for bit.
//
// v_mfma_i32_32x32x32_i8 contracts K = 32 per instruction = one Q8_0 block: A = 32 tokens x 32 codes (lane = token
// lane & 31, k-half lane >> 5, 16 bytes straight from the activation plane), B = 32 weight rows x 32 codes (lane = row,
// k-half; 16 bytes straight from the device row [codes K][scales K/32]).  The result tile holds 16 tokens of one weight
// row per lane, so d_w is a per-lane scalar and the 16 activation scales of the block come from LDS (staged once per
// workgroup as f32, [block][token]: four broadcast ds_read_b128 per block).  Every block costs one MFMA (32 cycles) and
// 16 x (convert, scale product, multiply-add): the fold is the bound (VALU), ~1/8 of the int8 matrix peak — against one
// pass over the weights per 16 tokens with the tiled mat-vec this replaces for T >= 32.
// Workgroup = 4 waves = 128 weight rows x 32 tokens; the four waves share the token tile (L1 hits on the activations).
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int Q80_UNROLL = 4;

__global__ __launch_bounds__(256) void mmq_q80_kernel(const uint8_t *__restrict__ W, size_t row_bytes, int n_rows, int K, int T,
                                                      const int8_t *__restrict__ aq, const uint16_t *__restrict__ ad, float *__restrict__ out,
                                                      int ld_out, const float *__restrict__ resid) {
    extern __shared__ __attribute__((aligned(16))) float s_da[];      // [K / 32][32 tokens]
    const int nb = K >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const int t0 = blockIdx.y * 32, r0 = blockIdx.x * 128 + wave * 32;
    for (int i = tid; i < nb * 32; i += 256) {
        const int b = i >> 5, m = i & 31;
        const int t = t0 + m < T ? t0 + m : T - 1;
        s_da[i] = h2f(ad[(size_t)t * nb + b]);
    }
    __syncthreads();
    if (r0 >= n_rows) return;                                          // (after the barrier: wave-uniform)
    const int row = r0 + n < n_rows ? r0 + n : n_rows - 1;
    const uint8_t *wrow = W + (size_t)row * row_bytes + 16 * kg;
    const uint16_t *wd = reinterpret_cast<const uint16_t *>(W + (size_t)row * row_bytes + K);
    const int tok = t0 + n < T ? t0 + n : T - 1;
    const int8_t *arow = aq + (size_t)tok * K + 16 * kg;
    float facc[16];
#pragma unroll
    for (int r = 0; r < 16; r++) facc[r] = 0.0f;
    i32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0;

    auto fold = [&](const i32x16 &c, float dw, int b) {
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            const f32x4 da4 = *reinterpret_cast<const f32x4 *>(s_da + b * 32 + 8 * rq + 4 * kg);    // tokens 8 rq + 4 kg + (0..3)
#pragma unroll
            for (int ri = 0; ri < 4; ri++) facc[rq * 4 + ri] += (float)c[rq * 4 + ri] * (dw * da4[ri]);
        }
    };
    int b = 0;
    for (; b + Q80_UNROLL <= nb; b += Q80_UNROLL) {                    // the loads of a group are issued before its first MFMA
        i32x4 a[Q80_UNROLL], w[Q80_UNROLL];
        uint16_t dh[Q80_UNROLL];
#pragma unroll
        for (int u = 0; u < Q80_UNROLL; u++) {
            a[u] = *reinterpret_cast<const i32x4 *>(arow + (size_t)(b + u) * 32);
            w[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(wrow + (size_t)(b + u) * 32));
            dh[u] = wd[b + u];
        }
#pragma unroll
        for (int u = 0; u < Q80_UNROLL; u++) {
            const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], w[u], z, 0, 0, 0);
            fold(c, h2f(dh[u]), b + u);
        }
    }
    for (; b < nb; b++) {
        const i32x4 a = *reinterpret_cast<const i32x4 *>(arow + (size_t)b * 32);
        const i32x4 w = *reinterpret_cast<const i32x4 *>(wrow + (size_t)b * 32);
        const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, z, 0, 0, 0);
        fold(c, h2f(wd[b]), b);
    }
    if (r0 + n < n_rows) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int t = t0 + 8 * (r >> 2) + 4 * kg + (r & 3);
            if (t < T) {
                const size_t o = (size_t)t * ld_out + row;
                out[o] = resid ? resid[o] + facc[r] : facc[r];
            }
        }
    }
}

}  // namespace

// Q4_0 / Q5_0 / IQ4_NL rows as Q8_0 device rows ([codes K][scales K/32]): code = nibble - 8, (nibble | fifth bit) - 16, level[nibble] - all int8 -
// with the block's own f16 scale.  The copy is EXACT (same integers, same scales), so the kernel above computes the same block sums as the format's
// own vec_dot; it is made once at load for prompt batches (1.06 B per weight beside the file's 0.56 - 0.69).
__global__ __launch_bounds__(256) void expand_nib32_q80_kernel(int type, const uint8_t *__restrict__ W, size_t row_bytes, int n_rows, int K, uint8_t *__restrict__ dst,
                                                               size_t dst_row) {
    const int row = blockIdx.y;
    const int nblk = K >> 5;
    const size_t half = (size_t)K >> 1;
    const uint8_t *r = W + (size_t)row * row_bytes;
    uint8_t *o = dst + (size_t)row * dst_row;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < K; e += gridDim.x * 256) {
        const int b = e >> 5, j = e & 31;
        const int nib = (r[(size_t)b * 16 + (j & 15)] >> (4 * (j >> 4))) & 0x0f;
        int code;
        if (type == T_Q4_0) code = nib - 8;
        else if (type == T_Q5_0) {
            const uint32_t qh = *reinterpret_cast<const uint32_t *>(r + half + (size_t)b * 4);
            code = (nib | (int)(((qh >> j) & 1u) << 4)) - 16;
        } else code = iq4nl_value(nib);
        o[e] = (uint8_t)(int8_t)code;
        if (j == 0) {
            const uint16_t d = *reinterpret_cast<const uint16_t *>(r + half + (type == T_Q5_0 ? (size_t)nblk * 4 : 0) + (size_t)b * 2);
            *reinterpret_cast<uint16_t *>(o + (size_t)K + (size_t)b * 2) = d;
        }
    }
}
size_t mmq_q80_copy_bytes(int type, int64_t n_rows, int K) {
    if ((type != T_Q4_0 && type != T_Q5_0 && type != T_IQ4_NL) || (K % 32) != 0 || K > 16384) return 0;
    return (size_t)n_rows * dev_row_bytes(T_Q8_0, K);
}
hipError_t launch_expand_q80_copy(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, uint8_t *dst, hipStream_t st) {
    if (!mmq_q80_copy_bytes(type, n_rows, K)) return hipErrorInvalidValue;
    for (int r0 = 0; r0 < n_rows; r0 += 65535) {
        const int nr = n_rows - r0 < 65535 ? n_rows - r0 : 65535;
        hipLaunchKernelGGL(expand_nib32_q80_kernel, dim3((unsigned)((K + 255) / 256 < 8 ? (K + 255) / 256 : 8), nr), dim3(256), 0, st, type, W + (size_t)r0 * row_bytes, row_bytes, nr, K,
                           dst + (size_t)r0 * dev_row_bytes(T_Q8_0, K), dev_row_bytes(T_Q8_0, K));
    }
    return hipGetLastError();
}

bool mmq_q80_applicable(int type, int K, int T) { return type == T_Q8_0 && T >= 32 && K >= 32 && (K % 32) == 0 && K <= 16384; }

hipError_t launch_mmq_q80(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q, float *out, int ld_out,
                          const float *resid, hipStream_t st) {
    if (!mmq_q80_applicable(T_Q8_0, K, T) || !q.qs0 || !q.d0) return hipErrorInvalidValue;
    const size_t lds = (size_t)(K >> 5) * 32 * sizeof(float);
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_q80_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((T + 31) / 32));
    hipLaunchKernelGGL(mmq_q80_kernel, grid, dim3(256), lds, st, W, row_bytes, n_rows, K, T, q.qs0, q.d0, out, ld_out, resid);
    return hipGetLastError();
}

}  // namespace mi355
