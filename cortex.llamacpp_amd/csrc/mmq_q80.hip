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
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int Q80_UNROLL = 4;

// MT token tiles of 32 per wave: a workgroup covers 128 weight rows x 32 MT tokens, so every weight byte is fetched T / (32 MT) times instead of T / 32
// (with MT = 1 an 8B-shaped prompt is bound by those re-reads out of L2: 8.5k tok/s).  The activation scales are staged per chunk of Q80_KC blocks.
constexpr int Q80_KC = 64;

template <int MT>
__global__ __launch_bounds__(256) void mmq_q80_kernel(const uint8_t *__restrict__ W, size_t row_bytes, int n_rows, int K, int T,
                                                      const int8_t *__restrict__ aq, const uint16_t *__restrict__ ad, float *__restrict__ out,
                                                      int ld_out, const float *__restrict__ resid) {
    __shared__ __attribute__((aligned(16))) float s_da[Q80_KC * 32 * MT];      // [block of the chunk][token of the workgroup's tile]
    constexpr int TT = 32 * MT;
    const int nb = K >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, kg = lane >> 5;
    const int t0 = blockIdx.y * TT, r0 = blockIdx.x * 128 + wave * 32;
    const bool rows_ok = r0 < n_rows;                                  // (wave-uniform; such a wave still takes part in the barriers)
    const int row = r0 + n < n_rows ? r0 + n : n_rows - 1;
    const uint8_t *wrow = W + (size_t)row * row_bytes + 16 * kg;
    const uint16_t *wd = reinterpret_cast<const uint16_t *>(W + (size_t)row * row_bytes + K);
    const int8_t *arow[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int tok = t0 + 32 * m + n < T ? t0 + 32 * m + n : T - 1;
        arow[m] = aq + (size_t)tok * K + 16 * kg;
    }
    float facc[MT][16];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) facc[m][r] = 0.0f;
    i32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0;

    // per output and block: scale = d_w * d_a (exact: two f16 values), product = (float)isum * scale, sum += product - the CPU's three roundings in the CPU's
    // order; written on pairs so that the multiplies and the add become v_pk_mul_f32 / v_pk_add_f32 (two outputs per instruction, same IEEE results)
    auto fold = [&](const i32x16 &c, float dw, int bl, int m) {        // bl: block within the chunk
        const f32x2 dw2 = {dw, dw};
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            const f32x4 da4 = *reinterpret_cast<const f32x4 *>(s_da + bl * TT + 32 * m + 8 * rq + 4 * kg);    // tokens 8 rq + 4 kg + (0..3) of tile m
#pragma unroll
            for (int rp = 0; rp < 2; rp++) {
                const f32x2 da2 = {da4[2 * rp], da4[2 * rp + 1]};
                const f32x2 cf = {(float)c[rq * 4 + 2 * rp], (float)c[rq * 4 + 2 * rp + 1]};
                const f32x2 sc = dw2 * da2;
                const f32x2 pr = cf * sc;
                f32x2 acc = {facc[m][rq * 4 + 2 * rp], facc[m][rq * 4 + 2 * rp + 1]};
                acc = acc + pr;
                facc[m][rq * 4 + 2 * rp] = acc.x; facc[m][rq * 4 + 2 * rp + 1] = acc.y;
            }
        }
    };
    for (int b0 = 0; b0 < nb; b0 += Q80_KC) {
        const int nbc = nb - b0 < Q80_KC ? nb - b0 : Q80_KC;
        __syncthreads();                                               // everyone is done with the previous chunk's scales
        for (int i = tid; i < nbc * TT; i += 256) {
            const int b = i / TT, m = i - b * TT;
            const int t = t0 + m < T ? t0 + m : T - 1;
            s_da[i] = h2f(ad[(size_t)t * nb + b0 + b]);
        }
        __syncthreads();
        if (!rows_ok) continue;
        int b = 0;
        for (; b + Q80_UNROLL <= nbc; b += Q80_UNROLL) {               // the loads of a group are issued before its first MFMA
            i32x4 w[Q80_UNROLL];
            uint16_t dh[Q80_UNROLL];
#pragma unroll
            for (int u = 0; u < Q80_UNROLL; u++) {
                w[u] = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(wrow + (size_t)(b0 + b + u) * 32));
                dh[u] = wd[b0 + b + u];
            }
#pragma unroll
            for (int m = 0; m < MT; m++) {
                i32x4 a[Q80_UNROLL];
#pragma unroll
                for (int u = 0; u < Q80_UNROLL; u++) a[u] = *reinterpret_cast<const i32x4 *>(arow[m] + (size_t)(b0 + b + u) * 32);
#pragma unroll
                for (int u = 0; u < Q80_UNROLL; u++) {
                    const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[u], w[u], z, 0, 0, 0);
                    fold(c, h2f(dh[u]), b + u, m);
                }
            }
        }
        for (; b < nbc; b++) {
            const i32x4 w = *reinterpret_cast<const i32x4 *>(wrow + (size_t)(b0 + b) * 32);
            const float dw = h2f(wd[b0 + b]);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const i32x4 a = *reinterpret_cast<const i32x4 *>(arow[m] + (size_t)(b0 + b) * 32);
                const i32x16 c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, w, z, 0, 0, 0);
                fold(c, dw, b, m);
            }
        }
    }
    if (rows_ok && r0 + n < n_rows) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int t = t0 + 32 * m + 8 * (r >> 2) + 4 * kg + (r & 3);
                if (t < T) {
                    const size_t o = (size_t)t * ld_out + row;
                    out[o] = resid ? resid[o] + facc[m][r] : facc[m][r];
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
    // four token tiles per wave once that still leaves a workgroup per CU; each output keeps its own block order, so the result does not depend on MT
    const long wg4 = (long)((n_rows + 127) / 128) * ((T + 127) / 128);
    if (T >= 128 && wg4 >= num_cu()) {
        const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((T + 127) / 128));
        hipLaunchKernelGGL((mmq_q80_kernel<4>), grid, dim3(256), 0, st, W, row_bytes, n_rows, K, T, q.qs0, q.d0, out, ld_out, resid);
    } else if (T >= 64 && wg4 * 2 >= num_cu()) {
        const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((T + 63) / 64));
        hipLaunchKernelGGL((mmq_q80_kernel<2>), grid, dim3(256), 0, st, W, row_bytes, n_rows, K, T, q.qs0, q.d0, out, ld_out, resid);
    } else {
        const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((T + 31) / 32));
        hipLaunchKernelGGL((mmq_q80_kernel<1>), grid, dim3(256), 0, st, W, row_bytes, n_rows, K, T, q.qs0, q.d0, out, ld_out, resid);
    }
    return hipGetLastError();
}

}  // namespace mi355
