// mmf.hip — batched contraction against F16 weight tensors on the matrix cores (ggml_mul_mat with an f16 src0: the CPU path rounds the activations to f16 and
// takes vec_dot_f16, f32 accumulation; SURVEY.md §8a row a10 for files that keep their projections in f16 — the reference's embedding smoke model is one, Makefile:6).
//
//   Y[t][n] (+ resid[t][n]) = sum_k W[n][k] * f16(X[t][k])
//
// Both operands are K-contiguous rows, which is exactly the register layout of v_mfma_f32_32x32x16_f16 (lane = (row m, k-group kg): eight consecutive k), so
// neither goes through LDS: a wave owns a 64 x 64 (weight rows x tokens) tile as 2 x 2 MFMA tiles and reads its A rows as 16-byte loads and its B rows as two
// float4 loads converted to halfs in registers; the four waves of a workgroup form a 128 x 128 tile and share their rows through the caches.  The products are
// exact in f32 and summed in f32 by the matrix pipe (the CPU sums lane-wise partials of the same products: equal up to f32 re-association).
// One wave per (row, token) - mmv_float_kernel, misc.hip - stays the path for single tokens and for f32 tensors.
#include "kernels.h"

namespace mi355 {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mmf16_kernel(const _Float16 *W, int N, int K, const float *X, int T, float *Y, int ldy, const float *resid) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64, t0 = blockIdx.y * 128 + (wave >> 1) * 64;
    if (n0 >= N || t0 >= T) return;                             // (wave-uniform)
    const _Float16 *wr[2];
    const float *xr[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int n = n0 + 32 * i + m, t = t0 + 32 * i + m;     // rows past the end read the last row; their results are never stored
        wr[i] = W + (size_t)(n < N ? n : N - 1) * K + 8 * kg;
        xr[i] = X + (size_t)(t < T ? t : T - 1) * K + 8 * kg;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
#pragma unroll 2
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            a[i] = *reinterpret_cast<const f16x8 *>(wr[i] + k0);
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xr[i] + k0), x1 = *reinterpret_cast<const f32x4 *>(xr[i] + k0 + 4);
            b[i][0] = (_Float16)x0.x; b[i][1] = (_Float16)x0.y; b[i][2] = (_Float16)x0.z; b[i][3] = (_Float16)x0.w;
            b[i][4] = (_Float16)x1.x; b[i][5] = (_Float16)x1.y; b[i][6] = (_Float16)x1.z; b[i][7] = (_Float16)x1.w;
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // register r of a lane: weight row (r & 3) + 8 (r >> 2) + 4 kg of the tile, token = lane & 31: four consecutive rows per 16-byte store
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int t = t0 + 32 * j + m;
            if (t >= T) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = n0 + 32 * i + 8 * q + 4 * kg;
                float *dst = Y + (size_t)t * ldy + n;
                const float *rs = resid ? resid + (size_t)t * ldy + n : nullptr;
                if (n + 3 < N && (ldy & 3) == 0) {
                    f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                    if (rs) { const f32x4 r4 = *reinterpret_cast<const f32x4 *>(rs); v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w; }
                    *reinterpret_cast<f32x4 *>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (n + e < N) dst[e] = rs ? rs[e] + acc[i][j][4 * q + e] : acc[i][j][4 * q + e];
                }
            }
        }
}

}  // namespace

bool mmf16_applicable(int type, int n_rows, int K, int T, const void *W, const void *x, const void *y) {
    return type == T_F16 && T >= 8 && n_rows >= 32 && (K % 16) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

hipError_t launch_mmf16(const uint8_t *W, int n_rows, int K, const float *x, int T, float *y, int ld_out, const float *resid, hipStream_t st) {
    const dim3 grid((unsigned)((n_rows + 127) / 128), (unsigned)((T + 127) / 128));
    hipLaunchKernelGGL(mmf16_kernel, grid, dim3(256), 0, st, reinterpret_cast<const _Float16 *>(W), n_rows, K, x, T, y, ld_out, resid);
    return hipGetLastError();
}

}  // namespace mi355
