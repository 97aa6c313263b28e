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
#include <cstdlib>

#include "kernels.h"

namespace mi355 {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// WT = 32 x 32 tiles a wave holds per side: 2 -> a workgroup covers 128 x 128 outputs, 1 -> 64 x 64 (four times the workgroups: shapes whose 128-tiles would leave
// most of the 256 CUs idle - the LLaVA tower's 577 x 1024 projections are 40 of them).  Every output is the same chain of matrix-core steps over k either way.
// XH: the activations arrive as f16 already (launch_f32_to_f16 - the same rounding, done once per row instead of once per workgroup column that reads it).
// Epilogue: y = resid + (acc + bias) * scale, each part optional (bias per output column, then the scale, then the residual row: the order of the separate
// bias / scale / add launches it replaces).
template <int WT, bool XH>
__global__ __launch_bounds__(256) void mmf16_kernel(const _Float16 *W, int N, int K, const void *Xv, int T, float *Y, int ldy, const float *resid, const float *bias,
                                                    float scale, int do_scale) {
    const float *X = reinterpret_cast<const float *>(Xv);
    const _Float16 *Xh = reinterpret_cast<const _Float16 *>(Xv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.x * (64 * WT) + (wave & 1) * (32 * WT), t0 = blockIdx.y * (64 * WT) + (wave >> 1) * (32 * WT);
    if (n0 >= N || t0 >= T) return;                             // (wave-uniform)
    const _Float16 *wr[WT], *xh[WT];
    const float *xr[WT];
#pragma unroll
    for (int i = 0; i < WT; i++) {
        const int n = n0 + 32 * i + m, t = t0 + 32 * i + m;     // rows past the end read the last row; their results are never stored
        wr[i] = W + (size_t)(n < N ? n : N - 1) * K + 8 * kg;
        xr[i] = X + (size_t)(t < T ? t : T - 1) * K + 8 * kg;
        xh[i] = Xh + (size_t)(t < T ? t : T - 1) * K + 8 * kg;
    }
    f32x16 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int j = 0; j < WT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;
#pragma unroll 2
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 a[WT], b[WT];
#pragma unroll
        for (int i = 0; i < WT; i++) {
            a[i] = *reinterpret_cast<const f16x8 *>(wr[i] + k0);
            if constexpr (XH) b[i] = *reinterpret_cast<const f16x8 *>(xh[i] + k0);
            else {
                const f32x4 x0 = *reinterpret_cast<const f32x4 *>(xr[i] + k0), x1 = *reinterpret_cast<const f32x4 *>(xr[i] + k0 + 4);
                b[i][0] = (_Float16)x0.x; b[i][1] = (_Float16)x0.y; b[i][2] = (_Float16)x0.z; b[i][3] = (_Float16)x0.w;
                b[i][4] = (_Float16)x1.x; b[i][5] = (_Float16)x1.y; b[i][6] = (_Float16)x1.z; b[i][7] = (_Float16)x1.w;
            }
        }
#pragma unroll
        for (int i = 0; i < WT; i++)
#pragma unroll
            for (int j = 0; j < WT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    // register r of a lane: weight row (r & 3) + 8 (r >> 2) + 4 kg of the tile, token = lane & 31: four consecutive rows per 16-byte store
#pragma unroll
    for (int i = 0; i < WT; i++)
#pragma unroll
        for (int j = 0; j < WT; j++) {
            const int t = t0 + 32 * j + m;
            if (t >= T) continue;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n = n0 + 32 * i + 8 * q + 4 * kg;
                float *dst = Y + (size_t)t * ldy + n;
                const float *rs = resid ? resid + (size_t)t * ldy + n : nullptr;
                float r[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    r[e] = acc[i][j][4 * q + e];
                    if (bias && n + e < N) r[e] = r[e] + bias[n + e];
                    if (do_scale) r[e] = r[e] * scale;
                }
                if (n + 3 < N && (ldy & 3) == 0) {
                    f32x4 v = {r[0], r[1], r[2], r[3]};
                    if (rs) { const f32x4 r4 = *reinterpret_cast<const f32x4 *>(rs); v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w; }
                    *reinterpret_cast<f32x4 *>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) if (n + e < N) dst[e] = rs ? rs[e] + r[e] : r[e];
                }
            }
        }
}

}  // namespace

bool mmf16_applicable(int type, int n_rows, int K, int T, const void *W, const void *x, const void *y) {
    return type == T_F16 && T >= 8 && n_rows >= 32 && (K % 16) == 0 && ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

namespace {
// The f16 x f16 product staged through LDS: 64 x 64 outputs per workgroup, 64 halves of k per stage (one 128-byte line of every row: eight lanes fetch a row's
// line together, where the direct kernel's lanes each fetch 16 bytes of a different row - 32 lines, on few L2 channels when the row stride is a power of two),
// double-buffered; a wave owns 32 x 32 outputs and reads its two fragments per matrix-core step from rows padded to 144 bytes (conflict-free 16-byte reads).
// The same chain of matrix-core steps over k as mmf16_kernel: the results agree bit for bit.
__global__ __launch_bounds__(256) void mmf16_lds_kernel(const _Float16 *__restrict__ W, int N, int K, const _Float16 *__restrict__ X, int T, float *Y, int ldy,
                                                        const float *resid, const float *bias, float scale, int do_scale) {
    constexpr int BK = 64, LDR = 72;                            // halves per stage, padded row length (144 bytes)
    __shared__ _Float16 sA[2][64 * LDR], sB[2][64 * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.x * 64, t0 = blockIdx.y * 64;
    const int lr = tid >> 3, lc = tid & 7;                      // loader: rows lr and lr + 32 of both tiles, 16-byte column lc
    const _Float16 *gw[2], *gx[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int n = n0 + lr + 32 * i, t = t0 + lr + 32 * i;   // rows past the end read the last row; their results are never stored
        gw[i] = W + (size_t)(n < N ? n : N - 1) * K + 8 * lc;
        gx[i] = X + (size_t)(t < T ? t : T - 1) * K + 8 * lc;
    }
    // PD stages of k are on their way at any time (registers), one is in LDS being used, the next is being written to the other LDS buffer: a stage's matrix
    // work is 4 steps (tens of ns) against a memory round trip of a microsecond, and the 577 x 1024 products have only 160 workgroups to overlap with
    constexpr int PD = 4;
    f16x8 ra[PD][2], rb[PD][2];
    // (no branch and no select around a load: either would make the compiler drain every load in flight where the paths join.  A piece past K - the last stage
    // of a K that is not a multiple of 64, and the stages requested past the end - is read from the row's last 16 bytes instead and never multiplied)
    auto fetch = [&](f16x8 (&a2)[2], f16x8 (&b2)[2], int k0) {
        const int k = k0 + 8 * lc < K ? k0 : K - 8 - 8 * lc;
#pragma unroll
        for (int i = 0; i < 2; i++) { a2[i] = *reinterpret_cast<const f16x8 *>(gw[i] + k); b2[i] = *reinterpret_cast<const f16x8 *>(gx[i] + k); }
    };
    auto put = [&](const f16x8 (&a2)[2], const f16x8 (&b2)[2], int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            *reinterpret_cast<f16x8 *>(&sA[buf][(lr + 32 * i) * LDR + 8 * lc]) = a2[i];
            *reinterpret_cast<f16x8 *>(&sB[buf][(lr + 32 * i) * LDR + 8 * lc]) = b2[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    const int ar = ((wave & 1) * 32 + m) * LDR + 8 * kg, br = ((wave >> 1) * 32 + m) * LDR + 8 * kg;
    const int nst = (K + BK - 1) / BK;
#pragma unroll
    for (int u = 0; u < PD; u++) fetch(ra[u], rb[u], u * BK);
    put(ra[0], rb[0], 0);
    __syncthreads();
    for (int st = 0; st < nst; st += PD) {
#pragma unroll
        for (int u = 0; u < PD; u++) {
            const int cur = st + u, buf = u & 1;
            if (cur >= nst) break;
            fetch(ra[u], rb[u], (cur + PD) * BK);               // slot u went to LDS a step ago: refill it PD stages ahead
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                if (cur * BK + 16 * s4 >= K) break;
                const f16x8 a = *reinterpret_cast<const f16x8 *>(&sA[buf][ar + 16 * s4]), b = *reinterpret_cast<const f16x8 *>(&sB[buf][br + 16 * s4]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
            if (cur + 1 < nst) put(ra[(u + 1) % PD], rb[(u + 1) % PD], buf ^ 1);
            __syncthreads();
        }
    }
    // register r of a lane: weight row (r & 3) + 8 (r >> 2) + 4 kg of the wave's tile, token = lane & 31 (as in mmf16_kernel)
    const int t = t0 + (wave >> 1) * 32 + m;
    if (t >= T) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int n = n0 + (wave & 1) * 32 + 8 * q + 4 * kg;
        float *dst = Y + (size_t)t * ldy + n;
        const float *rs = resid ? resid + (size_t)t * ldy + n : nullptr;
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            r[e] = acc[4 * q + e];
            if (bias && n + e < N) r[e] = r[e] + bias[n + e];
            if (do_scale) r[e] = r[e] * scale;
        }
        if (n + 3 < N && (ldy & 3) == 0) {
            f32x4 v = {r[0], r[1], r[2], r[3]};
            if (rs) { const f32x4 r4 = *reinterpret_cast<const f32x4 *>(rs); v.x = r4.x + v.x; v.y = r4.y + v.y; v.z = r4.z + v.z; v.w = r4.w + v.w; }
            *reinterpret_cast<f32x4 *>(dst) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) if (n + e < N) dst[e] = rs ? rs[e] + r[e] : r[e];
        }
    }
}

__global__ void f32_to_f16_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const f16x4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    reinterpret_cast<f16x4 *>(y)[i] = h;
}
}  // namespace

hipError_t launch_f32_to_f16(const float *x, void *y, size_t n, hipStream_t st) {
    if (n % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, x, reinterpret_cast<_Float16 *>(y), n / 4);
    return hipGetLastError();
}

template <bool XH>
static hipError_t launch_mmf16_t(const uint8_t *W, int n_rows, int K, const void *x, int T, float *y, int ld_out, const float *resid, const float *bias, float scale,
                                 bool do_scale, hipStream_t st) {
    const dim3 big((unsigned)((n_rows + 127) / 128), (unsigned)((T + 127) / 128));
    const char *sw = getenv("MI355_MMF16_TILE");                // "128": always the large tile (A/B and tests)
    if (big.x * big.y >= 256 || (sw && atoi(sw) == 128))
        hipLaunchKernelGGL((mmf16_kernel<2, XH>), big, dim3(256), 0, st, reinterpret_cast<const _Float16 *>(W), n_rows, K, x, T, y, ld_out, resid, bias, scale, (int)do_scale);
    else {
        const dim3 small((unsigned)((n_rows + 63) / 64), (unsigned)((T + 63) / 64));
        hipLaunchKernelGGL((mmf16_kernel<1, XH>), small, dim3(256), 0, st, reinterpret_cast<const _Float16 *>(W), n_rows, K, x, T, y, ld_out, resid, bias, scale, (int)do_scale);
    }
    return hipGetLastError();
}
hipError_t launch_mmf16(const uint8_t *W, int n_rows, int K, const float *x, int T, float *y, int ld_out, const float *resid, hipStream_t st) {
    return launch_mmf16_t<false>(W, n_rows, K, x, T, y, ld_out, resid, nullptr, 1.0f, false, st);
}
// the same product with the activation rows already rounded to f16 (launch_f32_to_f16): [T][K] halves; y = resid + (W x + bias) * scale, every part optional
hipError_t launch_mmf16_xh(const uint8_t *W, int n_rows, int K, const void *xh, int T, float *y, int ld_out, const float *resid, const float *bias, float scale, bool do_scale,
                           hipStream_t st) {
    const char *sw = getenv("MI355_MMF16_LDS");                 // "0": the direct-from-global kernel (A/B and tests)
    if (!(sw && atoi(sw) == 0) && (K % 8) == 0) {
        hipLaunchKernelGGL(mmf16_lds_kernel, dim3((unsigned)((n_rows + 63) / 64), (unsigned)((T + 63) / 64)), dim3(256), 0, st, reinterpret_cast<const _Float16 *>(W), n_rows, K,
                           reinterpret_cast<const _Float16 *>(xh), T, y, ld_out, resid, bias, scale, (int)do_scale);
        return hipGetLastError();
    }
    return launch_mmf16_t<true>(W, n_rows, K, xh, T, y, ld_out, resid, bias, scale, do_scale, st);
}

}  // namespace mi355
