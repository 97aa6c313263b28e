// act.hip — activation-side kernels: RMSNorm(+weight) fused with Q8_K / Q8_0 quantisation, plain
// quantisation, SwiGLU, residual add, row softmax.
//
// Stand in for ggml_rms_norm + ggml_mul, quantize_row_q8_K / quantize_row_q8_0 (the CPU backend's
// activation formats), ggml_silu*ggml_mul, ggml_add, ggml_soft_max_ext — all
// reached from the reference through llama_decode (src/llama_server_context.cc:1635); rows a9, a12, a14,
// a17 of SURVEY.md §8a.  Quantised codes and scales are bit-identical to the CPU restatement; only the
// f64 sum-of-squares is tree-ordered instead of sequential.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

// One workgroup per row.  do_norm: y = (x * rsqrt(mean(x^2)+eps)) * w, else y = x.
__global__ __launch_bounds__(256) void norm_quant_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                         int n, float eps, int do_norm, float *__restrict__ yf,
                                                         ActQuant q, int want_q8k, int want_q80, int8_t *__restrict__ bh, int8_t *__restrict__ bl) {
    __shared__ double red[4];
    // grid = (rows, splits): each workgroup re-derives the row scale (cheap, L2-resident) and quantises its share of blocks
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *xr = x + (size_t)row * n;
    float scale = 1.0f;
    if (do_norm) {
        double s = 0.0;
        for (int i = tid * 4; i < n; i += 256 * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(xr + i);
            s += (double)(v.x * v.x); s += (double)(v.y * v.y); s += (double)(v.z * v.z); s += (double)(v.w * v.w);
        }
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const double tot = red[0] + red[1] + red[2] + red[3];
        const float mean = (float)(tot / (double)n);
        scale = 1.0f / sqrtf(mean + eps);
    }
    const int nblk = n >> 8;
    for (int b = blockIdx.y * 4 + wave; b < nblk; b += 4 * gridDim.y) {
        const int e0 = b * 256 + lane * 4;
        float4 v = *reinterpret_cast<const float4 *>(xr + e0);
        if (do_norm) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + e0);
            v.x = (v.x * scale) * ww.x; v.y = (v.y * scale) * ww.y; v.z = (v.z * scale) * ww.z; v.w = (v.w * scale) * ww.w;
        }
        if (yf) *reinterpret_cast<float4 *>(yf + (size_t)row * n + e0) = v;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if (want_q8k) {
            uint32_t packed; int bs; float dq;
            wave_quant_q8k(vv, lane, packed, bs, dq);
            *reinterpret_cast<uint32_t *>(q.qs + (size_t)row * n + e0) = packed;
            if ((lane & 3) == 0) {
                const size_t bi = (size_t)row * (n >> 4) + b * 16 + (lane >> 2);
                q.bsums[bi] = (int16_t)bs;
                if (bh) { const int hi = bs >> 6; bh[bi] = (int8_t)hi; bl[bi] = (int8_t)(bs - 64 * hi); }   // mmq_prep_kernel's split, for the MFMA kernels
            }
            if (lane == 0) q.d[(size_t)row * nblk + b] = dq;
        }
        if (want_q80) {
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(q.qs0 + (size_t)row * n + e0) = packed;
            if ((lane & 7) == 0) q.d0[(size_t)row * (n >> 5) + b * 8 + (lane >> 3)] = f2h(d);
        }
    }
}

// Prompt batches (hundreds of rows): ONE wave per 256-block, up to 16 waves per workgroup, the block in registers from the first load to the codes - the kernel above
// reads a row twice (scale, then quantise) and walks a wave through its blocks one after the other, 7.8 us per launch at 512 x 4096 where the data moves in 3.
// do_norm: the row's sum of squares must come out with the bits of the kernel above (whose thread tid adds the squares of float4 tid, tid + 256, tid + 512 ..
// to ONE double, element by element): every thread leaves its products in LDS and the first 256 threads add them up in exactly that order, then the same wave sums
// and the same four partial sums.  grid = (rows, parts): a part is 16 blocks of the row; do_norm needs the whole row in one workgroup (n <= 16 * 256 * NV).
template <int NV>
__global__ __launch_bounds__(1024) void norm_quant_wide_kernel(const float *__restrict__ x, const float *__restrict__ w, int n, float eps, int do_norm,
                                                                float *__restrict__ yf, ActQuant q, int want_q8k, int want_q80, int8_t *__restrict__ bh, int8_t *__restrict__ bl) {
    extern __shared__ float prod[];                       // do_norm: [n] squares, float4 index major
    __shared__ double red[4];
    __shared__ float s_scale;
    const int row = blockIdx.x, u = threadIdx.x, lane = u & 63, wave = u >> 6, nw = blockDim.x >> 6;
    const int nblk = n >> 8;
    const float *xr = x + (size_t)row * n;
    float4 v[NV];
    int blk[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) {
        blk[j] = (do_norm ? 0 : blockIdx.y * 16 * NV) + wave + nw * j;
        const int bc = blk[j] < nblk ? blk[j] : nblk - 1;
        v[j] = *reinterpret_cast<const float4 *>(xr + bc * 256 + lane * 4);
    }
    float scale = 1.0f;
    if (do_norm) {
#pragma unroll
        for (int j = 0; j < NV; j++)
            if (blk[j] < nblk) *reinterpret_cast<float4 *>(prod + ((size_t)blk[j] * 64 + lane) * 4) = float4{v[j].x * v[j].x, v[j].y * v[j].y, v[j].z * v[j].z, v[j].w * v[j].w};
        __syncthreads();
        if (u < 256) {
            double s = 0.0;
            for (int i4 = u; i4 < (n >> 2); i4 += 256) {
                const float4 p = *reinterpret_cast<const float4 *>(prod + (size_t)i4 * 4);
                s += (double)p.x; s += (double)p.y; s += (double)p.z; s += (double)p.w;
            }
            s = wave_sum(s);
            if (lane == 0) red[wave] = s;
        }
        __syncthreads();
        if (u == 0) {
            const double tot = red[0] + red[1] + red[2] + red[3];
            const float mean = (float)(tot / (double)n);
            s_scale = 1.0f / sqrtf(mean + eps);
        }
        __syncthreads();
        scale = s_scale;
    }
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const int b = blk[j];
        if (b >= nblk) continue;                           // wave-uniform
        const int e0 = b * 256 + lane * 4;
        float4 t = v[j];
        if (do_norm) {
            const float4 ww = *reinterpret_cast<const float4 *>(w + e0);
            t.x = (t.x * scale) * ww.x; t.y = (t.y * scale) * ww.y; t.z = (t.z * scale) * ww.z; t.w = (t.w * scale) * ww.w;
        }
        if (yf) *reinterpret_cast<float4 *>(yf + (size_t)row * n + e0) = t;
        const float vv[4] = {t.x, t.y, t.z, t.w};
        if (want_q8k) {
            uint32_t packed; int bs; float dq;
            wave_quant_q8k(vv, lane, packed, bs, dq);
            *reinterpret_cast<uint32_t *>(q.qs + (size_t)row * n + e0) = packed;
            if ((lane & 3) == 0) {
                const size_t bi = (size_t)row * (n >> 4) + b * 16 + (lane >> 2);
                q.bsums[bi] = (int16_t)bs;
                if (bh) { const int hi = bs >> 6; bh[bi] = (int8_t)hi; bl[bi] = (int8_t)(bs - 64 * hi); }
            }
            if (lane == 0) q.d[(size_t)row * nblk + b] = dq;
        }
        if (want_q80) {
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(q.qs0 + (size_t)row * n + e0) = packed;
            if ((lane & 7) == 0) q.d0[(size_t)row * (n >> 5) + b * 8 + (lane >> 3)] = f2h(d);
        }
    }
}
// true: launched.  Rows of a prompt batch only (T >= 32); with the norm the row must fit one workgroup
static bool launch_wide(const float *x, const float *w, int n, int T, float eps, int do_norm, float *yf, const ActQuant &qq, int k, int z, int8_t *bh, int8_t *bl, hipStream_t st) {
    static const bool off = getenv("MI355_QUANT_WIDE") && getenv("MI355_QUANT_WIDE")[0] == '0';
    const int nblk = n >> 8;
    if (off || T < 32 || (n & 255) || nblk < 1) return false;
    if (do_norm) {
        if (nblk > 32 || (n & 1023)) return false;
        const int nv = nblk > 16 ? 2 : 1, nwv = (nblk + nv - 1) / nv;
        const size_t lds = (size_t)n * sizeof(float);
        if (nv == 1) hipLaunchKernelGGL(norm_quant_wide_kernel<1>, dim3(T, 1), dim3(64 * nwv), lds, st, x, w, n, eps, 1, yf, qq, k, z, bh, bl);
        else hipLaunchKernelGGL(norm_quant_wide_kernel<2>, dim3(T, 1), dim3(64 * nwv), lds, st, x, w, n, eps, 1, yf, qq, k, z, bh, bl);
        return true;
    }
    const int parts = (nblk + 15) / 16, nwv = nblk < 16 ? nblk : 16;
    hipLaunchKernelGGL(norm_quant_wide_kernel<1>, dim3(T, parts), dim3(64 * nwv), 0, st, x, w, n, eps, 0, yf, qq, k, z, bh, bl);
    return true;
}

// workgroups per row: a wave quantises the blocks b, b + 4 * splits, .. of its row, one after the other - a chain of wave-level steps per block that waits on
// itself - so a row of 16 .. 56 blocks is cut until a wave holds one or two of them (the row scale is re-derived per workgroup: the row sits in L2)
#ifndef MI355_QUANT_SPLIT_BLOCKS
#define MI355_QUANT_SPLIT_BLOCKS 8
#endif
static int quant_splits(int T, int n) {
    if (T < 8) return ((n >> 8) + 3) / 4;
    const int nblk = n >> 8;
    int s = (nblk + MI355_QUANT_SPLIT_BLOCKS - 1) / MI355_QUANT_SPLIT_BLOCKS;      // blocks per workgroup <= 8 (two per wave)
    if (T < 64 && s < 2) s = 2;
    while (s > 1 && (long long)T * s > 16384) s >>= 1;
    return s < 1 ? 1 : s;
}

hipError_t launch_rmsnorm_quant(const float *x, const float *w, int n, int T, float eps, float *y_f32,
                                const ActQuant *q, bool want_q8k, bool want_q80, hipStream_t st, int8_t *bh, int8_t *bl) {
    ActQuant qq;
    if (q) qq = *q;
    if (launch_wide(x, w, n, T, eps, 1, y_f32, qq, (int)(q && want_q8k), (int)(q && want_q80), (q && want_q8k) ? bh : nullptr, bl, st)) return hipGetLastError();
    const int splits = quant_splits(T, n);
    hipLaunchKernelGGL(norm_quant_kernel, dim3(T, splits), dim3(256), 0, st, x, w, n, eps, 1, y_f32, qq,
                       (int)(q && want_q8k), (int)(q && want_q80), (q && want_q8k) ? bh : nullptr, bl);
    return hipGetLastError();
}

hipError_t launch_quantize(const float *x, int n, int T, const ActQuant &q, bool want_q8k, bool want_q80, hipStream_t st, int8_t *bh, int8_t *bl) {
    if (launch_wide(x, nullptr, n, T, 0.0f, 0, nullptr, q, (int)want_q8k, (int)want_q80, want_q8k ? bh : nullptr, bl, st)) return hipGetLastError();
    const int splits = quant_splits(T, n);
    hipLaunchKernelGGL(norm_quant_kernel, dim3(T, splits), dim3(256), 0, st, x, (const float *)nullptr, n, 0.0f, 0,
                       (float *)nullptr, q, (int)want_q8k, (int)want_q80, want_q8k ? bh : nullptr, bl);
    return hipGetLastError();
}

// SwiGLU and the quantisation of its result for the down projection in one pass (prompt batches): y = silu(g) * u is
// quantised block by block as norm_quant_kernel does and never written as f32 — the arithmetic of swiglu_kernel followed by
// the quantiser, so the blocks are bit-identical to the two launches it replaces.
__global__ __launch_bounds__(256) void swiglu_quant_kernel(const float *__restrict__ g, const float *__restrict__ u, int n, ActQuant q,
                                                           int want_q8k, int want_q80, int8_t *__restrict__ bh, int8_t *__restrict__ bl) {
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = n >> 8;
    for (int b = blockIdx.y * 4 + wave; b < nblk; b += 4 * gridDim.y) {
        const int e0 = b * 256 + lane * 4;
        const float4 a = *reinterpret_cast<const float4 *>(g + (size_t)row * n + e0), c = *reinterpret_cast<const float4 *>(u + (size_t)row * n + e0);
        const float vv[4] = {(a.x / (1.0f + expf(-a.x))) * c.x, (a.y / (1.0f + expf(-a.y))) * c.y, (a.z / (1.0f + expf(-a.z))) * c.z,
                             (a.w / (1.0f + expf(-a.w))) * c.w};
        if (want_q8k) {
            uint32_t packed; int bs; float dq;
            wave_quant_q8k(vv, lane, packed, bs, dq);
            *reinterpret_cast<uint32_t *>(q.qs + (size_t)row * n + e0) = packed;
            if ((lane & 3) == 0) {
                const size_t bi = (size_t)row * (n >> 4) + b * 16 + (lane >> 2);
                q.bsums[bi] = (int16_t)bs;
                if (bh) { const int hi = bs >> 6; bh[bi] = (int8_t)hi; bl[bi] = (int8_t)(bs - 64 * hi); }
            }
            if (lane == 0) q.d[(size_t)row * nblk + b] = dq;
        }
        if (want_q80) {
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(q.qs0 + (size_t)row * n + e0) = packed;
            if ((lane & 7) == 0) q.d0[(size_t)row * (n >> 5) + b * 8 + (lane >> 3)] = f2h(d);
        }
    }
}
hipError_t launch_swiglu_quant(const float *g, const float *u, int n, int T, const ActQuant &q, bool want_q8k, bool want_q80, hipStream_t st,
                               int8_t *bh, int8_t *bl) {
    if ((n % 256) != 0 || T <= 0) return hipErrorInvalidValue;
    const int splits = quant_splits(T, n);
    hipLaunchKernelGGL(swiglu_quant_kernel, dim3(T, splits), dim3(256), 0, st, g, u, n, q, (int)want_q8k, (int)want_q80, want_q8k ? bh : nullptr, bl);
    return hipGetLastError();
}

// ---------------------------------------------------------------- elementwise
__global__ void swiglu_kernel(const float *g, const float *u, float *y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float a = g[i];
        y[i] = (a / (1.0f + expf(-a))) * u[i];
    }
}
hipError_t launch_swiglu(const float *g, const float *u, float *y, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, u, y, n);
    return hipGetLastError();
}
__global__ void add_kernel(const float *a, const float *b, float *y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
hipError_t launch_add(const float *a, const float *b, float *y, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, y, n);
    return hipGetLastError();
}

// the Q / K / V bias vectors of a qwen2-style attention block added to all T rows of the three projections in one launch (a null bias skips its segment);
// grid.y = 0, 1, 2 picks the projection
__global__ void add_qkv_bias_kernel(float *q, float *k, float *v, const float *bq, const float *bk, const float *bv, int nq, int nkv, int T) {
    const int sg = blockIdx.y;
    float *x = sg == 0 ? q : sg == 1 ? k : v;
    const float *b = sg == 0 ? bq : sg == 1 ? bk : bv;
    const int n = sg == 0 ? nq : nkv;
    if (!b) return;
    const int64_t tot = (int64_t)n * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) x[i] += b[i % n];
}
hipError_t launch_add_qkv_bias(float *q, float *k, float *v, const float *bq, const float *bk, const float *bv, int nq, int nkv, int T, hipStream_t st) {
    const int64_t tot = (int64_t)(nq > nkv ? nq : nkv) * T;
    const unsigned gx = (unsigned)std::min<int64_t>((tot + 255) / 256, 2048);
    hipLaunchKernelGGL(add_qkv_bias_kernel, dim3(gx, 3), dim3(256), 0, st, q, k, v, bq, bk, bv, nq, nkv, T);
    return hipGetLastError();
}

// LayerNorm with weight and bias (ggml_norm + ggml_mul + ggml_add, build_norm(LLM_NORM): encoder models), one workgroup per row: mean and variance in double
// (the CPU sums them in double, sequentially; here tree-ordered), y = (x - mean) * rsqrt(var + eps) * w + b with the CPU's operation order.  In place allowed.
// yh (optional): the same row rounded to f16, for a consumer that reads it that way (the f16 GEMM of the LLaVA tower).
__global__ __launch_bounds__(256) void layer_norm_kernel(const float *x, const float *w, const float *b, int n, float eps, float *y, _Float16 *yh) {
    __shared__ double red[4];
    const float *xr = x + (size_t)blockIdx.x * n;
    float *yr = y + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)xr[i];
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
    __syncthreads();
    double s2 = 0.0;
    for (int i = tid; i < n; i += 256) { const float v = xr[i] - mean; s2 += (double)(v * v); }
    s2 = wave_sum(s2);
    if (lane == 0) red[wave] = s2;
    __syncthreads();
    const float variance = (float)((red[0] + red[1] + red[2] + red[3]) / (double)n);
    const float scale = 1.0f / sqrtf(variance + eps);
    for (int i = tid; i < n; i += 256) {
        const float v = ((xr[i] - mean) * scale) * w[i] + b[i];
        yr[i] = v;
        if (yh) yh[(size_t)blockIdx.x * n + i] = (_Float16)v;
    }
}
hipError_t launch_layer_norm(const float *x, const float *w, const float *b, int n, int T, float eps, float *y, hipStream_t st) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)T), dim3(256), 0, st, x, w, b, n, eps, y, (_Float16 *)nullptr);
    return hipGetLastError();
}
hipError_t launch_layer_norm_h(const float *x, const float *w, const float *b, int n, int T, float eps, float *y, void *yh, hipStream_t st) {
    hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)T), dim3(256), 0, st, x, w, b, n, eps, y, reinterpret_cast<_Float16 *>(yh));
    return hipGetLastError();
}

// row softmax: y = softmax(x*scale + mask)   (ggml_soft_max_ext, max_bias = 0)
__global__ __launch_bounds__(256) void soft_max_kernel(const float *x, const float *mask, float *y, int n, float scale) {
    __shared__ float redf[4];
    __shared__ double redd[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *xr = x + (size_t)row * n;
    const float *mr = mask ? mask + (size_t)row * n : nullptr;
    float *yr = y + (size_t)row * n;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, xr[i] * scale + (mr ? mr[i] : 0.0f));
    mx = wave_max(mx);
    if (lane == 0) redf[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    double s = 0.0;
    for (int i = tid; i < n; i += 256) {
        const float e = expf((xr[i] * scale + (mr ? mr[i] : 0.0f)) - mx);
        yr[i] = e;
        s += (double)e;
    }
    s = wave_sum(s);
    if (lane == 0) redd[wave] = s;
    __syncthreads();
    const float inv = (float)(1.0 / (redd[0] + redd[1] + redd[2] + redd[3]));
    for (int i = tid; i < n; i += 256) yr[i] *= inv;
}
hipError_t launch_soft_max(const float *x, const float *mask, float *y, int n, int rows, float scale, hipStream_t st) {
    hipLaunchKernelGGL(soft_max_kernel, dim3(rows), dim3(256), 0, st, x, mask, y, n, scale);
    return hipGetLastError();
}

// ---------------------------------------------------------------- planes -> ggml blocks (tests only)
__global__ void pack_q8k_kernel(ActQuant q, int n, uint8_t *blocks) {
    const int row = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;   // 256 threads
    const int nblk = n >> 8;
    uint8_t *dst = blocks + ((size_t)row * nblk + b) * 292;
    if (tid == 0) *reinterpret_cast<float *>(dst) = q.d[(size_t)row * nblk + b];
    dst[4 + tid] = (uint8_t)q.qs[(size_t)row * n + b * 256 + tid];
    if (tid < 16) {
        const int16_t v = q.bsums[(size_t)row * (n >> 4) + b * 16 + tid];
        dst[260 + 2 * tid] = (uint8_t)(v & 0xff);
        dst[261 + 2 * tid] = (uint8_t)((v >> 8) & 0xff);
    }
}
hipError_t launch_pack_q8k_blocks(const ActQuant &q, int n, int T, uint8_t *blocks, hipStream_t st) {
    hipLaunchKernelGGL(pack_q8k_kernel, dim3(n >> 8, T), dim3(256), 0, st, q, n, blocks);
    return hipGetLastError();
}
__global__ void pack_q80_kernel(ActQuant q, int n, uint8_t *blocks) {
    const int row = blockIdx.y, b = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;  // 256 threads = 8 blocks
    const int nblk = n >> 5;
    if (b >= nblk) return;
    uint8_t *dst = blocks + ((size_t)row * nblk + b) * 34;
    if (j == 0) {
        const uint16_t d = q.d0[(size_t)row * nblk + b];
        dst[0] = (uint8_t)(d & 0xff);
        dst[1] = (uint8_t)(d >> 8);
    }
    dst[2 + j] = (uint8_t)q.qs0[(size_t)row * n + b * 32 + j];
}
hipError_t launch_pack_q80_blocks(const ActQuant &q, int n, int T, uint8_t *blocks, hipStream_t st) {
    hipLaunchKernelGGL(pack_q80_kernel, dim3(((n >> 5) + 7) / 8, T), dim3(256), 0, st, q, n, blocks);
    return hipGetLastError();
}

}  // namespace mi355
