// clip.hip — the small kernels of the LLaVA image encoder (CLIP ViT tower + MLP projector) beside the f16 matrix-core GEMM of mmf.hip and the LayerNorm of
// misc.hip: patch gathering for the convolution-as-GEMM, [class ; patches] + position embeddings, bias (+ scale), the tower's full (unmasked) attention in f32,
// GELU / quick-GELU with ggml's f16-table semantics.
//
// Stands in for the graph clip_image_encode builds (llama.cpp examples/llava/clip.cpp, reached from the reference through
// llava_image_embed_make_with_clip_img, /root/reference/src/llama_server_context.cc:820) — SURVEY.md §8 row f4.  One image is 577 rows of 1024, once per picture
// (five times with a LLaVA-1.6 image grid): small launches; the attention runs on the f32 matrix cores (clip_attn_mfma_kernel; LDS-tiled VALU kernel for other head sizes), the rest is written for clarity.
#include <cstdlib>

#include "kernels.h"
#include "dev_common.h"

namespace mi355 {

// patches[p][(c * P + ky) * P + kx] = img[c][py * P + ky][px * P + kx]; columns [3 P P, ld) are zero (the GEMM wants K % 16 == 0)
__global__ void clip_im2col_kernel(const float *__restrict__ img, int S, int P, int ld, float *__restrict__ patches) {
    const int G = S / P, p = blockIdx.x, py = p / G, px = p - py * G, KP = 3 * P * P;
    for (int j = threadIdx.x; j < ld; j += blockDim.x) {
        float v = 0.0f;
        if (j < KP) {
            const int c = j / (P * P), r = j - c * P * P, ky = r / P, kx = r - ky * P;
            v = img[(size_t)c * S * S + (size_t)(py * P + ky) * S + (px * P + kx)];
        }
        patches[(size_t)p * ld + j] = v;
    }
}
hipError_t launch_clip_im2col(const float *img, int S, int P, int ld, float *patches, hipStream_t st) {
    const int G = S / P;
    hipLaunchKernelGGL(clip_im2col_kernel, dim3(G * G), dim3(256), 0, st, img, S, P, ld, patches);
    return hipGetLastError();
}

// emb[0] = class + pos[0]; emb[t] = patch[t - 1] + pos[t]
__global__ void clip_embed_kernel(const float *__restrict__ patch, const float *__restrict__ cls, const float *__restrict__ pos, int E, float *__restrict__ emb) {
    const int t = blockIdx.x;
    for (int i = threadIdx.x; i < E; i += blockDim.x) {
        const float base = t == 0 ? cls[i] : patch[(size_t)(t - 1) * E + i];
        emb[(size_t)t * E + i] = base + pos[(size_t)t * E + i];
    }
}
hipError_t launch_clip_embed(const float *patch, const float *cls, const float *pos, int E, int T, float *emb, hipStream_t st) {
    hipLaunchKernelGGL(clip_embed_kernel, dim3(T), dim3(256), 0, st, patch, cls, pos, E, emb);
    return hipGetLastError();
}

// x[t][i] = (x[t][i] + b[i]) * scale   (scale applied as its own multiplication, as ggml_scale after ggml_add; 1: none)
__global__ void clip_bias_kernel(float *__restrict__ x, const float *__restrict__ b, int n, size_t total, float scale, int do_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float v = x[i] + b[i % (size_t)n];
    if (do_scale) v = v * scale;
    x[i] = v;
}
hipError_t launch_clip_bias(float *x, const float *b, int n, int T, float scale, bool do_scale, hipStream_t st) {
    const size_t total = (size_t)n * T;
    hipLaunchKernelGGL(clip_bias_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, b, n, total, scale, (int)do_scale);
    return hipGetLastError();
}

// full attention of the tower: every row sees every row.  One wave per (head, query row): scores of all keys (lanes stride the keys), softmax, then lane = output
// element (D <= 64) or pairs of them (D = 128).  q arrives scaled.
template <int D>
__global__ __launch_bounds__(256) void clip_attn_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, int T, int H,
                                                        float *__restrict__ out) {
    extern __shared__ float sm[];                       // [4 waves][T] probabilities
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int h = blockIdx.y, tq = blockIdx.x * 4 + wave;
    const int E = H * D;
    float *pr = sm + (size_t)wave * T;
    if (tq < T) {
        const float *qr = q + (size_t)tq * E + (size_t)h * D;
        float mx = -INFINITY;
        for (int tk = lane; tk < T; tk += 64) {
            const float *kr = k + (size_t)tk * E + (size_t)h * D;
            float s = 0.0f;
#pragma unroll 8
            for (int d = 0; d < D; d++) s += kr[d] * qr[d];
            pr[tk] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int tk = lane; tk < T; tk += 64) { const float e = expf(pr[tk] - mx); pr[tk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int tk = lane; tk < T; tk += 64) pr[tk] *= inv;
    }
    __syncthreads();
    if (tq < T) {
        for (int d = lane; d < D; d += 64) {
            float acc = 0.0f;
            for (int tk = 0; tk < T; tk++) acc += v[(size_t)tk * E + (size_t)h * D + d] * pr[tk];
            out[(size_t)tq * E + (size_t)h * D + d] = acc;
        }
    }
}
// The same attention, QT = 40 queries of one head per workgroup (15 x 16 workgroups for the 577 rows of ViT-L/14-336: one round of the 256 CUs): Q tile, a
// 64-key chunk of K (then of V) and the QT x T probabilities live in LDS; a thread owns one key (scores) or one output element (P.V) for QT / 4 (QT D / 256)
// queries, so an LDS operand feeds several sums; the LDS reads of step i + 1 are issued before the arithmetic of step i.  Every sum runs over the same index in
// the same order as in clip_attn_kernel (d ascending for a score, the keys ascending for an output; the soft-max is the same code): the two agree bit for bit.
template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void clip_attn_tiled_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                                                          const float *__restrict__ v, int T, int H, float *__restrict__ out,
                                                                                                          _Float16 *__restrict__ outh) {
    constexpr int QT = 40, KC = 64, LDK = D + 4, D4 = D / 4, SQ = QT / 4;
    constexpr int NQ = QT * D / 256, QSTEP = 256 / D;           // P.V: a thread's queries are qb + QSTEP * j, j < NQ, all for output element d
    static_assert(256 % D == 0 && (QT * D) % 256 == 0, "head size");
    extern __shared__ float sm[];
    const int Tp = (T + KC - 1) / KC * KC;
    float *sQ = sm, *sKV = sQ + QT * D, *sS = sKV + KC * LDK;   // [QT][D], [KC][LDK], [QT][Tp]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, q0 = blockIdx.x * QT, E = H * D;
    {                                                           // blockIdx.z: one of several images' rows, T apart
        const size_t img = (size_t)blockIdx.z * T * E;
        q += img; k += img; v += img; out += img;
        if (outh) outh += img;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int i = tid; i < QT * D4; i += 256) {
        const int r = i / D4, c4 = i - r * D4, tq = q0 + r < T ? q0 + r : T - 1;          // (rows past the end repeat the last one; never stored)
        *reinterpret_cast<f4 *>(sQ + r * D + 4 * c4) = *reinterpret_cast<const f4 *>(q + (size_t)tq * E + (size_t)h * D + 4 * c4);
    }
    // a chunk of K (then V) travels global -> registers -> LDS; the loads of chunk c + 1 are in flight while chunk c is being used
    constexpr int NST = KC * D4 / 256;                          // 16-byte pieces per thread and chunk
    f4 pre[NST];
    auto fetch = [&](const float *src, int c) {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const int i = tid + 256 * s, r = i / D4, c4 = i - r * D4, tk = c * KC + r;
            f4 val = {0.0f, 0.0f, 0.0f, 0.0f};                  // (rows past T are zero)
            if (tk < T) val = *reinterpret_cast<const f4 *>(src + (size_t)tk * E + (size_t)h * D + 4 * c4);
            pre[s] = val;
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const int i = tid + 256 * s, r = i / D4, c4 = i - r * D4;
            *reinterpret_cast<f4 *>(sKV + r * LDK + 4 * c4) = pre[s];
        }
    };
    const int NC = Tp / KC;
    fetch(k, 0);
    for (int c = 0; c < NC; c++) {
        __syncthreads();
        put();
        __syncthreads();
        if (c + 1 < NC) fetch(k, c + 1); else fetch(v, 0);
        float acc[SQ];
#pragma unroll
        for (int j = 0; j < SQ; j++) acc[j] = 0.0f;
        const float *kr = sKV + lane * LDK, *qr = sQ + wave * D;
        f4 kv = *reinterpret_cast<const f4 *>(kr), qv[SQ];
#pragma unroll
        for (int j = 0; j < SQ; j++) qv[j] = *reinterpret_cast<const f4 *>(qr + 4 * j * D);
        for (int d = 0; d < D; d += 4) {
            const int dn = d + 4 < D ? d + 4 : d;               // (the last step re-reads itself: no branch in the pipeline)
            const f4 kn = *reinterpret_cast<const f4 *>(kr + dn);
            f4 qn[SQ];
#pragma unroll
            for (int j = 0; j < SQ; j++) qn[j] = *reinterpret_cast<const f4 *>(qr + 4 * j * D + dn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SQ; j++) { acc[j] += kv.x * qv[j].x; acc[j] += kv.y * qv[j].y; acc[j] += kv.z * qv[j].z; acc[j] += kv.w * qv[j].w; }
            __builtin_amdgcn_sched_barrier(0);
            kv = kn;
#pragma unroll
            for (int j = 0; j < SQ; j++) qv[j] = qn[j];
        }
        const int tk = c * KC + lane;
#pragma unroll
        for (int j = 0; j < SQ; j++) sS[(wave + 4 * j) * Tp + tk] = acc[j];
    }
    __syncthreads();
    for (int j = 0; j < SQ; j++) {                              // soft-max of row wave + 4 j, as clip_attn_kernel does it
        float *pr = sS + (wave + 4 * j) * Tp;
        float mx = -INFINITY;
        for (int tk = lane; tk < T; tk += 64) mx = fmaxf(mx, pr[tk]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int tk = lane; tk < T; tk += 64) { const float e = expf(pr[tk] - mx); pr[tk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int tk = lane; tk < T; tk += 64) pr[tk] *= inv;
        for (int tk = T + lane; tk < Tp; tk += 64) pr[tk] = 0.0f;
    }
    const int d = tid % D, qb = tid / D;
    float o[NQ];
#pragma unroll
    for (int j = 0; j < NQ; j++) o[j] = 0.0f;
    for (int c = 0; c < NC; c++) {
        __syncthreads();
        put();
        __syncthreads();
        if (c + 1 < NC) fetch(v, c + 1);
        const int nk = T - c * KC < KC ? T - c * KC : KC;       // (keys past T hold zeros on both sides; skipping them keeps the sums those of the short loop)
        const int nk4 = nk & ~3;
        const float *vr = sKV + d, *pr = sS + qb * Tp + c * KC;
        float vv[4];
        f4 p[NQ];
        if (nk4 > 0) {
#pragma unroll
            for (int e = 0; e < 4; e++) vv[e] = vr[e * LDK];
#pragma unroll
            for (int j = 0; j < NQ; j++) p[j] = *reinterpret_cast<const f4 *>(pr + QSTEP * j * Tp);
        }
        for (int tk = 0; tk < nk4; tk += 4) {
            const int tn = tk + 4 < nk4 ? tk + 4 : tk;
            float vn[4];
            f4 pn[NQ];
#pragma unroll
            for (int e = 0; e < 4; e++) vn[e] = vr[(tn + e) * LDK];
#pragma unroll
            for (int j = 0; j < NQ; j++) pn[j] = *reinterpret_cast<const f4 *>(pr + QSTEP * j * Tp + tn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NQ; j++) { o[j] += vv[0] * p[j].x; o[j] += vv[1] * p[j].y; o[j] += vv[2] * p[j].z; o[j] += vv[3] * p[j].w; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; e++) vv[e] = vn[e];
#pragma unroll
            for (int j = 0; j < NQ; j++) p[j] = pn[j];
        }
        for (int tk = nk4; tk < nk; tk++) {                      // (the last chunk's odd keys)
            const float v0 = vr[tk * LDK];
#pragma unroll
            for (int j = 0; j < NQ; j++) o[j] += v0 * pr[QSTEP * j * Tp + tk];
        }
    }
#pragma unroll
    for (int j = 0; j < NQ; j++) {
        const int tq = q0 + qb + QSTEP * j;
        if (tq < T) {
            out[(size_t)tq * E + (size_t)h * D + d] = o[j];
            if (outh) outh[(size_t)tq * E + (size_t)h * D + d] = (_Float16)o[j];
        }
    }
}
// The tower's attention on the f32 matrix cores (head size 64): 32 queries of one head per workgroup; S = Q K^T as 32 x 32 tiles of v_mfma_f32_32x32x2_f32 (a
// wave per 32 keys of a 128-key chunk), the soft-max of clip_attn_kernel on the rows in LDS, O = P V as two 32 x 32 tiles per half of a chunk's keys (four waves:
// output half x key half; the two key halves are added at the end).  All operands are f32 and every product is exact in the f32 accumulators' sense; what
// differs from the other two kernels is the order of the f32 additions inside a dot product (the matrix core's, and the final pair) - a few 1e-7 relative.
__global__ __launch_bounds__(256) void clip_attn_mfma_kernel(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v, int T, int H,
                                                             float *__restrict__ out, _Float16 *__restrict__ outh) {
    constexpr int D = 64, QT = 32, KC = 128, LD = D + 1, D4 = D / 4;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    extern __shared__ float sm[];
    const int Tp = (T + KC - 1) / KC * KC, LS = Tp + 1;         // (odd row length: lanes that read a column of P hit 32 banks)
    float *sQ = sm, *sKV = sQ + QT * LD, *sS = sKV + KC * LD;   // [QT][LD], [KC][LD], [QT][LS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, q0 = blockIdx.x * QT, E = H * D;
    {
        const size_t img = (size_t)blockIdx.z * T * E;
        q += img; k += img; v += img; out += img;
        if (outh) outh += img;
    }
    for (int i = tid; i < QT * D4; i += 256) {
        const int r = i / D4, c4 = i - r * D4, tq = q0 + r < T ? q0 + r : T - 1;
        const f4 val = *reinterpret_cast<const f4 *>(q + (size_t)tq * E + (size_t)h * D + 4 * c4);
        float *d = sQ + r * LD + 4 * c4;
        d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    constexpr int NST = KC * D4 / 256;
    f4 pre[NST];
    auto fetch = [&](const float *src, int c) {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const int i = tid + 256 * s, r = i / D4, c4 = i - r * D4, tk = c * KC + r;
            const int tc = tk < T ? tk : T - 1;                  // (no branch around the load; rows past T are zeroed when they are written to LDS)
            pre[s] = *reinterpret_cast<const f4 *>(src + (size_t)tc * E + (size_t)h * D + 4 * c4);
        }
    };
    auto put = [&](int c) {
#pragma unroll
        for (int s = 0; s < NST; s++) {
            const int i = tid + 256 * s, r = i / D4, c4 = i - r * D4;
            const bool in = c * KC + r < T;
            float *d = sKV + r * LD + 4 * c4;
            d[0] = in ? pre[s].x : 0.0f; d[1] = in ? pre[s].y : 0.0f; d[2] = in ? pre[s].z : 0.0f; d[3] = in ? pre[s].w : 0.0f;
        }
    };
    const int NC = Tp / KC;
    fetch(k, 0);
    for (int c = 0; c < NC; c++) {
        __syncthreads();
        put(c);
        __syncthreads();
        if (c + 1 < NC) fetch(k, c + 1); else fetch(v, 0);
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.0f;
        const float *qa = sQ + l31 * LD + lh, *kb = sKV + (32 * wave + l31) * LD + lh;
#pragma unroll 8
        for (int s = 0; s < D / 2; s++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[2 * s], kb[2 * s], acc, 0, 0, 0);
        // register r: query (r & 3) + 8 (r >> 2) + 4 lh of the tile, key = 32 wave + (lane & 31) of the chunk
        float *sd = sS + c * KC + 32 * wave + l31;
#pragma unroll
        for (int r = 0; r < 16; r++) sd[((r & 3) + 8 * (r >> 2) + 4 * lh) * LS] = acc[r];
    }
    __syncthreads();
    for (int j = 0; j < QT / 4; j++) {                           // soft-max of row wave + 4 j, as clip_attn_kernel does it
        float *pr = sS + (wave + 4 * j) * LS;
        float mx = -INFINITY;
        for (int tk = lane; tk < T; tk += 64) mx = fmaxf(mx, pr[tk]);
        mx = wave_max(mx);
        float sum = 0.0f;
        for (int tk = lane; tk < T; tk += 64) { const float e = expf(pr[tk] - mx); pr[tk] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int tk = lane; tk < T; tk += 64) pr[tk] *= inv;
        for (int tk = T + lane; tk < Tp; tk += 64) pr[tk] = 0.0f;
    }
    const int dt = wave & 1, kh = wave >> 1;                    // this wave's output half (32 of the 64 elements) and half of every chunk's keys
    f16v o;
#pragma unroll
    for (int r = 0; r < 16; r++) o[r] = 0.0f;
    for (int c = 0; c < NC; c++) {
        __syncthreads();
        put(c);
        __syncthreads();
        if (c + 1 < NC) fetch(v, c + 1);
        const float *pa = sS + l31 * LS + c * KC + kh * 64 + lh, *vb = sKV + (kh * 64 + lh) * LD + dt * 32 + l31;
#pragma unroll 8
        for (int s = 0; s < 32; s++) o = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * s], vb[2 * s * LD], o, 0, 0, 0);
    }
    __syncthreads();                                            // (sKV is free: the second key half's sums go through it)
    float *red = sKV + (size_t)dt * 1024;
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) red[r * 64 + lane] = o[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float val = o[r] + red[r * 64 + lane];
            const int tq = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (tq < T) {
                const size_t at = (size_t)tq * E + (size_t)h * D + dt * 32 + l31;
                out[at] = val;
                if (outh) outh[at] = (_Float16)val;
            }
        }
    }
}
static hipError_t launch_clip_attn_mfma(const float *q, const float *k, const float *v, int T, int H, int n_img, float *out, _Float16 *outh, size_t lds, hipStream_t st) {
    if (lds > 65536) {      // the attribute belongs to the DEVICE's copy of the kernel: set on every such launch (a flag per process was wrong for a second GPU, and raced)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&clip_attn_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(clip_attn_mfma_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)H, (unsigned)n_img), dim3(256), lds, st, q, k, v, T, H, out, outh);
    return hipGetLastError();
}
template <int D>
static hipError_t launch_clip_attn_tiled(const float *q, const float *k, const float *v, int T, int H, int n_img, float *out, _Float16 *outh, size_t lds, hipStream_t st) {
    if (lds > 65536) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&clip_attn_tiled_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(clip_attn_tiled_kernel<D>, dim3((unsigned)((T + 39) / 40), (unsigned)H, (unsigned)n_img), dim3(256), lds, st, q, k, v, T, H, out, outh);
    return hipGetLastError();
}
// out_h (optional): the output rows once more, rounded to f16.  n_img images of T rows each, one after the other in every buffer: attention within an image.
hipError_t launch_clip_attn(const float *q, const float *k, const float *v, int T, int H, int D, float *out, void *out_h, int n_img, hipStream_t st) {
    _Float16 *outh = reinterpret_cast<_Float16 *>(out_h);
    const char *sw = getenv("MI355_CLIP_ATTN_TILED");           // (read at every launch - a few dozen an image - so that a test can compare the kernels)
    const bool tiled = !(sw && atoi(sw) == 0);
    const char *sm_ = getenv("MI355_CLIP_ATTN_MFMA");           // "0": the f32 VALU kernels only
    if (tiled && D == 64 && !(sm_ && atoi(sm_) == 0)) {
        const size_t Tp = (size_t)(T + 127) / 128 * 128, lds = ((size_t)32 * 65 + (size_t)128 * 65 + (size_t)32 * (Tp + 1)) * sizeof(float);
        if (lds <= 160 * 1024) return launch_clip_attn_mfma(q, k, v, T, H, n_img, out, outh, lds, st);
    }
    if (tiled && (D == 32 || D == 64 || D == 128)) {
        const size_t Tp = (size_t)(T + 63) / 64 * 64, lds = ((size_t)40 * D + (size_t)64 * (D + 4) + (size_t)40 * Tp) * sizeof(float);
        if (lds <= 160 * 1024) {
            switch (D) {
                case 32: return launch_clip_attn_tiled<32>(q, k, v, T, H, n_img, out, outh, lds, st);
                case 64: return launch_clip_attn_tiled<64>(q, k, v, T, H, n_img, out, outh, lds, st);
                default: return launch_clip_attn_tiled<128>(q, k, v, T, H, n_img, out, outh, lds, st);
            }
        }
    }
    const dim3 grid((unsigned)((T + 3) / 4), (unsigned)H);
    const size_t lds = (size_t)4 * T * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    for (int im = 0; im < n_img; im++) {
        const size_t o = (size_t)im * T * H * D;
        switch (D) {
            case 32: hipLaunchKernelGGL(clip_attn_kernel<32>, grid, dim3(256), lds, st, q + o, k + o, v + o, T, H, out + o); break;
            case 64: hipLaunchKernelGGL(clip_attn_kernel<64>, grid, dim3(256), lds, st, q + o, k + o, v + o, T, H, out + o); break;
            case 80: hipLaunchKernelGGL(clip_attn_kernel<80>, grid, dim3(256), lds, st, q + o, k + o, v + o, T, H, out + o); break;
            case 128: hipLaunchKernelGGL(clip_attn_kernel<128>, grid, dim3(256), lds, st, q + o, k + o, v + o, T, H, out + o); break;
            default: return hipErrorInvalidValue;
        }
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess || !out_h) return e;
    return launch_f32_to_f16(out, out_h, (size_t)n_img * T * H * D, st);
}

// ggml_gelu / ggml_gelu_quick as the CPU backend evaluates them: through tables indexed by the f16 bits of x, holding f16 results - y = half(f(half(x)));
// GELU leaves the table for |x| >= 10 (0 below, x above)
__global__ void clip_gelu_kernel(float *__restrict__ x, size_t n, int quick, _Float16 *__restrict__ xh) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float y;
    if (quick) {
        const float xh = h2f(f2h(v));
        y = h2f(f2h(xh * (1.0f / (1.0f + expf(-1.702f * xh)))));
    } else if (v <= -10.0f) y = 0.0f;
    else if (v >= 10.0f) y = v;
    else {
        const float xh = h2f(f2h(v));
        y = h2f(f2h(0.5f * xh * (1.0f + tanhf(0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh)))));
    }
    x[i] = y;
    if (xh) xh[i] = (_Float16)y;
}
hipError_t launch_clip_gelu(float *x, size_t n, bool quick, void *xh, hipStream_t st) {
    hipLaunchKernelGGL(clip_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, (int)quick, reinterpret_cast<_Float16 *>(xh));
    return hipGetLastError();
}

}  // namespace mi355
