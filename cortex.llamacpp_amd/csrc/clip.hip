// clip.hip — the small kernels of the LLaVA image encoder (CLIP ViT tower + MLP projector) beside the f16 matrix-core GEMM of mmf.hip and the LayerNorm of
// misc.hip: patch gathering for the convolution-as-GEMM, [class ; patches] + position embeddings, bias (+ scale), the tower's full (unmasked) attention in f32,
// GELU / quick-GELU with ggml's f16-table semantics.
//
// Stands in for the graph clip_image_encode builds (llama.cpp examples/llava/clip.cpp, reached from the reference through
// llava_image_embed_make_with_clip_img, /root/reference/src/llama_server_context.cc:820) — SURVEY.md §8 row f4.  One image is 577 rows of 1024: none of this is
// bandwidth- or matrix-bound work worth tuning; the kernels are written for clarity and run once per image.
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
hipError_t launch_clip_attn(const float *q, const float *k, const float *v, int T, int H, int D, float *out, hipStream_t st) {
    const dim3 grid((unsigned)((T + 3) / 4), (unsigned)H);
    const size_t lds = (size_t)4 * T * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    switch (D) {
        case 32: hipLaunchKernelGGL(clip_attn_kernel<32>, grid, dim3(256), lds, st, q, k, v, T, H, out); break;
        case 64: hipLaunchKernelGGL(clip_attn_kernel<64>, grid, dim3(256), lds, st, q, k, v, T, H, out); break;
        case 80: hipLaunchKernelGGL(clip_attn_kernel<80>, grid, dim3(256), lds, st, q, k, v, T, H, out); break;
        case 128: hipLaunchKernelGGL(clip_attn_kernel<128>, grid, dim3(256), lds, st, q, k, v, T, H, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ggml_gelu / ggml_gelu_quick as the CPU backend evaluates them: through tables indexed by the f16 bits of x, holding f16 results - y = half(f(half(x)));
// GELU leaves the table for |x| >= 10 (0 below, x above)
__global__ void clip_gelu_kernel(float *__restrict__ x, size_t n, int quick) {
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
}
hipError_t launch_clip_gelu(float *x, size_t n, bool quick, hipStream_t st) {
    hipLaunchKernelGGL(clip_gelu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, (int)quick);
    return hipGetLastError();
}

}  // namespace mi355
