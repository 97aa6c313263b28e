// misc.hip — weight row regrouping at upload, embedding lookup (ggml_get_rows with dequantisation),
// f32/f16 mat-vec (MoE router, unquantised tensors), row argmax (device-side greedy front end).
// SURVEY.md §8a rows a16 (get_rows), a18 (router), §8f.1 (device argmax).
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

// ---------------------------------------------------------------- ggml rows -> device rows
__global__ void repack_q6k_kernel(const uint8_t *src, uint8_t *dst, int nb, size_t dst_row) {
    const int row = blockIdx.y, sb = blockIdx.x, i = threadIdx.x;   // 256 threads, 210 used
    if (i >= 210) return;
    const uint8_t v = src[((size_t)row * nb + sb) * 210 + i];
    uint8_t *d = dst + (size_t)row * dst_row;
    if (i < 128) d[(size_t)sb * 128 + i] = v;
    else if (i < 192) d[(size_t)nb * 128 + (size_t)sb * 64 + (i - 128)] = v;
    else if (i < 208) d[(size_t)nb * 192 + (size_t)sb * 16 + (i - 192)] = v;
    else d[(size_t)nb * 208 + (size_t)sb * 2 + (i - 208)] = v;
}
__global__ void repack_q80_kernel(const uint8_t *src, uint8_t *dst, int nblk, int K, size_t dst_row) {
    const int row = blockIdx.y;
    const int b = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    if (b >= nblk) return;
    const uint8_t *s = src + ((size_t)row * nblk + b) * 34;
    uint8_t *d = dst + (size_t)row * dst_row;
    d[(size_t)b * 32 + j] = s[2 + j];
    if (j < 2) d[(size_t)K + (size_t)b * 2 + j] = s[j];
}
// Q2_K block: scales 16 | qs 64 | d 2 | dmin 2 -> planes qs | scales | (d, dmin);  Q3_K block: hmask 32 | qs 64 | scales 12 | d 2 -> hmask | qs | scales | d
__global__ void repack_q2k_kernel(const uint8_t *src, uint8_t *dst, int nb, size_t dst_row) {
    const int row = blockIdx.y, sb = blockIdx.x, i = threadIdx.x;   // 128 threads, 84 used
    if (i >= 84) return;
    const uint8_t v = src[((size_t)row * nb + sb) * 84 + i];
    uint8_t *d = dst + (size_t)row * dst_row;
    if (i < 16) d[(size_t)nb * 64 + (size_t)sb * 16 + i] = v;
    else if (i < 80) d[(size_t)sb * 64 + (i - 16)] = v;
    else d[(size_t)nb * 80 + (size_t)sb * 4 + (i - 80)] = v;
}
__global__ void repack_q3k_kernel(const uint8_t *src, uint8_t *dst, int nb, size_t dst_row) {
    const int row = blockIdx.y, sb = blockIdx.x, i = threadIdx.x;   // 128 threads, 110 used
    if (i >= 110) return;
    const uint8_t v = src[((size_t)row * nb + sb) * 110 + i];
    uint8_t *d = dst + (size_t)row * dst_row;
    if (i < 32) d[(size_t)sb * 32 + i] = v;
    else if (i < 96) d[(size_t)nb * 32 + (size_t)sb * 64 + (i - 32)] = v;
    else if (i < 108) d[(size_t)nb * 96 + (size_t)sb * 12 + (i - 96)] = v;
    else d[(size_t)nb * 108 + (size_t)sb * 2 + (i - 108)] = v;
}
// 32-element blocks with a 16-byte nibble field: Q4_0 / IQ4_NL (d | qs) and Q5_0 (d | qh | qs) -> planes qs | [qh] | d
__global__ void repack_nib32_kernel(const uint8_t *src, uint8_t *dst, int nblk, int K, size_t dst_row, int bsz) {
    const int row = blockIdx.y;
    const int b = blockIdx.x * 8 + (threadIdx.x >> 5), j = threadIdx.x & 31;
    if (b >= nblk || j >= bsz) return;
    const uint8_t v = src[((size_t)row * nblk + b) * bsz + j];
    uint8_t *d = dst + (size_t)row * dst_row;
    const size_t half = (size_t)K >> 1;
    if (bsz == 18) {
        if (j < 2) d[half + (size_t)b * 2 + j] = v; else d[(size_t)b * 16 + (j - 2)] = v;
    } else {
        if (j < 2) d[half + (size_t)nblk * 4 + (size_t)b * 2 + j] = v;
        else if (j < 6) d[half + (size_t)b * 4 + (j - 2)] = v;
        else d[(size_t)b * 16 + (j - 6)] = v;
    }
}
hipError_t launch_repack_rows(int type, const uint8_t *src, uint8_t *dst, int64_t K, int64_t n_rows, hipStream_t st) {
    const size_t drow = dev_row_bytes(type, K);
    if (type == T_Q4_0 || type == T_Q5_0 || type == T_IQ4_NL) {
        const int nblk = (int)(K >> 5);
        for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
            const int nr = (int)((n_rows - r0) < 65535 ? (n_rows - r0) : 65535);
            hipLaunchKernelGGL(repack_nib32_kernel, dim3((nblk + 7) / 8, nr), dim3(256), 0, st,
                               src + (size_t)r0 * ggml_row_bytes(type, K), dst + (size_t)r0 * drow, nblk, (int)K, drow, ggml_block_bytes(type));
        }
        return hipGetLastError();
    }
    if (type == T_Q2_K || type == T_Q3_K) {
        for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
            const int nr = (int)((n_rows - r0) < 65535 ? (n_rows - r0) : 65535);
            if (type == T_Q2_K) hipLaunchKernelGGL(repack_q2k_kernel, dim3((unsigned)(K >> 8), nr), dim3(128), 0, st,
                                                   src + (size_t)r0 * ggml_row_bytes(type, K), dst + (size_t)r0 * drow, (int)(K >> 8), drow);
            else hipLaunchKernelGGL(repack_q3k_kernel, dim3((unsigned)(K >> 8), nr), dim3(128), 0, st,
                                    src + (size_t)r0 * ggml_row_bytes(type, K), dst + (size_t)r0 * drow, (int)(K >> 8), drow);
        }
    } else if (type == T_Q6_K) {
        for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
            const int nr = (int)((n_rows - r0) < 65535 ? (n_rows - r0) : 65535);
            hipLaunchKernelGGL(repack_q6k_kernel, dim3((unsigned)(K >> 8), nr), dim3(256), 0, st,
                               src + (size_t)r0 * ggml_row_bytes(type, K), dst + (size_t)r0 * drow, (int)(K >> 8), drow);
        }
    } else if (type == T_Q8_0) {
        const int nblk = (int)(K >> 5);
        for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
            const int nr = (int)((n_rows - r0) < 65535 ? (n_rows - r0) : 65535);
            hipLaunchKernelGGL(repack_q80_kernel, dim3((nblk + 7) / 8, nr), dim3(256), 0, st,
                               src + (size_t)r0 * ggml_row_bytes(type, K), dst + (size_t)r0 * drow, nblk, (int)K, drow);
        }
    } else {
        if (drow == ggml_row_bytes(type, K)) {
            hipError_t e = hipMemcpyAsync(dst, src, drow * (size_t)n_rows, hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
        } else {
            hipError_t e = hipMemcpy2DAsync(dst, drow, src, ggml_row_bytes(type, K), ggml_row_bytes(type, K), (size_t)n_rows,
                                            hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

// (dequant_elem: one element of a device row - quant_dev.h, shared with the step set-up kernel of attn.hip)
__global__ void get_rows_kernel(int type, const uint8_t *table, int K, size_t row_bytes, const int32_t *ids, float *dst) {
    const int i = blockIdx.y;
    const uint8_t *row = table + (size_t)ids[i] * row_bytes;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K; e += gridDim.x * blockDim.x)
        dst[(size_t)i * K + e] = dequant_elem(type, row, K, e);
}
hipError_t launch_get_rows(int type, const uint8_t *table, int64_t K, const int32_t *ids, int n_ids, float *dst, hipStream_t st) {
    if (n_ids <= 0) return hipSuccess;
    int bx = (int)((K + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(get_rows_kernel, dim3(bx, n_ids), dim3(256), 0, st, type, table, (int)K, dev_row_bytes(type, K), ids, dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------- argmax per row (lowest index wins ties), two stages
constexpr int ARGMAX_PARTS = 64;
__device__ __forceinline__ void argmax_block_reduce(float &best, int &idx, float *bv, int *bi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if (lane == 0) { bv[wave] = best; bi[wave] = idx; }
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w < 4; w++)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
}
// One launch: every workgroup reduces its part, takes a ticket on the row's counter, and the LAST arriver reduces the 64 parts (device-coherent loads
// of what the others wrote, no cache-wide fence) and writes the winner - the second launch this replaced cost ~4.5 us of every decoded token.
__global__ __launch_bounds__(256) void argmax_part_kernel(const float *x, int n, float *pv, int *pi, unsigned *cnt, int32_t *out) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int last_sh;
    const int row = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const float *xr = x + (size_t)row * n;
    const int chunk = (n + ARGMAX_PARTS - 1) / ARGMAX_PARTS;
    const int lo = part * chunk, hi = lo + chunk < n ? lo + chunk : n;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += 256) {
        const float v = xr[i];
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    }
    argmax_block_reduce(best, idx, bv, bi);
    if (tid == 0) {
        __hip_atomic_store(pv + row * ARGMAX_PARTS + part, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pi + row * ARGMAX_PARTS + part, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the two write-through stores have left before the ticket does
        const unsigned t = __hip_atomic_fetch_add(cnt + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = t == ARGMAX_PARTS - 1 ? 1 : 0;
        if (last_sh) __hip_atomic_store(cnt + row, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
    }
    __syncthreads();
    if (!last_sh || tid >= 64) return;
    float b2 = __hip_atomic_load(pv + row * ARGMAX_PARTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int i2 = __hip_atomic_load(pi + row * ARGMAX_PARTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(b2, o, 64);
        const int oi = __shfl_xor(i2, o, 64);
        if (ov > b2 || (ov == b2 && oi < i2)) { b2 = ov; i2 = oi; }
    }
    if (tid == 0) out[row] = i2;
}
// scratch: cap_rows ticket words (zero between launches) | cap_rows * 64 part values | cap_rows * 64 part indices.  The ticket words sit at the HEAD of the
// scratch, at an address that does not depend on `rows`: behind the parts they moved with the row count, and a launch with fewer rows than its predecessor
// found a part value of the earlier launch where its (zero) counter should be (multi-slot serving: 2 generating slots, then 1).
hipError_t launch_argmax_rows(const float *x, int n, int rows, int32_t *out, float *scratch, int cap_rows, hipStream_t st) {
    if (rows < 1 || rows > cap_rows) return hipErrorInvalidValue;
    unsigned *cnt = reinterpret_cast<unsigned *>(scratch);
    float *pv = scratch + cap_rows;
    int *pi = reinterpret_cast<int *>(scratch + cap_rows + (size_t)cap_rows * ARGMAX_PARTS);
    hipLaunchKernelGGL(argmax_part_kernel, dim3(ARGMAX_PARTS, rows), dim3(256), 0, st, x, n, pv, pi, cnt, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------- top-k of a logits row (device-side sampling front end, kernels.h)
// Level 0: a workgroup takes 512 logits, applies the adjustments that fall into its range, sorts the 512 keys in LDS (bitonic, descending) and keeps
// its KK = next power of two >= k best.  Levels 1..: a workgroup merges 1024 / KK such lists the same way, until one list is left; the last level
// writes straight into the caller's (pinned) buffer.  Keys are unique (the token id is part of them), so the order is total and the result exact.
typedef unsigned long long topk_key;
__device__ __forceinline__ topk_key topk_make_key(float v, int tok) {
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;                  // order-preserving image of the float
    return ((topk_key)u << 32) | (topk_key)(0xffffffffu - (unsigned)tok);
}
template <int N, int NT>
__device__ __forceinline__ void topk_sort_desc(topk_key *s, int tid) {
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < N; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const topk_key a = s[i], b = s[ixj];
                    const bool desc = (i & k) == 0;             // this run ends up descending
                    if (desc ? a < b : a > b) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
}
// grid = (lists of 512 logits, rows).  Row r of the launch is logits row rows[r] of `base` with adjustments adj[r]; its lists go to part[r][...]
__global__ __launch_bounds__(256) void topk_part_kernel(const float *base, int n, const int *rows, int kk, const TopkAdj *adjs, topk_key *part, size_t part_stride) {
    __shared__ topk_key s[512];
    __shared__ int a_tok[TOPK_MAX_ADJ], a_cnt[TOPK_MAX_ADJ];
    __shared__ float a_bias[TOPK_MAX_ADJ];
    const int tid = threadIdx.x, first = blockIdx.x * 512, r = blockIdx.y;
    const TopkAdj &adj = adjs[r];
    const int na = adj.n;
    for (int a = tid; a < na; a += 256) { a_tok[a] = adj.tok[a]; a_bias[a] = adj.bias[a]; a_cnt[a] = adj.cnt[a]; }
    const float repeat = adj.repeat, freq = adj.freq, present = adj.present;
    const float *x = base + (size_t)rows[r] * n;
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
        const int i = first + e;
        topk_key key = 0;                                        // below every real key
        if (i < n) {
            float l = x[i];
            for (int a = 0; a < na; a++) {
                if (a_tok[a] != i) continue;
                l = l + a_bias[a];
                if (a_cnt[a] > 0) {
                    if (l <= 0.0f) l *= repeat; else l /= repeat;
                    l -= (float)a_cnt[a] * freq + present;
                }
            }
            key = topk_make_key(l, i);
        }
        s[e] = key;
    }
    __syncthreads();
    topk_sort_desc<512, 256>(s, tid);
    topk_key *o = part + (size_t)r * part_stride + (size_t)blockIdx.x * kk;
    for (int e = tid; e < kk; e += 256) o[e] = s[e];
}
__global__ __launch_bounds__(256) void topk_merge_kernel(const topk_key *in, size_t in_stride, int n_lists, int kk, topk_key *out, size_t out_stride, int out_count) {
    __shared__ topk_key s[1024];
    const int tid = threadIdx.x, per = 1024 / kk, l0 = blockIdx.x * per;
    const topk_key *src = in + (size_t)blockIdx.y * in_stride;
    for (int e = tid; e < 1024; e += 256) {
        const int l = l0 + e / kk;
        s[e] = l < n_lists ? src[(size_t)l * kk + (e % kk)] : 0;
    }
    __syncthreads();
    topk_sort_desc<1024, 256>(s, tid);
    topk_key *o = out + (size_t)blockIdx.y * out_stride + (size_t)blockIdx.x * out_count;
    for (int e = tid; e < out_count; e += 256) o[e] = s[e];
}
static int topk_kk(int k) { int kk = 16; while (kk < k) kk <<= 1; return kk; }
size_t topk_scratch_bytes(int n) { return (size_t)((n + 511) / 512) * TOPK_MAX_K * sizeof(topk_key) * 2; }
hipError_t launch_topk_rows(const float *base, int n, int n_rows, const int *rows_dev, int k, const TopkAdj *adjs_dev, void *scratch, unsigned long long *keys_out,
                            hipStream_t st) {
    if (k < 1 || k > TOPK_MAX_K || n < 1 || n_rows < 1) return hipErrorInvalidValue;
    const int kk = topk_kk(k);
    int lists = (n + 511) / 512;
    const size_t stride = topk_scratch_bytes(n) / sizeof(topk_key) / 2;      // keys per row and buffer
    topk_key *a = reinterpret_cast<topk_key *>(scratch), *b = a + stride * (size_t)n_rows;
    hipLaunchKernelGGL(topk_part_kernel, dim3(lists, n_rows), dim3(256), 0, st, base, n, rows_dev, kk, adjs_dev, a, stride);
    for (;;) {
        const int per = 1024 / kk, blocks = (lists + per - 1) / per;
        const bool last = blocks == 1;
        hipLaunchKernelGGL(topk_merge_kernel, dim3(blocks, n_rows), dim3(256), 0, st, a, stride, lists, kk, last ? keys_out : b, last ? (size_t)TOPK_MAX_K : stride, last ? k : kk);
        if (last) break;
        lists = blocks;
        topk_key *t = a; a = b; b = t;
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------- f32 / f16 weights: one wave per (row, token)
__global__ __launch_bounds__(256) void mmv_float_kernel(int type, const uint8_t *W, int n_rows, int K, const float *x, int T,
                                                        float *y, int ld_out, const float *resid) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= n_rows * T) return;
    const int r = gw % n_rows, t = gw / n_rows;
    const float *xr = x + (size_t)t * K;
    // sixteen loads in flight per lane, then the products are added in the same order as a plain k-loop would add them
    // (a dependent load per iteration made the 8 x 4096 router of a mixture-of-experts layer take 18 us)
    double s = 0.0;
    constexpr int UN = 16;
    if (type == T_F32) {
        const float *w = reinterpret_cast<const float *>(W) + (size_t)r * K;
        for (int k0 = lane; k0 < K; k0 += 64 * UN) {
            float wv[UN], xv[UN];
#pragma unroll
            for (int j = 0; j < UN; j++) { const int k = k0 + 64 * j, kc = k < K ? k : lane; wv[j] = w[kc]; xv[j] = xr[kc]; }
#pragma unroll
            for (int j = 0; j < UN; j++) if (k0 + 64 * j < K) s += (double)(wv[j] * xv[j]);
        }
    } else {
        const uint16_t *w = reinterpret_cast<const uint16_t *>(W) + (size_t)r * K;
        for (int k0 = lane; k0 < K; k0 += 64 * UN) {
            uint16_t wv[UN]; float xv[UN];
#pragma unroll
            for (int j = 0; j < UN; j++) { const int k = k0 + 64 * j, kc = k < K ? k : lane; wv[j] = w[kc]; xv[j] = xr[kc]; }
#pragma unroll
            for (int j = 0; j < UN; j++) if (k0 + 64 * j < K) s += (double)(h2f(wv[j]) * h2f(f2h(xv[j])));   // activations cast to f16 like the CPU path
        }
    }
    s = wave_sum(s);
    if (lane == 0) {
        const size_t o = (size_t)t * ld_out + r;
        y[o] = resid ? resid[o] + (float)s : (float)s;
    }
}
hipError_t launch_mmv_float(int type, const uint8_t *W, int n_rows, int K, const float *x, int T, float *y, int ld_out,
                            const float *resid, hipStream_t st) {
    const int waves = n_rows * T;
    hipLaunchKernelGGL(mmv_float_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, type, W, n_rows, K, x, T, y, ld_out, resid);
    return hipGetLastError();
}

// ---------------------------------------------------------------- MoE routing (build_moe_ffn: softmax -> top-k -> renormalise)
// forced (test hook, mi355_debug_force_moe_ids): the experts of token t are forced[t * k + j] instead of the k most probable; the weights are still this
// side's probabilities of those experts, renormalised
__global__ void moe_route_kernel(const float *logits, int T, int n_expert, int k, int32_t *ids, float *w, const int32_t *forced) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float *x = logits + (size_t)t * n_expert;
    float p[64];
    float mx = -INFINITY;
    for (int e = 0; e < n_expert; e++) mx = fmaxf(mx, x[e]);
    double sum = 0.0;
    for (int e = 0; e < n_expert; e++) { p[e] = expf(x[e] - mx); sum += (double)p[e]; }
    const float inv = (float)(1.0 / sum);
    for (int e = 0; e < n_expert; e++) p[e] *= inv;
    unsigned long long used = 0;
    float wsum = 0.0f;
    for (int j = 0; j < k; j++) {
        int best = -1;
        for (int e = 0; e < n_expert; e++)
            if (!((used >> e) & 1ull) && (best < 0 || p[e] > p[best])) best = e;
        if (forced) { best = forced[(size_t)t * k + j]; best = best < 0 ? 0 : best >= n_expert ? n_expert - 1 : best; }
        used |= 1ull << best;
        ids[(size_t)t * k + j] = best;
        w[(size_t)t * k + j] = p[best];
        wsum += p[best];
    }
    for (int j = 0; j < k; j++) w[(size_t)t * k + j] /= wsum;
}
hipError_t launch_moe_route(const float *logits, int T, int n_expert, int k, int32_t *ids, float *w, hipStream_t st, const int32_t *forced) {
    if (n_expert > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_route_kernel, dim3((T + 63) / 64), dim3(64), 0, st, logits, T, n_expert, k, ids, w, forced);
    return hipGetLastError();
}
// Router of a mixture-of-experts layer in ONE launch: logits = gate_inp . x (the arithmetic of mmv_float_kernel: lane l adds
// the products k = l, l + 64, ... in a double, waves reduced the same way), then the selection of moe_route_kernel, per
// token.  Workgroup = token; wave w computes experts w, w + 4, ...; thread 0 selects.  Same bits as the two launches.
__global__ __launch_bounds__(256) void moe_router_kernel(int type, const uint8_t *W, int n_expert, int K, const float *x, int k, float *logits_out,
                                                         int32_t *ids, float *w, const int32_t *forced) {
    __shared__ float lg[64];
    const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *xr = x + (size_t)t * K;
    constexpr int UN = 16;
    for (int e = wave; e < n_expert; e += 4) {
        double s = 0.0;
        if (type == T_F32) {
            const float *wr = reinterpret_cast<const float *>(W) + (size_t)e * K;
            for (int k0 = lane; k0 < K; k0 += 64 * UN) {
                float wv[UN], xv[UN];
#pragma unroll
                for (int j = 0; j < UN; j++) { const int kk = k0 + 64 * j, kc = kk < K ? kk : lane; wv[j] = wr[kc]; xv[j] = xr[kc]; }
#pragma unroll
                for (int j = 0; j < UN; j++) if (k0 + 64 * j < K) s += (double)(wv[j] * xv[j]);
            }
        } else {
            const uint16_t *wr = reinterpret_cast<const uint16_t *>(W) + (size_t)e * K;
            for (int k0 = lane; k0 < K; k0 += 64 * UN) {
                uint16_t wv[UN]; float xv[UN];
#pragma unroll
                for (int j = 0; j < UN; j++) { const int kk = k0 + 64 * j, kc = kk < K ? kk : lane; wv[j] = wr[kc]; xv[j] = xr[kc]; }
#pragma unroll
                for (int j = 0; j < UN; j++) if (k0 + 64 * j < K) s += (double)(h2f(wv[j]) * h2f(f2h(xv[j])));
            }
        }
        s = wave_sum(s);
        if (lane == 0) { lg[e] = (float)s; if (logits_out) logits_out[(size_t)t * n_expert + e] = (float)s; }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    float p[64];
    float mx = -INFINITY;
    for (int e = 0; e < n_expert; e++) mx = fmaxf(mx, lg[e]);
    double sum = 0.0;
    for (int e = 0; e < n_expert; e++) { p[e] = expf(lg[e] - mx); sum += (double)p[e]; }
    const float inv = (float)(1.0 / sum);
    for (int e = 0; e < n_expert; e++) p[e] *= inv;
    unsigned long long used = 0;
    float wsum = 0.0f;
    for (int j = 0; j < k; j++) {
        int best = -1;
        for (int e = 0; e < n_expert; e++)
            if (!((used >> e) & 1ull) && (best < 0 || p[e] > p[best])) best = e;
        if (forced) { best = forced[(size_t)t * k + j]; best = best < 0 ? 0 : best >= n_expert ? n_expert - 1 : best; }
        used |= 1ull << best;
        ids[(size_t)t * k + j] = best;
        w[(size_t)t * k + j] = p[best];
        wsum += p[best];
    }
    for (int j = 0; j < k; j++) w[(size_t)t * k + j] /= wsum;
}
hipError_t launch_moe_router(int type, const uint8_t *W, int n_expert, int K, const float *x, int T, int k, float *logits_out, int32_t *ids, float *w,
                             hipStream_t st, const int32_t *forced) {
    if (n_expert > 64 || (type != T_F32 && type != T_F16) || T <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_router_kernel, dim3(T), dim3(256), 0, st, type, W, n_expert, K, x, k, logits_out, ids, w, forced);
    return hipGetLastError();
}
__global__ void moe_combine_kernel(float *x, const float *eo, const float *w, int T, int E, int k, size_t eo_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * E) return;
    const int t = (int)(i / E);
    float o = eo[i] * w[(size_t)t * k];
    for (int j = 1; j < k; j++) o = o + eo[(size_t)j * eo_stride + i] * w[(size_t)t * k + j];
    x[i] = x[i] + o;
}
hipError_t launch_moe_combine(float *x, const float *eo, const float *w, int T, int E, int k, size_t eo_stride, hipStream_t st) {
    const int64_t n = (int64_t)T * E;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, eo, w, T, E, k, eo_stride);
    return hipGetLastError();
}
// ---------------------------------------------------------------- grouping by expert (ggml_mul_mat_id with a batch of tokens)
// One workgroup of 4 waves.  The selections are staged in LDS; wave w owns experts w, w + 4, ...: it walks the selections
// 64 at a time in (token, rank) order, a ballot marks the ones that chose its expert and a prefix count of the ballot
// gives each its row: stable, deterministic, no atomics.  (Runs once per layer of a prompt batch: a few microseconds.)
__global__ __launch_bounds__(256) void moe_group_kernel(const int32_t *ids, int n_sel, int k, int n_expert, int32_t *meta, int32_t *slot_of, int32_t *tok_of) {
    extern __shared__ int32_t sid[];               // [n_sel]
    __shared__ int cnt[64], off[65];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_sel; i += 256) sid[i] = ids[i];
    __syncthreads();
    for (int e = wave; e < n_expert; e += 4) {
        int c = 0;
        for (int b = 0; b < n_sel; b += 64) {
            const bool hit = b + lane < n_sel && sid[b + lane] == e;
            c += __popcll(__ballot(hit));
        }
        if (lane == 0) cnt[e] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int j = 0; j < n_expert; j++) { off[j] = run; run += cnt[j]; }
        off[n_expert] = run;
    }
    __syncthreads();
    if (tid < n_expert) { meta[tid] = cnt[tid]; meta[n_expert + tid] = off[tid]; }
    if (tid == 0) meta[2 * n_expert] = off[n_expert];
    for (int e = wave; e < n_expert; e += 4) {
        int r = off[e];
        for (int b = 0; b < n_sel; b += 64) {
            const bool hit = b + lane < n_sel && sid[b + lane] == e;
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int row = r + __popcll(m & ((1ull << lane) - 1ull));
                slot_of[b + lane] = row;
                tok_of[row] = (b + lane) / k;
            }
            r += __popcll(m);
        }
    }
}
hipError_t launch_moe_group(const int32_t *ids, int T, int k, int n_expert, int32_t *meta, int32_t *slot_of, int32_t *tok_of, hipStream_t st) {
    if (n_expert > 64 || T * k <= 0 || (size_t)T * k * 4 > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_group_kernel, dim3(1), dim3(256), (size_t)T * k * 4, st, ids, T * k, k, n_expert, meta, slot_of, tok_of);
    return hipGetLastError();
}
// one workgroup per grouped row: 16-byte copies of the code plane(s), then the small per-block planes
__global__ __launch_bounds__(256) void moe_gather_act_kernel(ActQuant src, const int32_t *tok_of, int K, ActQuant dst) {
    const int r = blockIdx.x, t = tok_of[r], tid = threadIdx.x;
    if (src.qs && dst.qs) {
        const uint4 *s = reinterpret_cast<const uint4 *>(src.qs + (size_t)t * K);
        uint4 *d = reinterpret_cast<uint4 *>(dst.qs + (size_t)r * K);
        for (int i = tid; i < (K >> 4); i += 256) d[i] = s[i];
        for (int i = tid; i < (K >> 8); i += 256) dst.d[(size_t)r * (K >> 8) + i] = src.d[(size_t)t * (K >> 8) + i];
        for (int i = tid; i < (K >> 4); i += 256) dst.bsums[(size_t)r * (K >> 4) + i] = src.bsums[(size_t)t * (K >> 4) + i];
    }
    if (src.qs0 && dst.qs0) {
        const uint4 *s = reinterpret_cast<const uint4 *>(src.qs0 + (size_t)t * K);
        uint4 *d = reinterpret_cast<uint4 *>(dst.qs0 + (size_t)r * K);
        for (int i = tid; i < (K >> 4); i += 256) d[i] = s[i];
        for (int i = tid; i < (K >> 5); i += 256) dst.d0[(size_t)r * (K >> 5) + i] = src.d0[(size_t)t * (K >> 5) + i];
    }
}
hipError_t launch_moe_gather_act(const ActQuant &src, const int32_t *tok_of, int n_rows, int K, const ActQuant &dst, hipStream_t st) {
    if (n_rows <= 0) return hipSuccess;
    if (K % 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_gather_act_kernel, dim3(n_rows), dim3(256), 0, st, src, tok_of, K, dst);
    return hipGetLastError();
}
__global__ void moe_scatter_combine_kernel(float *x, const float *y, const float *w, const int32_t *slot_of, int T, int E, int k) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)T * E) return;
    const int t = (int)(i / E), d = (int)(i - (int64_t)t * E);
    float o = y[(size_t)slot_of[(size_t)t * k] * E + d] * w[(size_t)t * k];
    for (int j = 1; j < k; j++) o = o + y[(size_t)slot_of[(size_t)t * k + j] * E + d] * w[(size_t)t * k + j];
    x[i] = x[i] + o;
}
hipError_t launch_moe_scatter_combine(float *x, const float *y, const float *w, const int32_t *slot_of, int T, int E, int k, hipStream_t st) {
    const int64_t n = (int64_t)T * E;
    hipLaunchKernelGGL(moe_scatter_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, y, w, slot_of, T, E, k);
    return hipGetLastError();
}

__global__ void gather_rows_kernel(const float *src, const int32_t *rows, int n, float *dst) {
    const int r = blockIdx.y;
    const float *s = src + (size_t)rows[r] * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[(size_t)r * n + i] = s[i];
}
hipError_t launch_gather_rows_f32(const float *src, const int32_t *rows, int n_rows, int n, float *dst, hipStream_t st) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(8, n_rows), dim3(256), 0, st, src, rows, n, dst);
    return hipGetLastError();
}

}  // namespace mi355

// ---------------------------------------------------------------- HBM read probe (bench.py: measured peak beside the spec peak)
__global__ __launch_bounds__(256) void hbm_read_kernel(const uint4 *src, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) *sink = acc;   // never true in practice; keeps the loads alive
}
extern "C" double hbm_read_probe(size_t bytes, int iters) {
    void *buf = nullptr, *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) return -1.0;
    (void)hipMemset(buf, 0x5a, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = mi355::num_cu() * 8;
    hipLaunchKernelGGL(hbm_read_kernel, dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)buf, bytes / 16, (unsigned *)sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; i++)
        hipLaunchKernelGGL(hbm_read_kernel, dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)buf, bytes / 16, (unsigned *)sink);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(sink);
    return (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
}
