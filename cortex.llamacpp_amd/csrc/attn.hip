// attn.hip — RoPE + KV-cache store, flash attention over an f16 / q8_0 / q4_0 cache, K re-rotation.
//
// Stand in for ggml_rope_ext (mode NORM / NEOX), ggml_cpy f32->{f16,q8_0,q4_0} into the cache views,
// ggml_flash_attn_ext (ggml-cpu flash_attn_ext_f16: Q converted to the K cache's vec_dot type, online
// softmax, F32 accumulation) and the K-shift graph — SURVEY.md §8a rows a11, a13, a15, a4; reached from the
// reference through llama_decode (src/llama_server_context.cc:1635) and llama_kv_cache_seq_add (:1290).
//
// Cache layout in HBM is head-major per layer: plane[g][cell][...] so that one kv-head's cells are a
// contiguous stream; quantised caches keep int8/int4 codes and f16 block scales in separate planes
// (same values as ggml block_q8_0 / block_q4_0, regrouped for 16-byte aligned coalesced loads).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cstring>
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

// ------------------------------------------------------------------------------------------
// cos/sin table of one position: theta_0 = pos, theta_{i+1} = theta_i * theta_scale, iterated in f32
// exactly like ggml_rope_cache_init, so the angles match the CPU path bit for bit (cosf/sinf within libm ulp).
__device__ __forceinline__ void rope_angle(int i_pair, int32_t pos, float theta_scale, const RopeArgs &ra, float &c, float &s) {
    float theta = (float)pos;
    for (int k = 0; k < i_pair; k++) theta *= theta_scale;
    const float f = ra.freq_factors ? ra.freq_factors[i_pair] : 1.0f;
    const float extrap = theta / f;
    float th = ra.freq_scale * extrap, m = ra.attn_factor;
    if (ra.ext_factor != 0.0f) {     // YaRN (rope_yarn): ramp 1 below corr_lo (the original angle), 0 above corr_hi (the interpolated one)
        const float y = ((float)i_pair - ra.corr_lo) / fmaxf(0.001f, ra.corr_hi - ra.corr_lo);
        const float mix = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * ra.ext_factor;
        th = th * (1.0f - mix) + extrap * mix;
        m *= 1.0f + 0.1f * logf(1.0f / ra.freq_scale);
    }
    c = cosf(th) * m;
    s = sinf(th) * m;
}

__device__ __forceinline__ void rope_heads_lds(float *buf, int n_head, int D, int n_rot, int neox, const float *cs, int tid, int nthr) {
    const int half = n_rot >> 1;
    for (int idx = tid; idx < n_head * half; idx += nthr) {
        const int h = idx / half, i = idx - h * half;
        float *p = buf + (size_t)h * D;
        const int a = neox ? i : 2 * i, b = neox ? i + half : 2 * i + 1;
        const float x0 = p[a], x1 = p[b], c = cs[2 * i], s = cs[2 * i + 1];
        p[a] = x0 * c - x1 * s;
        p[b] = x0 * s + x1 * c;
    }
}

// store one row of n = G*D f32 values (LDS or global) into cache planes at `cell`; 32 consecutive threads per block
__device__ __forceinline__ void store_row(const float *src, int G, int D, int type, uint8_t *plane, uint16_t *dplane,
                                          int n_ctx, int cell, int tid, int nthr) {
    const int n = G * D;
    for (int e0 = 0; e0 < n; e0 += nthr) {
        const int e = e0 + tid;
        const bool ok = e < n;
        const float v = ok ? src[e] : 0.0f;
        const int g = ok ? e / D : 0, dd = ok ? e - g * D : 0;
        const size_t rowi = (size_t)g * n_ctx + cell;
        if (type == T_F16) {
            if (ok) reinterpret_cast<uint16_t *>(plane)[rowi * D + dd] = f2h(v);
        } else if (type == T_Q8_0) {
            float am = fabsf(v);
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
            const float d = am / 127.0f;
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            if (ok) {
                reinterpret_cast<int8_t *>(plane)[rowi * D + dd] = (int8_t)roundf(v * id);
                if ((dd & 31) == 0) dplane[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
            }
        } else {  // T_Q4_0: first element with the largest magnitude sets d = max / -8
            unsigned long long key = ((unsigned long long)__float_as_uint(fabsf(v)) << 32) | (unsigned long long)(0xffffffffu - (unsigned)(dd & 31));
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
                const unsigned long long w = __shfl_xor(key, o, 64);
                key = w > key ? w : key;
            }
            const int imax = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            const float vmax = __shfl(v, (threadIdx.x & 32) + imax, 64);
            const float d = vmax / -8.0f;
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            int qv = (int)(int8_t)(v * id + 8.5f);
            qv = qv > 15 ? 15 : qv;
            const int other = __shfl_xor(qv, 16, 64);   // element j pairs with j+16 in one byte
            if (ok) {
                const int j = dd & 31;
                if (j < 16) plane[rowi * (D >> 1) + (dd >> 5) * 16 + j] = (uint8_t)(qv | (other << 4));
                if (j == 0) dplane[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
            }
        }
    }
}

// cos/sin table of every token of the micro-batch, computed once and reused by all layers
__global__ void rope_table_kernel(const int32_t *tok_pos, int T, RopeArgs ra, float theta_scale, float *cs_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = ra.n_rot >> 1;
    if (idx >= T * half) return;
    const int t = idx / half, i = idx - t * half;
    float c, s;
    rope_angle(i, tok_pos[t], theta_scale, ra, c, s);
    cs_out[(size_t)t * ra.n_rot + 2 * i] = c;
    cs_out[(size_t)t * ra.n_rot + 2 * i + 1] = s;
}
hipError_t launch_rope_table(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, hipStream_t st) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const int n = T * (ra.n_rot >> 1);
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 63) / 64), dim3(64), 0, st, tok_pos, T, ra, theta_scale, cs_out);
    return hipGetLastError();
}

// the two tiny launches that open every micro-batch as one: the cos/sin table (above) and the cell metadata of the
// batch's tokens (launch_kv_meta_set below); ~5 us of a 1.7 ms single-token step each
__global__ void step_setup_kernel(const int32_t *tok_pos, int T, RopeArgs ra, float theta_scale, float *cs_out, int32_t *cell_pos, uint64_t *cell_seq,
                                  const int32_t *tok_cell, const uint64_t *tok_seqmask, unsigned *zero_word, unsigned *epoch_word) {
    if (blockIdx.x == 0) {
        if (zero_word && threadIdx.x < 9) zero_word[threadIdx.x == 0 ? 0 : 32 * threadIdx.x] = 0u;   // barrier counters of the whole-step kernel (decode_mega.hip)
        if (epoch_word && threadIdx.x == 0) epoch_word[0] = epoch_word[0] + 1u;                       // step serial (decode_engine.hip)
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            cell_pos[tok_cell[t]] = tok_pos[t];
            cell_seq[tok_cell[t]] = tok_seqmask[t];
        }
    }
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = ra.n_rot >> 1;
    if (idx >= T * half) return;
    const int t = idx / half, i = idx - t * half;
    float c, s;
    rope_angle(i, tok_pos[t], theta_scale, ra, c, s);
    cs_out[(size_t)t * ra.n_rot + 2 * i] = c;
    cs_out[(size_t)t * ra.n_rot + 2 * i + 1] = s;
}
// ... and the embedding rows of the batch's tokens (ggml_get_rows of token_embd, dequantised: get_rows_kernel of misc.hip) in the same launch: grid
// (row parts, tokens) of 256 threads; the first T * n_rot / 2 threads of the grid also fill the cos / sin table, block (0, 0) the cell metadata
__global__ __launch_bounds__(256) void step_setup_embed_kernel(const int32_t *tok_pos, int T, RopeArgs ra, float theta_scale, float *cs_out, int32_t *cell_pos,
                                                               uint64_t *cell_seq, const int32_t *tok_cell, const uint64_t *tok_seqmask, unsigned *zero_word,
                                                               unsigned *epoch_word, int type, const uint8_t *table, int K, size_t row_bytes,
                                                               const int32_t *ids, float *dst) {
    const int lin = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (zero_word && threadIdx.x < 9) zero_word[threadIdx.x == 0 ? 0 : 32 * threadIdx.x] = 0u;
        if (epoch_word && threadIdx.x == 0) epoch_word[0] = epoch_word[0] + 1u;
        for (int t = threadIdx.x; t < T; t += blockDim.x) {
            cell_pos[tok_cell[t]] = tok_pos[t];
            cell_seq[tok_cell[t]] = tok_seqmask[t];
        }
    }
    const int half = ra.n_rot >> 1;
    if (lin < T * half) {
        const int t = lin / half, i = lin - t * half;
        float c, s;
        rope_angle(i, tok_pos[t], theta_scale, ra, c, s);
        cs_out[(size_t)t * ra.n_rot + 2 * i] = c;
        cs_out[(size_t)t * ra.n_rot + 2 * i + 1] = s;
    }
    const int i = blockIdx.y;
    const uint8_t *row = table + (size_t)ids[i] * row_bytes;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < K; e += gridDim.x * blockDim.x)
        dst[(size_t)i * K + e] = dequant_elem(type, row, K, e);
}
hipError_t launch_step_setup_embed(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell,
                                   const uint64_t *tok_seqmask, unsigned *zero_word, unsigned *epoch_word, int type, const uint8_t *table, int64_t K,
                                   const int32_t *ids, float *dst, hipStream_t st) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    int bx = (int)((K + 255) / 256);
    if (bx > 64) bx = 64;
    while ((long)bx * T * 256 < (long)T * (ra.n_rot >> 1)) bx++;          // (enough threads for the table: n_rot / 2 <= 256 * bx per token)
    hipLaunchKernelGGL(step_setup_embed_kernel, dim3(bx, T), dim3(256), 0, st, tok_pos, T, ra, theta_scale, cs_out, cell_pos, cell_seq, tok_cell, tok_seqmask,
                       zero_word, epoch_word, type, table, (int)K, dev_row_bytes(type, K), ids, dst);
    return hipGetLastError();
}
hipError_t launch_step_setup(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell,
                             const uint64_t *tok_seqmask, unsigned *zero_word, hipStream_t st, unsigned *epoch_word) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const int n = T * (ra.n_rot >> 1);
    hipLaunchKernelGGL(step_setup_kernel, dim3(n > 0 ? (n + 63) / 64 : 1), dim3(64), 0, st, tok_pos, T, ra, theta_scale, cs_out, cell_pos, cell_seq, tok_cell,
                       tok_seqmask, zero_word, epoch_word);
    return hipGetLastError();
}

// One workgroup per token: rope(q) in place; rope(k) -> cache; v -> cache.  cs_table (nullable): precomputed [T][n_rot].
__global__ __launch_bounds__(256) void rope_kv_store_kernel(float *q, const float *k, const float *v, int n_head, int G, int D,
                                                            const int32_t *tok_pos, const int32_t *tok_cell, RopeArgs ra,
                                                            float theta_scale, KVLayerView kv, int type_k, int type_v, int n_ctx,
                                                            const float *cs_table) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *cs = sm;                 // [n_rot]
    float *kbuf = sm + ra.n_rot;    // [G*D]
    const int t = blockIdx.x, tid = threadIdx.x;
    if (cs_table) {
        for (int i = tid; i < ra.n_rot; i += 256) cs[i] = cs_table[(size_t)t * ra.n_rot + i];
    } else {
        const int32_t pos = tok_pos[t];
        for (int i = tid; i < (ra.n_rot >> 1); i += 256) rope_angle(i, pos, theta_scale, ra, cs[2 * i], cs[2 * i + 1]);
    }
    if (k) for (int e = tid; e < G * D; e += 256) kbuf[e] = k[(size_t)t * G * D + e];
    __syncthreads();
    float *qr = q + (size_t)t * n_head * D;
    rope_heads_lds(qr, n_head, D, ra.n_rot, ra.neox, cs, tid, 256);   // q rotated in global memory
    if (k) {
        rope_heads_lds(kbuf, G, D, ra.n_rot, ra.neox, cs, tid, 256);
        __syncthreads();
        const int cell = tok_cell[t];
        store_row(kbuf, G, D, type_k, kv.k, kv.kd, n_ctx, cell, tid, 256);
        store_row(v + (size_t)t * G * D, G, D, type_v, kv.v, kv.vd, n_ctx, cell, tid, 256);
    }
}

hipError_t launch_rope_kv_store(float *q, const float *k, const float *v, int T, int n_head, int G, int D,
                                const int32_t *tok_pos, const int32_t *tok_cell, RopeArgs ra, KVLayerView kv,
                                int type_k, int type_v, int n_ctx, const float *cs_table, hipStream_t st) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const size_t lds = sizeof(float) * ((size_t)ra.n_rot + (size_t)G * D);
    hipLaunchKernelGGL(rope_kv_store_kernel, dim3(T), dim3(256), lds, st, q, k, v, n_head, G, D, tok_pos, tok_cell, ra,
                       theta_scale, kv, type_k, type_v, n_ctx, cs_table);
    return hipGetLastError();
}

hipError_t launch_rope_inplace(float *x, int T, int n_head, int D, const int32_t *tok_pos, RopeArgs ra, hipStream_t st) {
    KVLayerView kv{};
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const size_t lds = sizeof(float) * ((size_t)ra.n_rot);
    hipLaunchKernelGGL(rope_kv_store_kernel, dim3(T), dim3(256), lds, st, x, (const float *)nullptr, (const float *)nullptr,
                       n_head, 0, D, tok_pos, (const int32_t *)nullptr, ra, theta_scale, kv, T_F16, T_F16, 0, (const float *)nullptr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K-shift: re-rotate every cached K row whose position moved by delta (dequantise, rope(delta), requantise)
__global__ __launch_bounds__(256) void k_shift_kernel(KVLayerView kv, int type_k, int G, int D, int n_ctx, const int32_t *delta,
                                                      RopeArgs ra, float theta_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int cell = blockIdx.x, tid = threadIdx.x;
    const int32_t dl = delta[cell];
    if (dl == 0) return;
    float *cs = sm, *kbuf = sm + ra.n_rot;
    for (int i = tid; i < (ra.n_rot >> 1); i += 256) rope_angle(i, dl, theta_scale, ra, cs[2 * i], cs[2 * i + 1]);
    for (int e = tid; e < G * D; e += 256) {
        const int g = e / D, dd = e - g * D;
        const size_t rowi = (size_t)g * n_ctx + cell;
        float v;
        if (type_k == T_F16) v = h2f(reinterpret_cast<const uint16_t *>(kv.k)[rowi * D + dd]);
        else if (type_k == T_Q8_0) v = (float)reinterpret_cast<const int8_t *>(kv.k)[rowi * D + dd] * h2f(kv.kd[rowi * (D >> 5) + (dd >> 5)]);
        else {
            const int j = dd & 31;
            const uint8_t b = kv.k[rowi * (D >> 1) + (dd >> 5) * 16 + (j & 15)];
            v = (float)((j < 16 ? (b & 0x0f) : (b >> 4)) - 8) * h2f(kv.kd[rowi * (D >> 5) + (dd >> 5)]);
        }
        kbuf[e] = v;
    }
    __syncthreads();
    rope_heads_lds(kbuf, G, D, ra.n_rot, ra.neox, cs, tid, 256);
    __syncthreads();
    store_row(kbuf, G, D, type_k, kv.k, kv.kd, n_ctx, cell, tid, 256);
}
hipError_t launch_k_shift(KVLayerView kv, int type_k, int G, int D, int n_ctx, const int32_t *delta, RopeArgs ra, hipStream_t st) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const size_t lds = sizeof(float) * ((size_t)ra.n_rot + (size_t)G * D);
    hipLaunchKernelGGL(k_shift_kernel, dim3(n_ctx), dim3(256), lds, st, kv, type_k, G, D, n_ctx, delta, ra, theta_scale);
    return hipGetLastError();
}

__global__ void kv_meta_set_kernel(int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell, const int32_t *tok_pos,
                                   const uint64_t *tok_seqmask, int T, unsigned *zero_word) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (zero_word && t < 9) zero_word[t == 0 ? 0 : 32 * t] = 0u;   // barrier counters of the whole-step kernel that follows (decode_mega.hip)
    if (t < T) {
        cell_pos[tok_cell[t]] = tok_pos[t];
        cell_seq[tok_cell[t]] = tok_seqmask[t];
    }
}
hipError_t launch_kv_meta_set(int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell, const int32_t *tok_pos,
                              const uint64_t *tok_seqmask, int T, hipStream_t st, unsigned *zero_word) {
    hipLaunchKernelGGL(kv_meta_set_kernel, dim3((T + 255) / 256), dim3(256), 0, st, cell_pos, cell_seq, tok_cell, tok_pos, tok_seqmask, T, zero_word);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Flash attention, split over the KV axis.  Workgroup = (kv head g, token t, split s), 4 waves.
//   pass 1: scores of this split's cells for the R = H/G query heads that share g  -> LDS
//           (8 lanes per cell for D = 128: every lane loads 16 codes = 16 contiguous bytes of the row)
//   pass 2: m = max, p = exp(s - m), l = sum p   (whole split at once, so no in-loop rescale)
//   pass 3: acc[r][d] += p[r][c] * V[c][d]; thread = (4 consecutive d, one of 8 cell groups)
// then a combine kernel merges the splits with their (m, l).
constexpr int ATT_MAX_CHUNK = 1024;   // cells per split (LDS: R * chunk floats)

template <int D, int R>
__global__ __launch_bounds__(256) void flash_attn_split_kernel(const AttnArgs a) {
    constexpr int LPC = D / 16;          // lanes per cell in the score pass
    constexpr int CPW = 64 / LPC;        // cells per wave iteration
    constexpr int NB = D / 32;           // 32-element blocks per head row
    constexpr int DQ = D / 4;            // threads along d in the PV pass
    constexpr int NCG = 256 / DQ;        // cell groups in the PV pass
    extern __shared__ __attribute__((aligned(16))) uint8_t smraw[];
    float *S = reinterpret_cast<float *>(smraw);                         // [R][chunk]
    const int g = blockIdx.x, t = blockIdx.y, sp = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_kv = *a.n_kv_dev;
    int chunk = (n_kv + a.splits - 1) / a.splits;
    chunk = (chunk + 31) & ~31;
    if (chunk > ATT_MAX_CHUNK) chunk = ATT_MAX_CHUNK;   // host guarantees splits * ATT_MAX_CHUNK >= n_kv
    const int c_lo = sp * chunk;
    int c_hi = c_lo + chunk;
    if (c_hi > n_kv) c_hi = n_kv;
    const int ncell = c_hi > c_lo ? c_hi - c_lo : 0;
    float *qf = S + (size_t)R * chunk;                                   // [R][D]  q as the dot sees it (f16-rounded) / unused
    int8_t *qc = reinterpret_cast<int8_t *>(qf + R * D);                 // [R][D]  q8_0 codes
    float *qd = reinterpret_cast<float *>(qc + R * D);                   // [R][NB] q8_0 scales (f16-rounded)
    int *qs16 = reinterpret_cast<int *>(qd + R * NB);                    // [R][D/16] sums of 16 codes (q4_0 K)
    float *red = reinterpret_cast<float *>(qs16 + R * (D / 16));         // [R][2] m, l  then reduction scratch
    float *accs = red + 2 * R;                                           // [NCG][R][D]

    const int32_t tpos = a.tok_pos[t];
    const int tseq = a.tok_seq[t];
    const int H = a.H, n_ctx = a.n_ctx;
    const float *qrow = a.q + ((size_t)t * H + (size_t)g * R) * D;

    // ---- Q in the K cache's vec_dot type
    if (a.type_k == T_F16) {
        for (int e = tid; e < R * D; e += 256) qf[e] = h2f(f2h(qrow[e]));
    } else {
        for (int e0 = 0; e0 < R * D; e0 += 256) {       // R*D is a multiple of 32; 32 consecutive threads per block
            const int e = e0 + tid;
            const bool ok = e < R * D;
            const float v = ok ? qrow[e] : 0.0f;
            float am = fabsf(v);
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
            const float d = am / 127.0f;
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            const int qv = (int)roundf(v * id);
            int s16 = qv;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s16 += __shfl_xor(s16, o, 64);
            if (ok) {
                qc[e] = (int8_t)qv;
                if ((e & 31) == 0) qd[e >> 5] = h2f(f2h(d));
                if ((e & 15) == 0) qs16[e >> 4] = s16;
            }
        }
    }
    __syncthreads();

    // ---- pass 1: scores
    const int sub = lane % LPC;            // which 16-element slice of the row
    const int cw = lane / LPC;             // cell within the wave iteration
    for (int c0 = wave * CPW; c0 < ncell; c0 += 4 * CPW) {
        const int cl = c0 + cw;
        const int c = c_lo + cl;
        bool vis = cl < ncell;
        if (vis) {
            const int32_t cp = a.cell_pos[c];
            vis = cp >= 0 && cp <= tpos && ((a.cell_seq[c] >> tseq) & 1ull);
        }
        float sc[R];
#pragma unroll
        for (int r = 0; r < R; r++) sc[r] = 0.0f;
        const size_t rowi = (size_t)g * n_ctx + c;
        if (vis) {
            if (a.type_k == T_F16) {
                const uint4 *kp = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.kv.k) + rowi * D + sub * 16);
                const uint4 k0 = kp[0], k1 = kp[1];
                const uint32_t kw[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const float *qq = qf + r * D + sub * 16;
                    float s = 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        s += h2f((uint16_t)(kw[i] & 0xffff)) * qq[2 * i];
                        s += h2f((uint16_t)(kw[i] >> 16)) * qq[2 * i + 1];
                    }
                    sc[r] = s;
                }
            } else if (a.type_k == T_Q8_0) {
                const uint4 kq = *reinterpret_cast<const uint4 *>(a.kv.k + rowi * D + sub * 16);
                const float dk = h2f(a.kv.kd[rowi * NB + (sub >> 1)]);
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint4 qq = *reinterpret_cast<const uint4 *>(qc + r * D + sub * 16);
                    int s = 0;
                    s = dot4(kq.x, qq.x, s); s = dot4(kq.y, qq.y, s); s = dot4(kq.z, qq.z, s); s = dot4(kq.w, qq.w, s);
                    // pair the two halves of the 32-block in integer, then scale: sumi(32) * (dk * dq)
                    const int full = s + __shfl_xor(s, 1, 64);
                    sc[r] = (sub & 1) ? 0.0f : (float)full * (dk * qd[r * NB + (sub >> 1)]);
                }
            } else {  // T_Q4_0: 16 bytes per 32-block; low nibbles = elements 0..15, high = 16..31
                const uint4 kq = *reinterpret_cast<const uint4 *>(a.kv.k + rowi * (D >> 1) + (sub >> 1) * 16);
                const float dk = h2f(a.kv.kd[rowi * NB + (sub >> 1)]);
                const int shift = (sub & 1) * 4;
                const uint32_t n0 = (kq.x >> shift) & 0x0f0f0f0f, n1 = (kq.y >> shift) & 0x0f0f0f0f,
                               n2 = (kq.z >> shift) & 0x0f0f0f0f, n3 = (kq.w >> shift) & 0x0f0f0f0f;
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint4 qq = *reinterpret_cast<const uint4 *>(qc + r * D + sub * 16);
                    int s = 0;
                    s = dot4(n0, qq.x, s); s = dot4(n1, qq.y, s); s = dot4(n2, qq.z, s); s = dot4(n3, qq.w, s);
                    s -= 8 * qs16[r * (D / 16) + sub];
                    const int full = s + __shfl_xor(s, 1, 64);
                    sc[r] = (sub & 1) ? 0.0f : (float)full * dk * qd[r * NB + (sub >> 1)];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            float s = sc[r];
#pragma unroll
            for (int o = 1; o < LPC; o <<= 1) s += __shfl_xor(s, o, 64);
            if (sub == 0 && cl < ncell) S[r * chunk + cl] = vis ? s * a.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- pass 2: m, p, l per query head (wave w handles heads w, w+4, ...)
    for (int r = wave; r < R; r += 4) {
        float m = -INFINITY;
        for (int c = lane; c < ncell; c += 64) m = fmaxf(m, S[r * chunk + c]);
        m = wave_max(m);
        float l = 0.0f;
        for (int c = lane; c < ncell; c += 64) {
            const float s = S[r * chunk + c];
            const float p = (s == -INFINITY) ? 0.0f : expf(s - m);
            S[r * chunk + c] = p;
            l += p;
        }
        l = wave_sum(l);
        if (lane == 0) { red[2 * r] = m; red[2 * r + 1] = l; }
    }
    __syncthreads();

    // ---- pass 3: P.V
    const int dq = tid % DQ, cg = tid / DQ;
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
    for (int cl = cg; cl < ncell; cl += NCG) {
        float p[R];
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; r++) { p[r] = S[r * chunk + cl]; any |= p[r] != 0.0f; }
        if (!any) continue;
        const size_t rowi = (size_t)g * n_ctx + (c_lo + cl);
        float v4[4];
        if (a.type_v == T_F16) {
            const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(a.kv.v) + rowi * D + dq * 4);
            v4[0] = h2f((uint16_t)(w.x & 0xffff)); v4[1] = h2f((uint16_t)(w.x >> 16));
            v4[2] = h2f((uint16_t)(w.y & 0xffff)); v4[3] = h2f((uint16_t)(w.y >> 16));
        } else if (a.type_v == T_Q8_0) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + dq * 4);
            const float dv = h2f(a.kv.vd[rowi * NB + (dq >> 3)]);
            v4[0] = (float)(int8_t)(w & 0xff) * dv; v4[1] = (float)(int8_t)((w >> 8) & 0xff) * dv;
            v4[2] = (float)(int8_t)((w >> 16) & 0xff) * dv; v4[3] = (float)(int8_t)(w >> 24) * dv;
        } else {
            const int i = (dq * 4) & 31;
            const uint32_t w = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * (D >> 1) + (dq >> 3) * 16 + (i & 15));
            const float dv = h2f(a.kv.vd[rowi * NB + (dq >> 3)]);
            const uint32_t nb4 = (i < 16 ? w : (w >> 4)) & 0x0f0f0f0f;
            v4[0] = (float)((int)(nb4 & 0xff) - 8) * dv; v4[1] = (float)((int)((nb4 >> 8) & 0xff) - 8) * dv;
            v4[2] = (float)((int)((nb4 >> 16) & 0xff) - 8) * dv; v4[3] = (float)((int)(nb4 >> 24) - 8) * dv;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            acc[r][0] += v4[0] * p[r]; acc[r][1] += v4[1] * p[r]; acc[r][2] += v4[2] * p[r]; acc[r][3] += v4[3] * p[r];
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        *reinterpret_cast<float4 *>(accs + ((size_t)cg * R + r) * D + dq * 4) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    __syncthreads();
    // partial record per (t, h, split): [D acc][m][l]
    for (int e = tid; e < R * D; e += 256) {
        const int r = e / D, d = e - r * D;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NCG; j++) s += accs[((size_t)j * R + r) * D + d];
        const int h = g * R + r;
        float *dst = a.part + (((size_t)t * H + h) * a.splits + sp) * (D + 2);
        dst[d] = s;
        if (d == 0) { dst[D] = red[2 * r]; dst[D + 1] = red[2 * r + 1]; }
    }
}

// Merge the KV splits.  Workgroup = (256 consecutive output elements = 256/D heads, token t).
//  1. wave w < heads-per-block: lane = split; M = max m, weight_s = exp(m_s - M) / sum_s exp(m_s - M) l_s  -> LDS
//  2. every thread owns one output element and sums its split partials (independent loads, all in flight)
//  3. wave 0 re-reads the 256 merged values from LDS (4 per lane) and, when asked, quantises them for the attn_output
//     mat-vec (saves a launch on the decode path).
__global__ __launch_bounds__(256) void flash_attn_combine_kernel(const float *part, float *out, int H, int D, int splits,
                                                                 ActQuant q, int want_q8k, int want_q80, const int32_t *tok_nsplits, int8_t *bh, int8_t *bl) {
    extern __shared__ float wgt[];                 // [hpb][splits]
    __shared__ __attribute__((aligned(16))) float merged[256];
    const int t = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int E = H * D, nblk = E >> 8;
    const int stride_s = splits;                   // workspace stride; a token of a batched step may use fewer slots
    if (tok_nsplits) splits = tok_nsplits[t];
    const int hpb = 256 / D;                       // heads per block (D = 64 -> 4, D = 128 -> 2)
    const int h0 = b * hpb;
    if (wave < hpb) {
        const float *p = part + ((size_t)t * H + h0 + wave) * stride_s * (D + 2);
        float M = -INFINITY;
        for (int s0 = 0; s0 < splits; s0 += 64) {
            const int sidx = s0 + lane;
            const float m = sidx < splits ? p[(size_t)sidx * (D + 2) + D] : -INFINITY;
            M = fmaxf(M, wave_max(m));
        }
        float den = 0.0f;
        for (int s0 = 0; s0 < splits; s0 += 64) {
            const int sidx = s0 + lane;
            float w = 0.0f, l = 0.0f;
            if (sidx < splits) {
                const float m = p[(size_t)sidx * (D + 2) + D];
                l = p[(size_t)sidx * (D + 2) + D + 1];
                w = (m == -INFINITY) ? 0.0f : expf(m - M);
                wgt[wave * stride_s + sidx] = w;
            }
            den += wave_sum(w * l);
        }
        const float inv = 1.0f / den;
        for (int sidx = lane; sidx < splits; sidx += 64) wgt[wave * stride_s + sidx] *= inv;
    }
    __syncthreads();
    {
        const int e = tid, hl = e / D, d = e - hl * D;
        const float *p = part + ((size_t)t * H + h0 + hl) * stride_s * (D + 2) + d;
        float acc = 0.0f;
        for (int s = 0; s < splits; s++) {
            const float w = wgt[hl * stride_s + s];
            if (w != 0.0f) acc += w * p[(size_t)s * (D + 2)];        // chunks with no visible cell publish only (m, l)
        }
        merged[e] = acc;
        out[(size_t)t * E + b * 256 + e] = acc;
    }
    __syncthreads();
    if (wave == 0 && (want_q8k || want_q80)) {
        const float4 v4 = *reinterpret_cast<const float4 *>(merged + lane * 4);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const int e0 = b * 256 + lane * 4;
        if (want_q8k) {
            uint32_t packed; int bs; float dq;
            wave_quant_q8k(vv, lane, packed, bs, dq);
            *reinterpret_cast<uint32_t *>(q.qs + (size_t)t * E + e0) = packed;
            if ((lane & 3) == 0) {
                const size_t bi = (size_t)t * (E >> 4) + b * 16 + (lane >> 2);
                q.bsums[bi] = (int16_t)bs;
                if (bh) { const int hi = bs >> 6; bh[bi] = (int8_t)hi; bl[bi] = (int8_t)(bs - 64 * hi); }   // mmq_prep_kernel's split, for the MFMA kernels
            }
            if (lane == 0) q.d[(size_t)t * nblk + b] = dq;
        }
        if (want_q80) {
            uint32_t packed; float dd;
            wave_quant_q80(vv, packed, dd);
            *reinterpret_cast<uint32_t *>(q.qs0 + (size_t)t * E + e0) = packed;
            if ((lane & 7) == 0) q.d0[(size_t)t * (E >> 5) + b * 8 + (lane >> 3)] = f2h(dd);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Decode-step variants (T small, D = 128, NORM rope): everything a workgroup needs is requested in ONE round trip.

// K rope + K/V store, 4 consecutive elements per thread (two rope pairs), q8_0 blocks quantised with 8-lane DPP groups.
template <int TK, int TV>
__global__ __launch_bounds__(256) void kv_store_fast_kernel(const float *k, const float *v, int GD, int D, const float *cs_table, int n_rot,
                                                            const int32_t *tok_cell, KVLayerView kv, int n_ctx) {
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int cell = tok_cell[t];
    for (int e0 = tid * 4; e0 < GD; e0 += 1024) {
        float4 kk = *reinterpret_cast<const float4 *>(k + (size_t)t * GD + e0);
        const float4 vv4 = *reinterpret_cast<const float4 *>(v + (size_t)t * GD + e0);
        const int g = e0 / D, dd = e0 - g * D;
        if (dd < n_rot) {
            const float4 cs = *reinterpret_cast<const float4 *>(cs_table + (size_t)t * n_rot + dd);   // c0 s0 c1 s1
            const float x0 = kk.x, x1 = kk.y, x2 = kk.z, x3 = kk.w;
            kk.x = x0 * cs.x - x1 * cs.y; kk.y = x0 * cs.y + x1 * cs.x;
            kk.z = x2 * cs.z - x3 * cs.w; kk.w = x2 * cs.w + x3 * cs.z;
        }
        const size_t rowi = (size_t)g * n_ctx + cell;
        const float ka[4] = {kk.x, kk.y, kk.z, kk.w}, va[4] = {vv4.x, vv4.y, vv4.z, vv4.w};
        if (TK == T_F16) {
            uint2 o; o.x = (uint32_t)f2h(ka[0]) | ((uint32_t)f2h(ka[1]) << 16); o.y = (uint32_t)f2h(ka[2]) | ((uint32_t)f2h(ka[3]) << 16);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(kv.k) + rowi * D + dd) = o;
        } else {
            uint32_t packed; float d;
            wave_quant_q80(ka, packed, d);
            *reinterpret_cast<uint32_t *>(kv.k + rowi * D + dd) = packed;
            if ((lane & 7) == 0) kv.kd[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
        }
        if (TV == T_F16) {
            uint2 o; o.x = (uint32_t)f2h(va[0]) | ((uint32_t)f2h(va[1]) << 16); o.y = (uint32_t)f2h(va[2]) | ((uint32_t)f2h(va[3]) << 16);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(kv.v) + rowi * D + dd) = o;
        } else {
            uint32_t packed; float d;
            wave_quant_q80(va, packed, d);
            *reinterpret_cast<uint32_t *>(kv.v + rowi * D + dd) = packed;
            if ((lane & 7) == 0) kv.vd[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
        }
    }
}

// Prompt batches: the same K / V part (grid row y >= q_parts) plus the rotation of q in place (rows y < q_parts, 1024 elements of the
// token's H * D each), four consecutive elements = two rope pairs per thread.  Same arithmetic as rope_kv_store_kernel (one workgroup per
// token, one element per thread and pass, byte stores into a q8_0 cache: 14 us per layer at 512 tokens against 6).
template <int TK, int TV>
__global__ __launch_bounds__(256) void rope_q_kv_store_fast_kernel(float *q, int HD, int q_parts, const float *k, const float *v, int GD, int D,
                                                                   const float *cs_table, int n_rot, const int32_t *tok_cell, KVLayerView kv, int n_ctx) {
    const int t = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    if (part < q_parts) {
        const int e0 = part * 1024 + tid * 4;
        const int dd = e0 % D;
        if (dd < n_rot) {
            float4 *qp = reinterpret_cast<float4 *>(q + (size_t)t * HD + e0);
            float4 x = *qp;
            const float4 cs = *reinterpret_cast<const float4 *>(cs_table + (size_t)t * n_rot + dd);   // c0 s0 c1 s1
            const float x0 = x.x, x1 = x.y, x2 = x.z, x3 = x.w;
            x.x = x0 * cs.x - x1 * cs.y; x.y = x0 * cs.y + x1 * cs.x;
            x.z = x2 * cs.z - x3 * cs.w; x.w = x2 * cs.w + x3 * cs.z;
            *qp = x;
        }
        return;
    }
    const int cell = tok_cell[t];
    {
        const int e0 = (part - q_parts) * 1024 + tid * 4;
        float4 kk = *reinterpret_cast<const float4 *>(k + (size_t)t * GD + e0);
        const float4 vv4 = *reinterpret_cast<const float4 *>(v + (size_t)t * GD + e0);
        const int g = e0 / D, dd = e0 - g * D;
        if (dd < n_rot) {
            const float4 cs = *reinterpret_cast<const float4 *>(cs_table + (size_t)t * n_rot + dd);   // c0 s0 c1 s1
            const float x0 = kk.x, x1 = kk.y, x2 = kk.z, x3 = kk.w;
            kk.x = x0 * cs.x - x1 * cs.y; kk.y = x0 * cs.y + x1 * cs.x;
            kk.z = x2 * cs.z - x3 * cs.w; kk.w = x2 * cs.w + x3 * cs.z;
        }
        const size_t rowi = (size_t)g * n_ctx + cell;
        const float ka[4] = {kk.x, kk.y, kk.z, kk.w}, va[4] = {vv4.x, vv4.y, vv4.z, vv4.w};
        if (TK == T_F16) {
            uint2 o; o.x = (uint32_t)f2h(ka[0]) | ((uint32_t)f2h(ka[1]) << 16); o.y = (uint32_t)f2h(ka[2]) | ((uint32_t)f2h(ka[3]) << 16);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(kv.k) + rowi * D + dd) = o;
        } else {
            uint32_t packed; float d;
            wave_quant_q80(ka, packed, d);
            *reinterpret_cast<uint32_t *>(kv.k + rowi * D + dd) = packed;
            if ((lane & 7) == 0) kv.kd[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
        }
        if (TV == T_F16) {
            uint2 o; o.x = (uint32_t)f2h(va[0]) | ((uint32_t)f2h(va[1]) << 16); o.y = (uint32_t)f2h(va[2]) | ((uint32_t)f2h(va[3]) << 16);
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(kv.v) + rowi * D + dd) = o;
        } else {
            uint32_t packed; float d;
            wave_quant_q80(va, packed, d);
            *reinterpret_cast<uint32_t *>(kv.v + rowi * D + dd) = packed;
            if ((lane & 7) == 0) kv.vd[rowi * (D >> 5) + (dd >> 5)] = f2h(d);
        }
    }
}

bool rope_q_kv_store_fast_applicable(int H, int G, int D, int type_k, int type_v, const RopeArgs &ra) {
    return kv_store_fast_applicable(G, D, type_k, type_v, ra) && (H * D) % 1024 == 0 && D % 4 == 0;
}
hipError_t launch_rope_q_kv_store_fast(float *q, const float *k, const float *v, int T, int H, int G, int D, const float *cs_table, RopeArgs ra,
                                       const int32_t *tok_cell, KVLayerView kv, int type_k, int type_v, int n_ctx, hipStream_t st) {
    if (!cs_table || !rope_q_kv_store_fast_applicable(H, G, D, type_k, type_v, ra)) return hipErrorInvalidValue;
    const int q_parts = H * D / 1024, kv_parts = G * D / 1024;
    const dim3 grid((unsigned)T, (unsigned)(q_parts + kv_parts));
#define RKS(TK, TV) hipLaunchKernelGGL((rope_q_kv_store_fast_kernel<TK, TV>), grid, dim3(256), 0, st, q, H * D, q_parts, k, v, G * D, D, cs_table, ra.n_rot, tok_cell, kv, n_ctx)
    if (type_k == T_F16 && type_v == T_F16) RKS(T_F16, T_F16);
    else if (type_k == T_Q8_0 && type_v == T_Q8_0) RKS(T_Q8_0, T_Q8_0);
    else if (type_k == T_Q8_0 && type_v == T_F16) RKS(T_Q8_0, T_F16);
    else RKS(T_F16, T_Q8_0);
#undef RKS
    return hipGetLastError();
}

bool kv_store_fast_applicable(int G, int D, int type_k, int type_v, const RopeArgs &ra) {
    return !ra.neox && (G * D) % 1024 == 0 && D % 32 == 0 && (ra.n_rot % 4) == 0 && (type_k == T_F16 || type_k == T_Q8_0) && (type_v == T_F16 || type_v == T_Q8_0);
}
hipError_t launch_kv_store_fast(const float *k, const float *v, int T, int G, int D, const float *cs_table, RopeArgs ra,
                                const int32_t *tok_cell, KVLayerView kv, int type_k, int type_v, int n_ctx, hipStream_t st) {
#define KVS(TK, TV) hipLaunchKernelGGL((kv_store_fast_kernel<TK, TV>), dim3(T), dim3(256), 0, st, k, v, G * D, D, cs_table, ra.n_rot, tok_cell, kv, n_ctx)
    if (type_k == T_F16 && type_v == T_F16) KVS(T_F16, T_F16);
    else if (type_k == T_Q8_0 && type_v == T_Q8_0) KVS(T_Q8_0, T_Q8_0);
    else if (type_k == T_Q8_0 && type_v == T_F16) KVS(T_Q8_0, T_F16);
    else KVS(T_F16, T_Q8_0);
#undef KVS
    return hipGetLastError();
}

}  // namespace mi355
#include "attn_decode_dev.h"   // DecodeFuse, flash_attn_decode_item / flash_attn_decode_kernel
namespace mi355 {

bool flash_attn_decode_applicable(const AttnArgs &a, const RopeArgs &ra) {
    const int R = a.H / a.G;
    const bool pow2 = R == 1 || R == 2 || R == 4 || R == 8;
    // 3, 5, 6, 7 query heads per kv head (Llama-3.2-3B, Qwen2-1.5B / 7B, Yi-34B): head_dim 128 and one cache type for K and V only (fewer instantiations)
    const bool odd = (R == 3 || R == 5 || R == 6 || R == 7) && a.D == 128 && a.type_k == a.type_v;
    // NEOX pairing (qwen2): the kernel rotates whole heads only
    if (ra.neox && ra.n_rot != a.D) return false;
    // a q4_0 cache (K and V): head_dim 128, the power-of-two head ratios
    if (a.type_k == T_Q4_0 || a.type_v == T_Q4_0) return a.type_k == a.type_v && a.D == 128 && pow2 && a.T <= 64 && a.n_kv_max <= 64 * 2048;
    return (a.D == 128 || a.D == 64) && (pow2 || odd) && a.T <= 64 && a.n_kv_max <= 64 * 2048 &&
           (a.type_k == T_F16 || a.type_k == T_Q8_0) && (a.type_v == T_F16 || a.type_v == T_Q8_0);
}
int flash_attn_decode_splits(int n_kv_max) { return n_kv_max > 0 ? (n_kv_max + 63) / 64 : 1; }

// q is the UN-rotated query; a.splits must be flash_attn_decode_splits(a.n_kv_max)
// knew / vnew / tok_cell (all or none): the K rope + KV store of the step's tokens happens inside this launch (every token of the batch
// belongs to a different sequence, so no token reads another one's new cell) instead of in launch_kv_store_fast before it
hipError_t launch_flash_attn_decode(const AttnArgs &a, const float *cs_table, RopeArgs ra, hipStream_t st, const float *knew, const float *vnew,
                                    const int32_t *tok_cell) {
    const int R = a.H / a.G;
    const dim3 grid(a.G, a.splits, a.T);
    DecodeFuse nofz{};
    nofz.neox = ra.neox;
    const bool store = knew && vnew && tok_cell;
    if (store) { nofz.knew = knew; nofz.vnew = vnew; nofz.tok_cell = tok_cell; }
#define FAD_D(RR, TK, TV, DD) do { if (store) hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, true, false, DD>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, nofz); \
                                   else hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, false, false, DD>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, nofz); } while (0)
#define FAD(RR, TK, TV) do { if (a.D == 64) FAD_D(RR, TK, TV, 64); else FAD_D(RR, TK, TV, 128); } while (0)
#define FAD_T(RR)                                                              \
    if (a.type_k == T_Q4_0) FAD_D(RR, T_Q4_0, T_Q4_0, 128);                    \
    else if (a.type_k == T_F16 && a.type_v == T_F16) FAD(RR, T_F16, T_F16);    \
    else if (a.type_k == T_Q8_0 && a.type_v == T_Q8_0) FAD(RR, T_Q8_0, T_Q8_0); \
    else if (a.type_k == T_Q8_0) FAD(RR, T_Q8_0, T_F16);                       \
    else FAD(RR, T_F16, T_Q8_0);
#define FAD_O(RR) if (a.type_k == T_F16) FAD_D(RR, T_F16, T_F16, 128); else FAD_D(RR, T_Q8_0, T_Q8_0, 128);
    switch (R) {
        case 1: FAD_T(1) break;
        case 2: FAD_T(2) break;
        case 4: FAD_T(4) break;
        case 8: FAD_T(8) break;
        case 3: FAD_O(3) break;
        case 5: FAD_O(5) break;
        case 6: FAD_O(6) break;
        case 7: FAD_O(7) break;
        default: return hipErrorInvalidValue;
    }
#undef FAD_O
#undef FAD_T
#undef FAD
#undef FAD_D
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nblk = (a.H * a.D) >> 8;
    ActQuant qq;
    if (a.out_q) qq = *a.out_q;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, a.T), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       qq, (int)(a.out_q && a.out_q8k), (int)(a.out_q && a.out_q80), a.tok_nchunks, a.out_q ? a.out_bh : nullptr, a.out_q ? a.out_bl : nullptr);
    return hipGetLastError();
}

// Single-token step in ONE launch: K rope + KV store + attention + split merge + quantise (see DecodeFuse).
bool flash_attn_decode_fused_applicable(const AttnArgs &a, const RopeArgs &ra) {
    const int R = a.H / a.G;
    // (R * D < 256: 256 / (R * D) neighbouring kv heads share a merge ticket, see attn_decode_dev.h)
    const int rd = R * a.D;
    const int gp = 256 / (rd % 256 == 0 ? 256 : rd % 128 == 0 ? 128 : rd % 64 == 0 ? 64 : 32);      // kv heads per merge ticket (attn_decode_dev.h)
    return a.T == 1 && flash_attn_decode_applicable(a, ra) && a.G % gp == 0 && (ra.n_rot % 4) == 0 && a.splits <= 64;
}
// diagnosis: MI355_ATTN_PROBE=1 makes the fused decode attention stamp its phases (DecodeFuse::probe); the stamps of the
// LAST launch are summarised on stderr by attn_probe_report() (called when a context is destroyed)
static unsigned long long *g_attn_probe = nullptr;
static int g_attn_probe_G = 0, g_attn_probe_splits = 0;
static unsigned long long *attn_probe_buffer() {
    static const bool on = getenv("MI355_ATTN_PROBE") && getenv("MI355_ATTN_PROBE")[0] == '1';
    if (on && !g_attn_probe && hipMalloc((void **)&g_attn_probe, (4096 + 64) * 8) == hipSuccess) (void)hipMemset(g_attn_probe, 0, (4096 + 64) * 8);
    return g_attn_probe;
}
void attn_probe_report() {
    if (!g_attn_probe || g_attn_probe_G <= 0) return;
    std::vector<unsigned long long> t(4096 + 64);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(t.data(), g_attn_probe, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    const int n = std::min(2048, g_attn_probe_G * g_attn_probe_splits);
    unsigned long long t0 = ~0ull, s_hi = 0, e_hi = 0;
    double dur = 0;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        if (!t[2 * i] || !t[2 * i + 1]) continue;
        t0 = std::min(t0, t[2 * i]); s_hi = std::max(s_hi, t[2 * i]); e_hi = std::max(e_hi, t[2 * i + 1]);
        dur += (double)(t[2 * i + 1] - t[2 * i]) * 0.01; cnt++;
    }
    if (!cnt) return;
    fprintf(stderr, "attn probe: %d workgroups, starts spread %.2f us, mean item %.2f us, last partial stored at %.2f us\n", cnt,
            (double)(s_hi - t0) * 0.01, dur / cnt, (double)(e_hi - t0) * 0.01);
    for (int g = 0; g < g_attn_probe_G && g < 8; g++) {
        const unsigned long long *m = &t[4096 + 8 * g];
        if (!m[0]) continue;
        fprintf(stderr, "  head %d merge: ticket at %.2f us, weights +%.2f, sums +%.2f, stores +%.2f -> done at %.2f us\n", g, (double)(m[0] - t0) * 0.01,
                (double)(m[1] - m[0]) * 0.01, (double)(m[2] - m[1]) * 0.01, (double)(m[3] - m[2]) * 0.01, (double)(m[3] - t0) * 0.01);
    }
}

hipError_t launch_flash_attn_decode_fused(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew,
                                          const int32_t *tok_cell, unsigned *counters, hipStream_t st) {
    const int R = a.H / a.G;
    const dim3 grid(a.G, a.splits, 1);
    g_attn_probe_G = a.G; g_attn_probe_splits = a.splits;
    DecodeFuse fz{};
    fz.knew = knew; fz.vnew = vnew; fz.tok_cell = tok_cell; fz.counters = counters;
    fz.neox = ra.neox;
    if (a.out_q) fz.q = *a.out_q;
    fz.want_q8k = (int)(a.out_q && a.out_q8k); fz.want_q80 = (int)(a.out_q && a.out_q80);
    fz.probe = attn_probe_buffer();
    // The chunk partials are published with write-through stores and a relaxed ticket instead of an agent-scope release /
    // acquire per workgroup: that pair is a write-back + invalidate of the whole L2, and hundreds of workgroups issuing it at
    // once serialise (tools/bench_gridbar.hip: 25 us for 512).  Measured on the 8B step: 424 -> 448 tok/s at 4000 cells.
    // MI355_ATTN_COH=0 restores the fenced form.
    static const bool coh = !(getenv("MI355_ATTN_COH") && getenv("MI355_ATTN_COH")[0] == '0');
#define FAD_D(RR, TK, TV, DD) do { if (coh && counters) hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, true, true, DD>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, fz); \
                                   else hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, true, false, DD>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, fz); } while (0)
#define FAD(RR, TK, TV) do { if (a.D == 64) FAD_D(RR, TK, TV, 64); else FAD_D(RR, TK, TV, 128); } while (0)
#define FAD_T(RR)                                                              \
    if (a.type_k == T_Q4_0) FAD_D(RR, T_Q4_0, T_Q4_0, 128);                    \
    else if (a.type_k == T_F16 && a.type_v == T_F16) FAD(RR, T_F16, T_F16);    \
    else if (a.type_k == T_Q8_0 && a.type_v == T_Q8_0) FAD(RR, T_Q8_0, T_Q8_0); \
    else if (a.type_k == T_Q8_0) FAD(RR, T_Q8_0, T_F16);                       \
    else FAD(RR, T_F16, T_Q8_0);
#define FAD_O(RR) if (a.type_k == T_F16) FAD_D(RR, T_F16, T_F16, 128); else FAD_D(RR, T_Q8_0, T_Q8_0, 128);
    switch (R) {
        case 1: FAD_T(1) break;
        case 2: FAD_T(2) break;
        case 4: FAD_T(4) break;
        case 8: FAD_T(8) break;
        case 3: FAD_O(3) break;
        case 5: FAD_O(5) break;
        case 6: FAD_O(6) break;
        case 7: FAD_O(7) break;
        default: return hipErrorInvalidValue;
    }
#undef FAD_O
#undef FAD_T
#undef FAD
#undef FAD_D
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || counters) return e;
    const int nblk = (a.H * a.D) >> 8;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, 1), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       fz.q, fz.want_q8k, fz.want_q80, (const int32_t *)nullptr, (int8_t *)nullptr, (int8_t *)nullptr);
    return hipGetLastError();
}

// The same merge with ONE wave per 256-element block of the output row, for prompt batches (round 5): the kernel above spends a 256-thread workgroup, three
// barrier phases and an LDS round trip on 3 KB of data - 8192 such workgroups for a 512-token prompt, 17 us.  Here a wave computes the softmax weights of its
// block's heads the same way (lane = split: the same wave_max / wave_sum, so the same bits), hands them to its lanes with readlane, sums the partial rows four
// elements per lane in split order and quantises the block it holds - no barrier, no LDS.  Up to 64 splits.
template <int D>
__global__ __launch_bounds__(256) void flash_attn_combine_wave_kernel(const float *__restrict__ part, float *__restrict__ out, int H, int splits, int nblk,
                                                                      ActQuant q, int want_q8k, int want_q80, int8_t *__restrict__ bh, int8_t *__restrict__ bl) {
    constexpr int HPB = 256 / D;
    const int t = blockIdx.y, lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;                                       // wave-uniform
    const int E = H * D, h0 = b * HPB;
    float wv[HPB];                                               // lane s: the weight of split s, per head of the block
#pragma unroll
    for (int hh = 0; hh < HPB; hh++) {
        const float *p = part + ((size_t)t * H + h0 + hh) * splits * (D + 2);
        const float m = lane < splits ? p[(size_t)lane * (D + 2) + D] : -INFINITY;
        const float M = fmaxf(-INFINITY, wave_max(m));
        float w = 0.0f, l = 0.0f;
        if (lane < splits) {
            l = p[(size_t)lane * (D + 2) + D + 1];
            w = (m == -INFINITY) ? 0.0f : expf(m - M);
        }
        float den = 0.0f;
        den += wave_sum(w * l);
        const float inv = 1.0f / den;
        wv[hh] = w * inv;
    }
    const int e = lane * 4, hl = e / D, d = e - hl * D;
    const float *pr = part + ((size_t)t * H + h0 + hl) * splits * (D + 2) + d;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < splits; s++) {
        float w = 0.0f;
#pragma unroll
        for (int hh = 0; hh < HPB; hh++) { const float wh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[hh]), s)); if (hh == hl) w = wh; }
        if (w != 0.0f) {                                         // chunks with no visible cell publish only (m, l)
            const float2 a0 = *reinterpret_cast<const float2 *>(pr + (size_t)s * (D + 2)), a1 = *reinterpret_cast<const float2 *>(pr + (size_t)s * (D + 2) + 2);
            acc[0] += w * a0.x; acc[1] += w * a0.y; acc[2] += w * a1.x; acc[3] += w * a1.y;
        }
    }
    const int e0 = b * 256 + e;
    *reinterpret_cast<float4 *>(out + (size_t)t * E + e0) = float4{acc[0], acc[1], acc[2], acc[3]};
    if (want_q8k) {
        uint32_t packed; int bs; float dq;
        wave_quant_q8k(acc, lane, packed, bs, dq);
        *reinterpret_cast<uint32_t *>(q.qs + (size_t)t * E + e0) = packed;
        if ((lane & 3) == 0) {
            const size_t bi = (size_t)t * (E >> 4) + b * 16 + (lane >> 2);
            q.bsums[bi] = (int16_t)bs;
            if (bh) { const int hi = bs >> 6; bh[bi] = (int8_t)hi; bl[bi] = (int8_t)(bs - 64 * hi); }
        }
        if (lane == 0) q.d[(size_t)t * nblk + b] = dq;
    }
    if (want_q80) {
        uint32_t packed; float dd;
        wave_quant_q80(acc, packed, dd);
        *reinterpret_cast<uint32_t *>(q.qs0 + (size_t)t * E + e0) = packed;
        if ((lane & 7) == 0) q.d0[(size_t)t * (E >> 5) + b * 8 + (lane >> 3)] = f2h(dd);
    }
}

// (D + 32 floats per record: what attn_out.hip keeps per head and chunk - whole 128-byte lines; the other kernels use D + 2 of them)
size_t flash_attn_workspace_floats(int T, int H, int D, int splits) { return (size_t)T * H * splits * (D + 32); }

hipError_t launch_flash_attn_combine(const AttnArgs &a, int splits, hipStream_t st) {
    if (a.D != 64 && a.D != 128) return hipErrorInvalidValue;
    const int nblk = (a.H * a.D) >> 8;
    ActQuant qq;
    if (a.out_q) qq = *a.out_q;
    static const bool wave_off = getenv("MI355_ATTN_COMBINE_WAVE") && getenv("MI355_ATTN_COMBINE_WAVE")[0] == '0';
    if (!wave_off && a.T >= 32 && splits <= 64) {                 // prompt batches: one wave per block
        const dim3 grid((unsigned)((nblk + 3) / 4), (unsigned)a.T);
        if (a.D == 128) hipLaunchKernelGGL(flash_attn_combine_wave_kernel<128>, grid, dim3(256), 0, st, a.part, a.out, a.H, splits, nblk, qq, (int)(a.out_q && a.out_q8k),
                                           (int)(a.out_q && a.out_q80), a.out_q ? a.out_bh : nullptr, a.out_q ? a.out_bl : nullptr);
        else hipLaunchKernelGGL(flash_attn_combine_wave_kernel<64>, grid, dim3(256), 0, st, a.part, a.out, a.H, splits, nblk, qq, (int)(a.out_q && a.out_q8k),
                                (int)(a.out_q && a.out_q80), a.out_q ? a.out_bh : nullptr, a.out_q ? a.out_bl : nullptr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, a.T), dim3(256), (size_t)(256 / a.D) * splits * 4, st, a.part, a.out, a.H, a.D, splits,
                       qq, (int)(a.out_q && a.out_q8k), (int)(a.out_q && a.out_q80), (const int32_t *)nullptr, a.out_q ? a.out_bh : nullptr, a.out_q ? a.out_bl : nullptr);
    return hipGetLastError();
}

int flash_attn_pick_splits(int T, int G, int n_kv_max) {
    int min_splits = (n_kv_max + ATT_MAX_CHUNK - 1) / ATT_MAX_CHUNK;
    if (min_splits < 1) min_splits = 1;
    // enough workgroups to cover the chip, at least 64 cells per split
    int want = (num_cu() * 2) / (T * G > 0 ? T * G : 1);
    int by_len = (n_kv_max + 63) / 64;
    if (want > by_len) want = by_len;
    if (want > 64) want = 64;                 // the merge kernel holds one split per lane
    if (want < min_splits) want = min_splits;
    if (want < 1) want = 1;
    return want;
}

template <int D, int R>
static hipError_t launch_fa(const AttnArgs &a, hipStream_t st) {
    int chunk = (a.n_kv_max + a.splits - 1) / a.splits;
    chunk = (chunk + 31) & ~31;
    if (chunk > ATT_MAX_CHUNK) chunk = ATT_MAX_CHUNK;
    constexpr int DQ = D / 4, NCG = 256 / DQ;
    const size_t lds = sizeof(float) * ((size_t)R * chunk + (size_t)R * D) + (size_t)R * D + sizeof(float) * R * (D / 32) +
                       sizeof(int) * R * (D / 16) + sizeof(float) * 2 * R + sizeof(float) * (size_t)NCG * R * D;
    hipLaunchKernelGGL((flash_attn_split_kernel<D, R>), dim3(a.G, a.T, a.splits), dim3(256), lds, st, a);
    return hipGetLastError();
}

// ---- opt-in: an f16 cache read with the arithmetic of the reference's CPU path (ggml_compute_forward_flash_attn_ext_f16;
// the reference's default cache is f16: /root/reference/src/llama_engine.cc:628-637).  That path visits the visible cells ONE AT A TIME in cell order: the score
// is a double-precision sum of the 128 f16 x f16 products in element order, the online softmax rescales the accumulator whenever the running maximum moves, and
// V is accumulated in FP16 - every `acc += v * p` and every rescale rounds to half precision.  The kernels above accumulate V in f32 over chunks in parallel,
// so their f16-cache output sits ~1e-2 from the CPU's (the noise of 4096 half-precision roundings), while matching its f32-accumulating restatement to 1e-6.
// This kernel reproduces the CPU sequence itself; it is three phases per 2048-cell stretch of the cache, one workgroup per (head, token):
//   1. scores: a thread owns whole cells - the same double accumulation in the same element order as the CPU, hence the same bits;
//   2. the online-softmax factors of every cell (ms: what the accumulator is scaled by, vs: the weight of the cell's V row) by a prefix maximum over the
//      visible cells - they depend on the scores only, not on the accumulator;
//   3. the accumulation proper, sequential in the cell index, one thread per output element: acc = half(float(acc) * ms); acc = half(float(acc) + v * vs),
//      V staged through LDS 64 cells at a time.
// What can still differ from the CPU is expf (the device's against the host libm's, both within an ulp of f32): a flip of the last f32 bit of a weight moves a
// half-precision rounding once in ~10^4 operations.  Cost: phase 3 is a dependent chain of two roundings per cell - 27 us per layer at 4096 cells, against 13
// for the parallel kernels - so this is a parity mode (MI355_FA_V_ACC=f16 or the debug option "fa_v_acc_f16"), not the default.
static int g_fa_v16 = -1;
void set_fa_v_acc_f16(int on) { g_fa_v16 = on; }
bool fa_v_acc_f16_enabled() {
    static const bool env = getenv("MI355_FA_V_ACC") && !strcmp(getenv("MI355_FA_V_ACC"), "f16");
    return g_fa_v16 >= 0 ? g_fa_v16 != 0 : env;
}
constexpr int V16_CHUNK = 2048;
template <int D>
__global__ __launch_bounds__(256) void flash_attn_v16_kernel(AttnArgs a) {
    __shared__ float s_vs[V16_CHUNK];            // phase 1: score (NaN = not visible); phase 2 on: vs
    __shared__ float s_ms[V16_CHUNK];            // ms
    __shared__ float s_tmax[256];
    __shared__ __attribute__((aligned(16))) uint16_t s_q[D];
    __shared__ __attribute__((aligned(16))) uint16_t s_v[64 * D];     // phase 3: 64 cells of this kv head's V rows
    const int h = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const int g = h / (a.H / a.G);
    int n_kv = *a.n_kv_dev;
    if (n_kv > a.n_kv_max) n_kv = a.n_kv_max;
    const int tpos = a.tok_pos[t], tseq = a.tok_seq[t];
    if (tid < D) s_q[tid] = f2h(a.q[((size_t)t * a.H + h) * D + tid]);      // the query row in K's vec_dot type: f16, round to nearest even
    __syncthreads();
    const uint16_t *kbase = reinterpret_cast<const uint16_t *>(a.kv.k) + (size_t)g * a.n_ctx * D;
    const uint16_t *vbase = reinterpret_cast<const uint16_t *>(a.kv.v) + (size_t)g * a.n_ctx * D;
    float M = -INFINITY, S = 0.0f;               // (every thread carries them: the same values everywhere)
    uint16_t acc = 0;                            // thread d < D: element d of the accumulator, as fp16 bits
    for (int c0 = 0; c0 < n_kv; c0 += V16_CHUNK) {
        const int nc = min(V16_CHUNK, n_kv - c0);
        // ---- 1. scores
        for (int i = tid; i < nc; i += 256) {
            const int c = c0 + i;
            const int cp = a.cell_pos[c];
            const bool vis = cp >= 0 && cp <= tpos && ((a.cell_seq[c] >> tseq) & 1ull);
            float sc = __builtin_nanf("");
            if (vis) {
                const uint4 *kr = reinterpret_cast<const uint4 *>(kbase + (size_t)c * D);
                double sum = 0.0;
#pragma unroll 4
                for (int e8 = 0; e8 < D / 8; e8++) {
                    const uint4 kv = kr[e8];
                    const uint4 qv = *reinterpret_cast<const uint4 *>(s_q + e8 * 8);
                    const unsigned kw[4] = {kv.x, kv.y, kv.z, kv.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        sum += (double)(h2f((uint16_t)(kw[w] & 0xffffu)) * h2f((uint16_t)(qw[w] & 0xffffu)));
                        sum += (double)(h2f((uint16_t)(kw[w] >> 16)) * h2f((uint16_t)(qw[w] >> 16)));
                    }
                }
                sc = (float)sum * a.scale;
            }
            s_vs[i] = sc;
        }
        __syncthreads();
        // ---- 2. softmax factors.  Thread tid owns cells [tid * per, +per): its maximum, the maxima before it, then its cells in order
        const int per = (nc + 255) / 256;
        const int i0 = min(tid * per, nc), i1 = min(i0 + per, nc);
        float lm = -INFINITY;
        for (int i = i0; i < i1; i++) { const float sc = s_vs[i]; if (sc == sc) lm = fmaxf(lm, sc); }
        s_tmax[tid] = lm;
        __syncthreads();
        float Mp = M;                            // maximum over the visible cells before this thread's first
        for (int j = 0; j < tid; j++) Mp = fmaxf(Mp, s_tmax[j]);
        float Mall = M;
        for (int j = 0; j < 256; j++) Mall = fmaxf(Mall, s_tmax[j]);
        for (int i = i0; i < i1; i++) {
            const float sc = s_vs[i];
            float ms = 1.0f, vs = 0.0f;                             // (not visible: leaves the accumulator and the sum as they are)
            if (sc == sc) {
                if (sc > Mp) { ms = expf(Mp - sc); vs = 1.0f; Mp = sc; }
                else { ms = 1.0f; vs = expf(sc - Mp); }
            }
            s_ms[i] = ms; s_vs[i] = vs;
        }
        M = Mall;
        __syncthreads();
        // ---- 3. the accumulation, in cell order.  V reaches LDS 64 cells at a time (all 256 threads, 16-byte loads; the next 64 cells are in flight while
        // the chain works through the current ones: a chain that waited for its own 2-byte loads took 2 ms per layer at 4000 cells)
        constexpr int SUB = 64, NV4 = SUB * D * 2 / 16 / 256;       // uint4 per thread and sub-chunk (D = 128: 4, D = 64: 2)
        const uint4 *vg = reinterpret_cast<const uint4 *>(vbase + (size_t)c0 * D);
        const int n_v4 = nc * D * 2 / 16;                           // uint4 of this stretch
        uint4 stage[NV4];
#pragma unroll
        for (int k = 0; k < NV4; k++) { const int j = k * 256 + tid; stage[k] = j < n_v4 ? vg[j] : uint4{0, 0, 0, 0}; }
        for (int i0s = 0; i0s < nc; i0s += SUB) {
#pragma unroll
            for (int k = 0; k < NV4; k++) reinterpret_cast<uint4 *>(s_v)[k * 256 + tid] = stage[k];
            __syncthreads();
            if (i0s + SUB < nc) {
                const int base = (i0s + SUB) * D * 2 / 16;
#pragma unroll
                for (int k = 0; k < NV4; k++) { const int j = base + k * 256 + tid; stage[k] = j < n_v4 ? vg[j] : uint4{0, 0, 0, 0}; }
            }
            if (tid < D) {
                const int ne = min(SUB, nc - i0s);
                // (no branch in the chain, so that the LDS reads of the next cells are issued ahead of it: a cell that is not visible has ms = 1, vs = 0 and
                // contributes +0 - x * 1 and x + 0 are x itself)
#pragma unroll 8
                for (int u = 0; u < ne; u++) {
                    const float ms = s_ms[i0s + u], vs = s_vs[i0s + u];
                    const float vv = h2f(s_v[u * D + tid]);
                    acc = f2h(h2f(acc) * ms);                        // (ms == 1: the value itself)
                    const float prod = vs != 0.0f ? vv * vs : 0.0f;  // (a hole may hold anything, Inf included)
                    acc = f2h(h2f(acc) + prod);
                    const float Sm = S * ms;
                    S = Sm + vs;
                }
            }
            __syncthreads();
        }
    }
    if (tid < D) {
        const float inv = 1.0f / S;
        a.out[((size_t)t * a.H + h) * D + tid] = h2f(acc) * inv;
    }
}
static hipError_t launch_fa_v16(const AttnArgs &a, hipStream_t st) {
    if (a.type_k != T_F16 || a.type_v != T_F16 || (a.D != 128 && a.D != 64) || a.H % a.G) return hipErrorInvalidValue;
    if (a.D == 128) hipLaunchKernelGGL(flash_attn_v16_kernel<128>, dim3(a.H, a.T), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(flash_attn_v16_kernel<64>, dim3(a.H, a.T), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.out_q) return launch_quantize(a.out, a.H * a.D, a.T, *a.out_q, a.out_q8k, a.out_q80, st, a.out_bh, a.out_bl);
    return hipSuccess;
}

hipError_t launch_flash_attn(const AttnArgs &a, hipStream_t st) {
    if (fa_v_acc_f16_enabled() && a.type_k == T_F16 && a.type_v == T_F16) return launch_fa_v16(a, st);
    static const bool mfma_prefill = !(getenv("MI355_ATTN_PREFILL_MFMA") && getenv("MI355_ATTN_PREFILL_MFMA")[0] == '0');
    if (mfma_prefill && flash_attn_prefill_applicable(a)) return launch_flash_attn_prefill(a, st);   // prompt processing: matrix cores
    const int R = a.H / a.G;
    hipError_t e = hipErrorInvalidValue;
    if ((size_t)a.splits * ATT_MAX_CHUNK < (size_t)a.n_kv_max) return hipErrorInvalidValue;
#define FA_CASE(DD, RR) if (a.D == DD && R == RR) e = launch_fa<DD, RR>(a, st);
    FA_CASE(128, 1) FA_CASE(128, 2) FA_CASE(128, 4) FA_CASE(128, 8)
    FA_CASE(64, 1) FA_CASE(64, 2) FA_CASE(64, 4) FA_CASE(64, 8)
    // query / kv head ratios the tuned kernels are not cut for (Llama-3.2-3B: 3, Yi-34B: 7, ...): this general kernel is the only path
    FA_CASE(128, 3) FA_CASE(128, 5) FA_CASE(128, 6) FA_CASE(128, 7)
    FA_CASE(64, 3) FA_CASE(64, 5) FA_CASE(64, 6) FA_CASE(64, 7)
#undef FA_CASE
    if (e != hipSuccess) return e;
    const int nblk = (a.H * a.D) >> 8;
    ActQuant qq;
    if (a.out_q) qq = *a.out_q;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, a.T), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       qq, (int)(a.out_q && a.out_q8k), (int)(a.out_q && a.out_q80), a.tok_nchunks, a.out_q ? a.out_bh : nullptr, a.out_q ? a.out_bl : nullptr);
    return hipGetLastError();
}

}  // namespace mi355
