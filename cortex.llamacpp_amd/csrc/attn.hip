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
#include <cstdlib>

#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

// ------------------------------------------------------------------------------------------
// cos/sin table of one position: theta_0 = pos, theta_{i+1} = theta_i * theta_scale, iterated in f32
// exactly like ggml_rope_cache_init, so the angles match the CPU path bit for bit (cosf/sinf within libm ulp).
__device__ __forceinline__ void rope_angle(int i_pair, int32_t pos, float theta_scale, float freq_scale, const float *ff,
                                           float &c, float &s) {
    float theta = (float)pos;
    for (int k = 0; k < i_pair; k++) theta *= theta_scale;
    const float f = ff ? ff[i_pair] : 1.0f;
    const float th = freq_scale * (theta / f);
    c = cosf(th);
    s = sinf(th);
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
    rope_angle(i, tok_pos[t], theta_scale, ra.freq_scale, ra.freq_factors, c, s);
    cs_out[(size_t)t * ra.n_rot + 2 * i] = c;
    cs_out[(size_t)t * ra.n_rot + 2 * i + 1] = s;
}
hipError_t launch_rope_table(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, hipStream_t st) {
    const float theta_scale = powf(ra.freq_base, -2.0f / (float)ra.n_rot);
    const int n = T * (ra.n_rot >> 1);
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 63) / 64), dim3(64), 0, st, tok_pos, T, ra, theta_scale, cs_out);
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
        for (int i = tid; i < (ra.n_rot >> 1); i += 256) rope_angle(i, pos, theta_scale, ra.freq_scale, ra.freq_factors, cs[2 * i], cs[2 * i + 1]);
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
    for (int i = tid; i < (ra.n_rot >> 1); i += 256) rope_angle(i, dl, theta_scale, ra.freq_scale, ra.freq_factors, cs[2 * i], cs[2 * i + 1]);
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
                                   const uint64_t *tok_seqmask, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) {
        cell_pos[tok_cell[t]] = tok_pos[t];
        cell_seq[tok_cell[t]] = tok_seqmask[t];
    }
}
hipError_t launch_kv_meta_set(int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell, const int32_t *tok_pos,
                              const uint64_t *tok_seqmask, int T, hipStream_t st) {
    hipLaunchKernelGGL(kv_meta_set_kernel, dim3((T + 255) / 256), dim3(256), 0, st, cell_pos, cell_seq, tok_cell, tok_pos, tok_seqmask, T);
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
                                                                 ActQuant q, int want_q8k, int want_q80, const int32_t *tok_nsplits) {
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
            if ((lane & 3) == 0) q.bsums[(size_t)t * (E >> 4) + b * 16 + (lane >> 2)] = (int16_t)bs;
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

// Decode attention: workgroup = (kv head g, chunk of 64 cells, token t).  The query heads of the group are read
// un-rotated, rotated with the cos/sin table and converted to the K cache's dot type here (so no separate pass over q),
// and K / V / scales / cell table of the chunk are all requested before the first wait.
//
// FUSED (single-token step, R*D a multiple of 256) folds the two neighbouring launches into this one:
//  * KV store: the workgroup whose chunk holds the token's cell rotates K, converts K/V to the cache type, writes the
//    cache row and patches its own registers / LDS scales with the same codes (no other workgroup reads that cell);
//  * merge: every workgroup publishes its chunk partial, then takes a ticket on a per-kv-head counter; the LAST arriver
//    merges the splits of its R heads exactly as flash_attn_combine_kernel does, writes the f32 rows and the quantised
//    activation of the attn_output mat-vec, and re-arms the counter.  Nobody waits: no spin, no co-residency needed.
struct DecodeFuse {
    const float *knew, *vnew;      // [G*D] un-rotated K and V of the token
    const int32_t *tok_cell;       // [1]
    unsigned *counters;            // [G], zero between launches
    ActQuant q;
    int want_q8k, want_q80;
};

template <int R, int TK, int TV, bool FUSED>
__global__ __launch_bounds__(256) void flash_attn_decode_kernel(const AttnArgs a, const float *cs_table, int n_rot, const DecodeFuse fz) {
    constexpr int D = 128, C = 64, NB = 4;
    constexpr int KP = TK == T_F16 ? 4 : 2;                 // 16-byte K pieces per thread
    constexpr int LPC = TK == T_F16 ? 16 : 8;               // lanes per cell in the score pass
    __shared__ __attribute__((aligned(16))) float qf[R * D];         // rotated q (f16-rounded for an f16 cache)
    __shared__ __attribute__((aligned(16))) int8_t qc[R * D];        // q8_0 codes of q
    __shared__ float qd[R * NB];
    __shared__ float S[R * C];
    __shared__ float ml[R * 2];
    __shared__ int vis[C];
    __shared__ uint32_t ksc[C * NB / 2], vsc[C * NB / 2];           // f16 block scales of the chunk
    __shared__ __attribute__((aligned(16))) float accs[8 * R * D];
    __shared__ __attribute__((aligned(16))) uint8_t newk[D * 2], newv[D * 2];   // the token's own cache row (codes or f16)
    __shared__ uint32_t newkd[2], newvd[2];                                      // its f16 block scales, two per word
    __shared__ int last_flag;
    const int g = blockIdx.x, sp = blockIdx.y, t = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_ctx = a.n_ctx, H = a.H;
    int chunk = sp;
    if (a.tok_chunks) {                                        // walk this token's chunk list only (sequences own cache regions)
        if (sp >= a.tok_nchunks[t]) return;
        chunk = a.tok_chunks[(size_t)t * a.chunk_stride + sp];
    }
    const int c_lo = chunk * C;
    const size_t head_row0 = (size_t)g * n_ctx;

    // ---- issue every global load of the workgroup
    int cpos = -1;
    unsigned long long cseq = 0;
    if (tid < C && c_lo + tid < n_ctx) { cpos = a.cell_pos[c_lo + tid]; cseq = a.cell_seq[c_lo + tid]; }
    const int32_t tpos = a.tok_pos[t];
    const int tseq = a.tok_seq[t];
    if (!FUSED && a.T > 1) {
        // batched steps (one token per sequence): most chunks hold other sequences' cells only.  Decide that before
        // touching K / V: such a chunk publishes (m, l) = (-inf, 0) and leaves; the merge skips it.
        const int mine = (tid < C && cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
        if (!__syncthreads_or(mine)) {
            if (tid < R) {
                float *dst = a.part + (((size_t)t * H + (size_t)g * R + tid) * a.splits + sp) * (D + 2);
                dst[D] = -INFINITY; dst[D + 1] = 0.0f;
            }
            return;
        }
    }
    // q pairs: R*64 pairs, pair pp -> (head r = pp / 64, i = pp % 64); NORM pairing (2i, 2i+1)
    constexpr int NPAIR = R * 64, PPT = (NPAIR + 255) / 256;
    float2 qv[PPT], csv[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int pp = tid + 256 * j;
        if (pp < NPAIR) {
            const int r = pp >> 6, i = pp & 63;
            qv[j] = *reinterpret_cast<const float2 *>(a.q + ((size_t)t * H + (size_t)g * R + r) * D + 2 * i);
            csv[j] = (2 * i < n_rot) ? *reinterpret_cast<const float2 *>(cs_table + (size_t)t * n_rot + 2 * i) : make_float2(1.0f, 0.0f);
        }
    }
    uint4 kreg[KP];
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + 256 * j;
        int cell = c_lo + p / LPC;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TK == T_F16) kreg[j] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.kv.k) + rowi * D + (p % LPC) * 8);
        else kreg[j] = *reinterpret_cast<const uint4 *>(a.kv.k + rowi * D + (p % LPC) * 16);
    }
    const int dq = tid & 31, cg = tid >> 5;
    uint2 vreg[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int cell = c_lo + cg + 8 * i;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TV == T_F16) vreg[i] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(a.kv.v) + rowi * D + dq * 4);
        else { vreg[i].x = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + dq * 4); vreg[i].y = 0; }
    }
    uint32_t ks2 = 0, vs2 = 0;
    if (tid < C * NB / 2) {
        int cell = c_lo + tid / 2;
        if (cell >= n_ctx) cell = n_ctx - 1;
        if (TK != T_F16) ks2 = *reinterpret_cast<const uint32_t *>(a.kv.kd + (head_row0 + cell) * NB + (tid & 1) * 2);
        if (TV != T_F16) vs2 = *reinterpret_cast<const uint32_t *>(a.kv.vd + (head_row0 + cell) * NB + (tid & 1) * 2);
    }

    // ---- FUSED: this token's K/V row (wave 0: lanes 0..31 rotate + convert K, lanes 32..63 convert V; 4 elements each)
    int own_cl = -1;                                   // chunk-local index of the token's cell, if this chunk holds it
    if (FUSED) {
        const int cellnew = fz.tok_cell[0];
        if (cellnew >= c_lo && cellnew < c_lo + C) own_cl = cellnew - c_lo;
        if (own_cl >= 0 && wave == 0) {
            const bool isk = lane < 32;
            const int dd = (lane & 31) * 4;
            float4 x4 = *reinterpret_cast<const float4 *>((isk ? fz.knew : fz.vnew) + (size_t)g * D + dd);
            if (isk && dd < n_rot) {
                const float4 cs = *reinterpret_cast<const float4 *>(cs_table + dd);   // c0 s0 c1 s1
                const float x0 = x4.x, x1 = x4.y, x2 = x4.z, x3 = x4.w;
                x4.x = x0 * cs.x - x1 * cs.y; x4.y = x0 * cs.y + x1 * cs.x;
                x4.z = x2 * cs.z - x3 * cs.w; x4.w = x2 * cs.w + x3 * cs.z;
            }
            const float xa[4] = {x4.x, x4.y, x4.z, x4.w};
            const size_t rowi = head_row0 + cellnew;
            const int TT = isk ? TK : TV;              // (TK, TV are compile-time; the select folds per branch below)
            uint32_t packed = 0; float dsc = 0.0f;
            if (TK != T_F16 || TV != T_F16) wave_quant_q80(xa, packed, dsc);   // whole wave takes part (8-lane groups)
            if (TT == T_F16) {
                uint2 o; o.x = (uint32_t)f2h(xa[0]) | ((uint32_t)f2h(xa[1]) << 16); o.y = (uint32_t)f2h(xa[2]) | ((uint32_t)f2h(xa[3]) << 16);
                *reinterpret_cast<uint2 *>((isk ? newk : newv) + dd * 2) = o;
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(isk ? a.kv.k : a.kv.v) + rowi * D + dd) = o;
            } else {
                *reinterpret_cast<uint32_t *>((isk ? newk : newv) + dd) = packed;
                *reinterpret_cast<uint32_t *>((isk ? a.kv.k : a.kv.v) + rowi * D + dd) = packed;
                if ((lane & 7) == 0) {
                    const uint16_t hd = f2h(dsc);
                    (isk ? a.kv.kd : a.kv.vd)[rowi * NB + (dd >> 5)] = hd;
                    reinterpret_cast<uint16_t *>(isk ? newkd : newvd)[dd >> 5] = hd;
                }
            }
        }
    }

    // ---- q: rotate, convert
    if (tid < C) vis[tid] = (cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
    if (tid < C * NB / 2) { ksc[tid] = ks2; vsc[tid] = vs2; }
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int pp = tid + 256 * j;
        if (pp < NPAIR) {
            const float x0 = qv[j].x, x1 = qv[j].y, c = csv[j].x, s = csv[j].y;
            float y0 = x0 * c - x1 * s, y1 = x0 * s + x1 * c;
            if (TK == T_F16) { y0 = h2f(f2h(y0)); y1 = h2f(f2h(y1)); }
            *reinterpret_cast<float2 *>(qf + 2 * pp) = make_float2(y0, y1);
        }
    }
    __syncthreads();
    if (FUSED && own_cl >= 0) {   // workgroup-uniform: use the row just produced instead of what the cache held before
        if (tid < 2) {
            if (TK != T_F16) ksc[own_cl * 2 + tid] = newkd[tid];
            if (TV != T_F16) vsc[own_cl * 2 + tid] = newvd[tid];
        }
#pragma unroll
        for (int j = 0; j < KP; j++) {
            const int p = tid + 256 * j;
            if (p / LPC == own_cl) kreg[j] = *reinterpret_cast<const uint4 *>(newk + (p % LPC) * 16);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (cg + 8 * i == own_cl) {
                if (TV == T_F16) vreg[i] = *reinterpret_cast<const uint2 *>(newv + dq * 8);
                else { vreg[i].x = *reinterpret_cast<const uint32_t *>(newv + dq * 4); vreg[i].y = 0; }
            }
        }
    }
    if (TK != T_F16) {   // q8_0 of the rotated q: 4 values per thread, 8-lane groups
        for (int e0 = tid * 4; e0 < R * D; e0 += 1024) {
            const float4 v4 = *reinterpret_cast<const float4 *>(qf + e0);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(qc + e0) = packed;
            if ((lane & 7) == 0) qd[e0 >> 5] = h2f(f2h(d));
        }
        __syncthreads();
    }

    // ---- scores
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + 256 * j;
        const int cl = p / LPC, piece = p % LPC;
        float sc[R];
        if (TK == T_F16) {
            const uint32_t kw[4] = {kreg[j].x, kreg[j].y, kreg[j].z, kreg[j].w};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float *qq = qf + r * D + piece * 8;
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    s += h2f((uint16_t)(kw[i] & 0xffff)) * qq[2 * i];
                    s += h2f((uint16_t)(kw[i] >> 16)) * qq[2 * i + 1];
                }
                s += dpp_f<DPP_QP_1032>(s); s += dpp_f<DPP_QP_2301>(s); s += dpp_f<DPP_HALF_MIRROR>(s); s += dpp_f<DPP_MIRROR>(s);
                sc[r] = s;
            }
        } else {
            const uint32_t kpair = ksc[cl * 2 + (piece >> 2)];
            const float dk = h2f((uint16_t)(((piece >> 1) & 1) ? (kpair >> 16) : (kpair & 0xffff)));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint4 qq = *reinterpret_cast<const uint4 *>(qc + r * D + piece * 16);
                int s = 0;
                s = dot4(kreg[j].x, qq.x, s); s = dot4(kreg[j].y, qq.y, s); s = dot4(kreg[j].z, qq.z, s); s = dot4(kreg[j].w, qq.w, s);
                s += dpp_i<DPP_QP_1032>(s);                    // both halves of the 32-block (integer)
                float f = (piece & 1) ? 0.0f : (float)s * (dk * qd[r * NB + (piece >> 1)]);
                f += dpp_f<DPP_QP_1032>(f); f += dpp_f<DPP_QP_2301>(f); f += dpp_f<DPP_HALF_MIRROR>(f);   // 8 lanes
                sc[r] = f;
            }
        }
        if (piece == 0) {
            const bool v = vis[cl] != 0;
#pragma unroll
            for (int r = 0; r < R; r++) S[r * C + cl] = v ? sc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- softmax of the chunk: wave w -> heads w, w+4, ...; lane = cell
    for (int r = wave; r < R; r += 4) {
        const float s = S[r * C + lane];
        const float m = wave_max(s);
        const float p = (s == -INFINITY) ? 0.0f : expf(s - m);
        const float l = wave_sum(p);
        S[r * C + lane] = p;
        if (lane == 0) { ml[2 * r] = m; ml[2 * r + 1] = l; }
    }
    __syncthreads();

    // ---- P.V
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int cl = cg + 8 * i;
        float v4[4];
        if (TV == T_F16) {
            v4[0] = h2f((uint16_t)(vreg[i].x & 0xffff)); v4[1] = h2f((uint16_t)(vreg[i].x >> 16));
            v4[2] = h2f((uint16_t)(vreg[i].y & 0xffff)); v4[3] = h2f((uint16_t)(vreg[i].y >> 16));
        } else {
            const uint32_t vpair = vsc[cl * 2 + (dq >> 4)];
            const float dv = h2f((uint16_t)(((dq >> 3) & 1) ? (vpair >> 16) : (vpair & 0xffff)));
            const uint32_t w = vreg[i].x;
            v4[0] = (float)(int8_t)(w & 0xff) * dv; v4[1] = (float)(int8_t)((w >> 8) & 0xff) * dv;
            v4[2] = (float)(int8_t)((w >> 16) & 0xff) * dv; v4[3] = (float)(int8_t)(w >> 24) * dv;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float p = S[r * C + cl];
            acc[r][0] += v4[0] * p; acc[r][1] += v4[1] * p; acc[r][2] += v4[2] * p; acc[r][3] += v4[3] * p;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        *reinterpret_cast<float4 *>(accs + ((size_t)cg * R + r) * D + dq * 4) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    __syncthreads();
    for (int e = tid; e < R * D; e += 256) {
        const int r = e / D, d = e - r * D;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) s += accs[((size_t)j * R + r) * D + d];
        float *dst = a.part + (((size_t)t * H + (size_t)g * R + r) * a.splits + sp) * (D + 2);
        dst[d] = s;
        if (d == 0) { dst[D] = ml[2 * r]; dst[D + 1] = ml[2 * r + 1]; }
    }
    if (!FUSED) return;
    if (fz.counters == nullptr) return;               // store-fused only: the merge runs as its own launch

    // ---- FUSED: ticket; the last workgroup of this kv head merges
    const int stride_s = a.splits;                             // workspace stride; with a chunk list fewer slots are in use
    const int splits = a.tok_nchunks ? a.tok_nchunks[0] : a.splits;
    // every wave's partial stores must have reached L2 before thread 0 releases them device-wide: __syncthreads() fences
    // LDS only (hipcc emitted no vmcnt wait before the barrier here), so each wave drains its own stores first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a workgroup-scope release fence compiles to nothing on this target)
    __syncthreads();
    if (tid == 0) {
        // ONE release per workgroup (the barrier ordered the other waves' stores before it), then the ticket
        const unsigned old = __hip_atomic_fetch_add(fz.counters + g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (old == (unsigned)splits - 1u) ? 1 : 0;
        if (last_flag) __hip_atomic_store(fz.counters + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    if (!last_flag) return;
    float *wgt = S;                                    // [R][64] split weights (S is free now: R * C floats)
    float *merged = accs;                              // [R * D]
    for (int r = wave; r < R; r += 4) {                // same arithmetic as flash_attn_combine_kernel
        const float *p = a.part + ((size_t)g * R + r) * stride_s * (D + 2);
        float m = -INFINITY, l = 0.0f;
        if (lane < splits) {
            m = __hip_atomic_load(p + (size_t)lane * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l = __hip_atomic_load(p + (size_t)lane * (D + 2) + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float M = wave_max(m);
        const float w = (lane < splits && m != -INFINITY) ? expf(m - M) : 0.0f;
        const float den = wave_sum(w * l);
        const float inv = 1.0f / den;
        wgt[r * 64 + lane] = w * inv;
    }
    __syncthreads();
    const int E = H * D;
    for (int e = tid; e < R * D; e += 256) {
        const int r = e / D, d = e - r * D;
        const float *p = a.part + ((size_t)g * R + r) * stride_s * (D + 2) + d;
        float acc = 0.0f;
#pragma unroll 8
        for (int s2 = 0; s2 < splits; s2++)
            acc += wgt[r * 64 + s2] * __hip_atomic_load(p + (size_t)s2 * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        merged[e] = acc;
        a.out[(size_t)g * R * D + e] = acc;
    }
    __syncthreads();
    constexpr int NBLK = (R * D) >> 8;                 // 256-blocks this kv head owns in the H*D row
    if (wave < NBLK && (fz.want_q8k || fz.want_q80)) {
        for (int b = wave; b < NBLK; b += 4) {
            const float4 v4 = *reinterpret_cast<const float4 *>(merged + b * 256 + lane * 4);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            const int gb = ((g * R * D) >> 8) + b;     // global block index
            const int e0 = gb * 256 + lane * 4;
            if (fz.want_q8k) {
                uint32_t packed; int bs; float dq8;
                wave_quant_q8k(vv, lane, packed, bs, dq8);
                *reinterpret_cast<uint32_t *>(fz.q.qs + e0) = packed;
                if ((lane & 3) == 0) fz.q.bsums[gb * 16 + (lane >> 2)] = (int16_t)bs;
                if (lane == 0) fz.q.d[gb] = dq8;
            }
            if (fz.want_q80) {
                uint32_t packed; float dd;
                wave_quant_q80(vv, packed, dd);
                *reinterpret_cast<uint32_t *>(fz.q.qs0 + e0) = packed;
                if ((lane & 7) == 0) fz.q.d0[gb * 8 + (lane >> 3)] = f2h(dd);
            }
        }
    }
    (void)E;
}

bool flash_attn_decode_applicable(const AttnArgs &a, const RopeArgs &ra) {
    const int R = a.H / a.G;
    return a.D == 128 && !ra.neox && (R == 1 || R == 2 || R == 4 || R == 8) && a.T <= 64 && a.n_kv_max <= 64 * 2048 &&
           (a.type_k == T_F16 || a.type_k == T_Q8_0) && (a.type_v == T_F16 || a.type_v == T_Q8_0);
}
int flash_attn_decode_splits(int n_kv_max) { return n_kv_max > 0 ? (n_kv_max + 63) / 64 : 1; }

// q is the UN-rotated query; a.splits must be flash_attn_decode_splits(a.n_kv_max)
hipError_t launch_flash_attn_decode(const AttnArgs &a, const float *cs_table, RopeArgs ra, hipStream_t st) {
    const int R = a.H / a.G;
    const dim3 grid(a.G, a.splits, a.T);
    const DecodeFuse nofz{};
#define FAD(RR, TK, TV) hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, false>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, nofz)
#define FAD_T(RR)                                                              \
    if (a.type_k == T_F16 && a.type_v == T_F16) FAD(RR, T_F16, T_F16);         \
    else if (a.type_k == T_Q8_0 && a.type_v == T_Q8_0) FAD(RR, T_Q8_0, T_Q8_0); \
    else if (a.type_k == T_Q8_0) FAD(RR, T_Q8_0, T_F16);                       \
    else FAD(RR, T_F16, T_Q8_0);
    switch (R) {
        case 1: FAD_T(1) break;
        case 2: FAD_T(2) break;
        case 4: FAD_T(4) break;
        case 8: FAD_T(8) break;
        default: return hipErrorInvalidValue;
    }
#undef FAD_T
#undef FAD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int nblk = (a.H * a.D) >> 8;
    ActQuant qq;
    if (a.out_q) qq = *a.out_q;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, a.T), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       qq, (int)(a.out_q && a.out_q8k), (int)(a.out_q && a.out_q80), a.tok_nchunks);
    return hipGetLastError();
}

// Single-token step in ONE launch: K rope + KV store + attention + split merge + quantise (see DecodeFuse).
bool flash_attn_decode_fused_applicable(const AttnArgs &a, const RopeArgs &ra) {
    const int R = a.H / a.G;
    return a.T == 1 && flash_attn_decode_applicable(a, ra) && (R == 2 || R == 4 || R == 8) && (ra.n_rot % 4) == 0 && a.splits <= 64;
}
hipError_t launch_flash_attn_decode_fused(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew,
                                          const int32_t *tok_cell, unsigned *counters, hipStream_t st) {
    const int R = a.H / a.G;
    const dim3 grid(a.G, a.splits, 1);
    DecodeFuse fz{};
    fz.knew = knew; fz.vnew = vnew; fz.tok_cell = tok_cell; fz.counters = counters;
    if (a.out_q) fz.q = *a.out_q;
    fz.want_q8k = (int)(a.out_q && a.out_q8k); fz.want_q80 = (int)(a.out_q && a.out_q80);
#define FAD(RR, TK, TV) hipLaunchKernelGGL((flash_attn_decode_kernel<RR, TK, TV, true>), grid, dim3(256), 0, st, a, cs_table, ra.n_rot, fz)
#define FAD_T(RR)                                                              \
    if (a.type_k == T_F16 && a.type_v == T_F16) FAD(RR, T_F16, T_F16);         \
    else if (a.type_k == T_Q8_0 && a.type_v == T_Q8_0) FAD(RR, T_Q8_0, T_Q8_0); \
    else if (a.type_k == T_Q8_0) FAD(RR, T_Q8_0, T_F16);                       \
    else FAD(RR, T_F16, T_Q8_0);
    switch (R) {
        case 2: FAD_T(2) break;
        case 4: FAD_T(4) break;
        case 8: FAD_T(8) break;
        default: return hipErrorInvalidValue;
    }
#undef FAD_T
#undef FAD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || counters) return e;
    const int nblk = (a.H * a.D) >> 8;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, 1), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       fz.q, fz.want_q8k, fz.want_q80, (const int32_t *)nullptr);
    return hipGetLastError();
}

size_t flash_attn_workspace_floats(int T, int H, int D, int splits) { return (size_t)T * H * splits * (D + 2); }

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

hipError_t launch_flash_attn(const AttnArgs &a, hipStream_t st) {
    static const bool mfma_prefill = !(getenv("MI355_ATTN_PREFILL_MFMA") && getenv("MI355_ATTN_PREFILL_MFMA")[0] == '0');
    if (mfma_prefill && flash_attn_prefill_applicable(a)) return launch_flash_attn_prefill(a, st);   // prompt processing: matrix cores
    const int R = a.H / a.G;
    hipError_t e = hipErrorInvalidValue;
    if ((size_t)a.splits * ATT_MAX_CHUNK < (size_t)a.n_kv_max) return hipErrorInvalidValue;
#define FA_CASE(DD, RR) if (a.D == DD && R == RR) e = launch_fa<DD, RR>(a, st);
    FA_CASE(128, 1) FA_CASE(128, 2) FA_CASE(128, 4) FA_CASE(128, 8)
    FA_CASE(64, 1) FA_CASE(64, 2) FA_CASE(64, 4) FA_CASE(64, 8)
#undef FA_CASE
    if (e != hipSuccess) return e;
    const int nblk = (a.H * a.D) >> 8;
    ActQuant qq;
    if (a.out_q) qq = *a.out_q;
    hipLaunchKernelGGL(flash_attn_combine_kernel, dim3(nblk, a.T), dim3(256), (size_t)(256 / a.D) * a.splits * 4, st, a.part, a.out, a.H, a.D, a.splits,
                       qq, (int)(a.out_q && a.out_q8k), (int)(a.out_q && a.out_q80), a.tok_nchunks);
    return hipGetLastError();
}

}  // namespace mi355
