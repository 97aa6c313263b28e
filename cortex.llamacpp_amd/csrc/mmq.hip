// mmq.hip — batched prefill contraction  Y[t][r] = dot(W[r,:], act_q8K[t,:])  on the MFMA matrix cores.
//
// Stands in for ggml's mul_mat_q / the CPU backend's chunked mul_mat for prompt processing,
// reached from the reference through llama_decode with n_tokens > 1 (src/llama_server_context.cc:1568-1607, 1635).
//
// Integer-exact by construction: every 32-weight sub-block is ONE v_mfma_i32_32x32x32_i8 (int8 weights unpacked from
// the 4/5/6-bit codes x int8 Q8_K activation codes -> int32), the 6-bit sub-block scales are applied to the int32
// tile with 24-bit integer multiply-adds, and the Q4_K/Q5_K "mins" term  sum_j m_j * bsum_j  is two more small MFMAs on
// the int16 block sums split into (hi, lo) int8 planes (bsum = 64*hi + lo).  So per (token, row, super-block) the
// integers  isum = sum_j sc_j * (q . q8)  and  msum = sum_j m_j * bsum_j  are bit-identical to ggml_vec_dot_q*_K_q8_K;
// only the final f32 accumulation order over super-blocks differs.
//
// Tile mapping (wave64, 32x32x32): M = tokens, N = weight rows.  The MFMA result lane holds ONE weight row
// (n = lane & 31) and 16 tokens, so the row's d / dmin / sub-block scales are per-lane scalars.  A workgroup = 8 waves
// = 128 rows x 128 tokens; the activation tile of one super-block (128 x 256 int8, + block sums + scales) is staged in
// LDS once per workgroup and read as MFMA A operands by all eight waves (4 row tiles x 2 token-tile pairs) (row stride padded against bank conflicts).
#include "kernels.h"

namespace mi355 {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int TOK_TILE = 128, ROW_TILE = 128, A_STRIDE = 272;   // 256 codes + 16 B pad per token row

__device__ __forceinline__ i32x4 as_i32x4(uint4 v) { i32x4 r; r.x = (int)v.x; r.y = (int)v.y; r.z = (int)v.z; r.w = (int)v.w; return r; }
__device__ __forceinline__ i32x16 mfma_i8(i32x4 a, i32x4 b, i32x16 c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ uint4 ldg16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }

struct Smem {
    int8_t *aq;        // [TOK_TILE][A_STRIDE]  codes of the current super-block
    int8_t *bh, *bl;   // [TOK_TILE][8]         block-sum planes (hi, lo) of the 8 sub-blocks
    float *yd;         // [TOK_TILE]            Q8_K scale of the super-block
};

// per-lane weight-row state of one super-block
template <int TYPE> struct RowSB;

template <> struct RowSB<T_Q4_K> {
    uint4 q[4];            // chunk c: 16 bytes holding 16 low nibbles (sub-block 2c) and 16 high nibbles (2c+1)
    uint32_t sc_lo, sc_hi, mn_lo, mn_hi;   // 8 scales / 8 mins, one byte each
    float d, dmin;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg) {
        const uint8_t *b = row + (size_t)sb * 144;
        const uint4 h = ldg16(b);
#pragma unroll
        for (int c = 0; c < 4; c++) q[c] = ldg16(b + 16 + 32 * c + 16 * kg);
        d = h2f((uint16_t)(h.x & 0xffff)); dmin = h2f((uint16_t)(h.x >> 16));
        // scales bytes s0..s11 in h.y h.z h.w; j<4: sc = s[j]&63, m = s[j+4]&63; j>=4: sc = (s[j+4]&15)|((s[j-4]>>6)<<4), m = (s[j+4]>>4)|((s[j]>>6)<<4)
        sc_lo = h.y & 0x3f3f3f3f; mn_lo = h.z & 0x3f3f3f3f;
        sc_hi = (h.w & 0x0f0f0f0f) | ((h.y >> 2) & 0x30303030);
        mn_hi = ((h.w >> 4) & 0x0f0f0f0f) | ((h.z >> 2) & 0x30303030);
    }
    __device__ __forceinline__ int scale(int j) const { return (int)(((j < 4 ? sc_lo : sc_hi) >> (8 * (j & 3))) & 0xff); }
    __device__ __forceinline__ i32x4 bop(int j) const {     // MFMA B operand of sub-block j: 16 weights of this lane's row
        const uint4 v = q[j >> 1];
        const int sh = (j & 1) * 4;
        i32x4 r;
        r.x = (int)((v.x >> sh) & 0x0f0f0f0f); r.y = (int)((v.y >> sh) & 0x0f0f0f0f);
        r.z = (int)((v.z >> sh) & 0x0f0f0f0f); r.w = (int)((v.w >> sh) & 0x0f0f0f0f);
        return r;
    }
};

template <> struct RowSB<T_Q5_K> {
    uint4 q[4], qh;
    uint32_t sc_lo, sc_hi, mn_lo, mn_hi;
    float d, dmin;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg) {
        const uint8_t *b = row + (size_t)sb * 176;
        const uint4 h = ldg16(b);
        qh = ldg16(b + 16 + 16 * kg);
#pragma unroll
        for (int c = 0; c < 4; c++) q[c] = ldg16(b + 48 + 32 * c + 16 * kg);
        d = h2f((uint16_t)(h.x & 0xffff)); dmin = h2f((uint16_t)(h.x >> 16));
        sc_lo = h.y & 0x3f3f3f3f; mn_lo = h.z & 0x3f3f3f3f;
        sc_hi = (h.w & 0x0f0f0f0f) | ((h.y >> 2) & 0x30303030);
        mn_hi = ((h.w >> 4) & 0x0f0f0f0f) | ((h.z >> 2) & 0x30303030);
    }
    __device__ __forceinline__ int scale(int j) const { return (int)(((j < 4 ? sc_lo : sc_hi) >> (8 * (j & 3))) & 0xff); }
    __device__ __forceinline__ i32x4 bop(int j) const {
        const uint4 v = q[j >> 1];
        const int sh = (j & 1) * 4;
        i32x4 r;
        r.x = (int)(((v.x >> sh) & 0x0f0f0f0f) | (((qh.x >> j) & 0x01010101u) << 4));
        r.y = (int)(((v.y >> sh) & 0x0f0f0f0f) | (((qh.y >> j) & 0x01010101u) << 4));
        r.z = (int)(((v.z >> sh) & 0x0f0f0f0f) | (((qh.z >> j) & 0x01010101u) << 4));
        r.w = (int)(((v.w >> sh) & 0x0f0f0f0f) | (((qh.w >> j) & 0x01010101u) << 4));
        return r;
    }
};

// Q6_K device row planes: ql | qh | scales | d.  K-step s (0..7) = half n = s>>2, quarter k = s&3: 32 weights whose
// first 16 (kg = 0) use scale 8n+2k and last 16 (kg = 1) scale 8n+2k+1, so each K-step is issued as TWO MFMAs with the
// other half of the B operand zeroed.
template <> struct RowSB<T_Q6_K> {
    uint4 ql[4], qhv[2];
    uint4 scv;             // 16 int8 scales
    float d;
    int kg;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg_) {
        kg = kg_;
#pragma unroll
        for (int n = 0; n < 2; n++) {
            ql[2 * n]     = ldg16(row + (size_t)sb * 128 + 64 * n + 16 * kg);        // l = 16kg..   (quarters 0 / 2)
            ql[2 * n + 1] = ldg16(row + (size_t)sb * 128 + 64 * n + 32 + 16 * kg);   // l+32         (quarters 1 / 3)
            qhv[n] = ldg16(row + (size_t)nb * 128 + (size_t)sb * 64 + 32 * n + 16 * kg);
        }
        scv = ldg16(row + (size_t)nb * 192 + (size_t)sb * 16);
        d = h2f(*reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2));
    }
    __device__ __forceinline__ int scale16(int g) const {   // signed int8 scale of 16-group g
        const uint32_t w = g < 4 ? scv.x : g < 8 ? scv.y : g < 12 ? scv.z : scv.w;
        return (int)(int8_t)((w >> (8 * (g & 3))) & 0xff);
    }
    __device__ __forceinline__ i32x4 bop(int s) const {     // signed (q - 32) codes of K-step s for this lane's 16 weights
        const int n = s >> 2, k = s & 3;
        const uint4 l4 = ql[2 * n + (k & 1)];
        const uint4 h4 = qhv[n];
        const int nsh = (k >> 1) * 4, hsh = 2 * k;
        i32x4 r;
#define Q6B(lv, hv) (int)((((((lv) >> nsh) & 0x0f0f0f0f) | ((((hv) >> hsh) & 0x03030303u) << 4)) + 0x60606060u) ^ 0x80808080u)
        r.x = Q6B(l4.x, h4.x); r.y = Q6B(l4.y, h4.y); r.z = Q6B(l4.z, h4.z); r.w = Q6B(l4.w, h4.w);
#undef Q6B
        return r;
    }
};

template <int TYPE>
__global__ __launch_bounds__(512, 1) void mmq_kernel(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T,
                                                     const int8_t *aq, const float *ad, const int8_t *abh, const int8_t *abl,
                                                     float *out, int ld_out, const float *resid) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Smem S;
    S.aq = reinterpret_cast<int8_t *>(smem);
    S.bh = S.aq + TOK_TILE * A_STRIDE;
    S.bl = S.bh + TOK_TILE * 8;
    S.yd = reinterpret_cast<float *>(S.bl + TOK_TILE * 8);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = K >> 8;
    const int row0 = blockIdx.x * ROW_TILE + (wave & 3) * 32;   // 8 waves: 4 row tiles x 2 token-tile pairs
    const int th = wave >> 2;
    const int tok0 = blockIdx.y * TOK_TILE;
    const int n = lane & 31, kg = lane >> 5;
    int my_row = row0 + n;
    const bool row_ok = my_row < n_rows;
    if (!row_ok) my_row = n_rows - 1;
    const uint8_t *rowp = W + (size_t)my_row * row_bytes;
    // activation planes are allocated for a whole number of token tiles (rows past T hold stale data whose results are
    // never stored), so every address below is  uniform base + small 32-bit per-thread offset  (no per-thread 64-bit
    // pointers for the compiler to hoist and spill)
    const int8_t *aq_t = aq + (size_t)tok0 * K;
    const float *ad_t = ad + (size_t)tok0 * nb;
    const int8_t *bh_t = abh + (size_t)tok0 * nb * 8, *bl_t = abl + (size_t)tok0 * nb * 8;
    const unsigned st_off = (unsigned)(tid >> 4) * (unsigned)K + (unsigned)(tid & 15) * 16u;   // staging: token tid/16 (+32 i), piece tid%16
    const unsigned st_lds = (unsigned)(tid >> 4) * A_STRIDE + (unsigned)(tid & 15) * 16u;

    float facc[2][16];
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
        for (int r = 0; r < 16; r++) facc[tt][r] = 0.0f;

    for (int sb = 0; sb < nb; sb++) {
        RowSB<TYPE> R;
        R.load(rowp, nb, sb, kg);                    // weight loads in flight while the activation tile is staged
        __syncthreads();                             // previous super-block fully consumed
        {   // stage 128 tokens x 256 codes: 2048 x 16 B, 8 per thread, all loads issued before the LDS writes
            const int8_t *src = aq_t + (size_t)sb * 256;
            uint4 tmp[4];
#pragma unroll
            for (int i = 0; i < 4; i++) tmp[i] = ldg16(src + st_off + (unsigned)i * 32u * (unsigned)K);
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<uint4 *>(S.aq + st_lds + i * 32 * A_STRIDE) = tmp[i];
        }
        if (tid < TOK_TILE) {
            S.yd[tid] = ad_t[(unsigned)tid * (unsigned)nb + (unsigned)sb];
            if (TYPE != T_Q6_K) {
                *reinterpret_cast<uint2 *>(S.bh + tid * 8) = *reinterpret_cast<const uint2 *>(bh_t + ((unsigned)tid * (unsigned)nb + (unsigned)sb) * 8u);
                *reinterpret_cast<uint2 *>(S.bl + tid * 8) = *reinterpret_cast<const uint2 *>(bl_t + ((unsigned)tid * (unsigned)nb + (unsigned)sb) * 8u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
            const int tq = 2 * th + tt;                                            // token tile of this wave
            const int8_t *arow = S.aq + (tq * 32 + n) * A_STRIDE + 16 * kg;     // A operand: token m = lane & 31
            i32x16 isum;
#pragma unroll
            for (int r = 0; r < 16; r++) isum[r] = 0;
            if constexpr (TYPE == T_Q6_K) {
                const RowSB<T_Q6_K> &R6 = R;
#pragma unroll
                for (int s = 0; s < 8; s++) {
                    const i32x4 a = as_i32x4(*reinterpret_cast<const uint4 *>(arow + 32 * s));
                    const i32x4 b = R6.bop(s);
                    const i32x4 z = {0, 0, 0, 0};
                    i32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; r++) zero[r] = 0;
                    const i32x16 p0 = mfma_i8(a, kg == 0 ? b : z, zero);       // 16-group 2s   (k 0..15)
                    const i32x16 p1 = mfma_i8(a, kg == 1 ? b : z, zero);       // 16-group 2s+1 (k 16..31)
                    const int s0 = R6.scale16(2 * s), s1 = R6.scale16(2 * s + 1);
#pragma unroll
                    for (int r = 0; r < 16; r++) isum[r] += __mul24(s0, p0[r]) + __mul24(s1, p1[r]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    facc[tt][r] += (R6.d * S.yd[tq * 32 + m]) * (float)isum[r];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const i32x4 a = as_i32x4(*reinterpret_cast<const uint4 *>(arow + 32 * j));
                    i32x16 zero;
#pragma unroll
                    for (int r = 0; r < 16; r++) zero[r] = 0;
                    const i32x16 p = mfma_i8(a, R.bop(j), zero);
                    const int sc = R.scale(j);
#pragma unroll
                    for (int r = 0; r < 16; r++) isum[r] += __mul24(sc, p[r]);
                    __builtin_amdgcn_sched_barrier(0);              // keep one MFMA result live (register pressure)
                }
                // mins: msum[m][n] = sum_j m_j[n] * bsum_j[m],  bsum = 64*hi + lo; K = 8 of 32 used (lanes kg == 0, first 8 bytes)
                i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0}, bm = {0, 0, 0, 0};
                if (kg == 0) {
                    const uint2 h2 = *reinterpret_cast<const uint2 *>(S.bh + (tq * 32 + n) * 8);
                    const uint2 l2 = *reinterpret_cast<const uint2 *>(S.bl + (tq * 32 + n) * 8);
                    ah.x = (int)h2.x; ah.y = (int)h2.y; al.x = (int)l2.x; al.y = (int)l2.y;
                    bm.x = (int)R.mn_lo; bm.y = (int)R.mn_hi;
                }
                i32x16 zero;
#pragma unroll
                for (int r = 0; r < 16; r++) zero[r] = 0;
                const i32x16 mh = mfma_i8(ah, bm, zero), mlo = mfma_i8(al, bm, zero);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const float yd = S.yd[tq * 32 + m];
                    const int msum = 64 * mh[r] + mlo[r];
                    facc[tt][r] += (R.d * yd) * (float)isum[r] - (R.dmin * yd) * (float)msum;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // store: lane = weight row n, regs = tokens; 32 consecutive rows per token -> 128-B coalesced
    if (row_ok) {
#pragma unroll
        for (int tt = 0; tt < 2; tt++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int gt = tok0 + (2 * th + tt) * 32 + m;
                if (gt < T) {
                    const size_t o = (size_t)gt * ld_out + row0 + n;
                    out[o] = resid ? resid[o] + facc[tt][r] : facc[tt][r];
                }
            }
    }
}

// block sums of 32 codes split into int8 planes: bsum = 64*hi + lo, hi in [-64, 63], lo in [0, 63]
__global__ void mmq_prep_kernel(const int16_t *bsums, int K, int T, int8_t *bh, int8_t *bl) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over T * K/32
    const size_t n = (size_t)T * (K >> 5);
    if (i >= n) return;
    const int s = (int)bsums[2 * i] + (int)bsums[2 * i + 1];
    const int hi = s >> 6;                    // arithmetic shift: floor(s / 64)
    bh[i] = (int8_t)hi;
    bl[i] = (int8_t)(s - 64 * hi);
}

}  // namespace

bool mmq_applicable(int type, int K, int T) {
    return (type == T_Q4_K || type == T_Q5_K || type == T_Q6_K) && (K % 256) == 0 && T >= 32;
}
size_t mmq_prep_bytes(int K, int T) { return (size_t)T * (K >> 5); }

hipError_t launch_mmq_prep(const ActQuant &q, int K, int T, int8_t *bh, int8_t *bl, hipStream_t st) {
    const size_t n = (size_t)T * (K >> 5);
    hipLaunchKernelGGL(mmq_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q.bsums, K, T, bh, bl);
    return hipGetLastError();
}

// out[t * ld_out + r] = (resid ? resid[...] : 0) + dot(W[r], act[t]);  bh/bl from launch_mmq_prep
hipError_t launch_mmq(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q,
                      const int8_t *bh, const int8_t *bl, float *out, int ld_out, const float *resid, hipStream_t st) {
    const dim3 grid((n_rows + ROW_TILE - 1) / ROW_TILE, (T + TOK_TILE - 1) / TOK_TILE);
    const size_t lds = (size_t)TOK_TILE * A_STRIDE + TOK_TILE * 16 + TOK_TILE * 4;
    switch (type) {
        case T_Q4_K: hipLaunchKernelGGL(mmq_kernel<T_Q4_K>, grid, dim3(512), lds, st, W, row_bytes, n_rows, K, T, q.qs, q.d, bh, bl, out, ld_out, resid); break;
        case T_Q5_K: hipLaunchKernelGGL(mmq_kernel<T_Q5_K>, grid, dim3(512), lds, st, W, row_bytes, n_rows, K, T, q.qs, q.d, bh, bl, out, ld_out, resid); break;
        case T_Q6_K: hipLaunchKernelGGL(mmq_kernel<T_Q6_K>, grid, dim3(512), lds, st, W, row_bytes, n_rows, K, T, q.qs, q.d, bh, bl, out, ld_out, resid); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mi355
