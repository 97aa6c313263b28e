// mmvq.hip — quantised mat-vec  y[t][r] = dot(W[r,:], act_q8[t,:])  for the decode path.
//
// Stands in for ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K / ggml_vec_dot_q8_0_q8_0 as driven by
// ggml_compute_forward_mul_mat (upstream ggml-cpu; absent from /root/reference — SURVEY.md §8a rows a8/a9),
// reached from the reference through llama_decode (src/llama_server_context.cc:1635).
//
// Design (gfx950, wave64, HBM-bound):
//  * one wave owns a PAIR of weight rows at a time and keeps their partial sums in registers;
//  * every lane issues 16-byte coalesced loads: 8 consecutive lanes cover the 128 quant bytes of one
//    256-weight super-block, 8 super-blocks per wave pass; headers (d, dmin, 6-bit scales) are one more
//    16-byte load that the 8 lanes of a super-block share through the coalescer;
//  * 2 rows x 2 passes are in flight before the first use (>= 8 x 16 B per lane outstanding);
//  * the int8 activation vector (Q8_K: int8 + f32 d per 256 + int16 sums per 16; Q8_0: int8 + f16 d per 32)
//    is staged once per workgroup into LDS and read with ds_read_b128;
//  * integer work is v_dot4_i32_i8; integer partial sums are exactly the CPU backend's, only the final
//    f32 summation order differs (lane partials are reduced with a wave butterfly).
//  * epilogues fuse the residual add (attn_output / ffn_down) and SwiGLU (ffn_gate + ffn_up).
#include "kernels.h"

namespace mi355 {

// ------------------------------------------------------------------------------------------
// LDS view of the staged activations of one workgroup
struct ActLds {
    const int8_t *qs;      // [T][K]      q8_K codes
    const float *d;        // [T][K/256]
    const int16_t *bs;     // [T][K/16]
    const int8_t *qs0;     // [T][K]      q8_0 codes
    const uint16_t *d0;    // [T][K/32]   f16
    int K;
};

__device__ __forceinline__ uint4 ld16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }

template <int TYPE> struct Item;

// ---- Q4_K : ggml super-block  [d f16][dmin f16][scales 12][qs 128]  (144 B, 16-B aligned) ----
template <> struct Item<T_Q4_K> {
    static constexpr int EPP = 2048;  // elements per wave pass
    uint4 q;
    int sc_lo, sc_hi, m_lo, m_hi;
    float d, dmin;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane, uint4 &hdr) {
        sb = pass * 8 + (lane >> 3);
        valid = sb < (K >> 8);
        if (valid) {
            const uint8_t *b = row + (size_t)sb * 144;
            hdr = ld16(b);
            q = ld16(b + 16 + (lane & 7) * 16);
        }
    }
    __device__ __forceinline__ void prep(const uint4 &hdr, int lane) {
        if (!valid) return;
        const int v = lane & 7, c = v >> 1;
        d = h2f((uint16_t)(hdr.x & 0xffff));
        dmin = h2f((uint16_t)(hdr.x >> 16));
        const int sh = (c & 1) * 16;
        const uint32_t a16 = (hdr.y >> sh) & 0xffff, b16 = (hdr.z >> sh) & 0xffff, c16 = (hdr.w >> sh) & 0xffff;
        uint32_t sc, mn;
        if (c < 2) {
            sc = a16 & 0x3f3f;
            mn = b16 & 0x3f3f;
        } else {
            sc = (c16 & 0x0f0f) | ((a16 >> 2) & 0x3030);
            mn = ((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030);
        }
        sc_lo = sc & 0xff; sc_hi = sc >> 8;
        m_lo = mn & 0xff;  m_hi = mn >> 8;
    }
    // integer partials of this lane for token t
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, c = v >> 1, h = v & 1;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 64 * c + 16 * h;
        const uint4 lo = ld16(a), hi = ld16(a + 32);
        int dl = 0, dh = 0;
        dl = dot4(q.x & 0x0f0f0f0f, lo.x, dl); dh = dot4((q.x >> 4) & 0x0f0f0f0f, hi.x, dh);
        dl = dot4(q.y & 0x0f0f0f0f, lo.y, dl); dh = dot4((q.y >> 4) & 0x0f0f0f0f, hi.y, dh);
        dl = dot4(q.z & 0x0f0f0f0f, lo.z, dl); dh = dot4((q.z >> 4) & 0x0f0f0f0f, hi.z, dh);
        dl = dot4(q.w & 0x0f0f0f0f, lo.w, dl); dh = dot4((q.w >> 4) & 0x0f0f0f0f, hi.w, dh);
        isum = sc_lo * dl + sc_hi * dh;
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 4 * c + h;
        msum = m_lo * (int)bs[0] + m_hi * (int)bs[2];
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum - (dmin * yd) * (float)msum;
    }
};

// ---- Q5_K : [d][dmin][scales 12][qh 32][qs 128]  (176 B, 16-B aligned) ----------------------
template <> struct Item<T_Q5_K> {
    static constexpr int EPP = 2048;
    uint4 q, qh;
    int sc_lo, sc_hi, m_lo, m_hi;
    float d, dmin;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane, uint4 &hdr) {
        sb = pass * 8 + (lane >> 3);
        valid = sb < (K >> 8);
        if (valid) {
            const uint8_t *b = row + (size_t)sb * 176;
            hdr = ld16(b);
            qh = ld16(b + 16 + (lane & 1) * 16);
            q = ld16(b + 48 + (lane & 7) * 16);
        }
    }
    __device__ __forceinline__ void prep(const uint4 &hdr, int lane) {
        if (!valid) return;
        const int v = lane & 7, c = v >> 1;
        d = h2f((uint16_t)(hdr.x & 0xffff));
        dmin = h2f((uint16_t)(hdr.x >> 16));
        const int sh = (c & 1) * 16;
        const uint32_t a16 = (hdr.y >> sh) & 0xffff, b16 = (hdr.z >> sh) & 0xffff, c16 = (hdr.w >> sh) & 0xffff;
        uint32_t sc, mn;
        if (c < 2) {
            sc = a16 & 0x3f3f;
            mn = b16 & 0x3f3f;
        } else {
            sc = (c16 & 0x0f0f) | ((a16 >> 2) & 0x3030);
            mn = ((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030);
        }
        sc_lo = sc & 0xff; sc_hi = sc >> 8;
        m_lo = mn & 0xff;  m_hi = mn >> 8;
        // fold the 5th bits into the nibble planes once: q.x.. stay packed, qh becomes per-plane masks
        const int s0 = 2 * c, s1 = 2 * c + 1;
        uint4 l5, h5;
        l5.x = ((qh.x >> s0) & 0x01010101u) << 4; h5.x = ((qh.x >> s1) & 0x01010101u) << 4;
        l5.y = ((qh.y >> s0) & 0x01010101u) << 4; h5.y = ((qh.y >> s1) & 0x01010101u) << 4;
        l5.z = ((qh.z >> s0) & 0x01010101u) << 4; h5.z = ((qh.z >> s1) & 0x01010101u) << 4;
        l5.w = ((qh.w >> s0) & 0x01010101u) << 4; h5.w = ((qh.w >> s1) & 0x01010101u) << 4;
        lo5 = l5; hi5 = h5;
    }
    uint4 lo5, hi5;
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, c = v >> 1, h = v & 1;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 64 * c + 16 * h;
        const uint4 lo = ld16(a), hi = ld16(a + 32);
        int dl = 0, dh = 0;
        dl = dot4((q.x & 0x0f0f0f0f) | lo5.x, lo.x, dl); dh = dot4(((q.x >> 4) & 0x0f0f0f0f) | hi5.x, hi.x, dh);
        dl = dot4((q.y & 0x0f0f0f0f) | lo5.y, lo.y, dl); dh = dot4(((q.y >> 4) & 0x0f0f0f0f) | hi5.y, hi.y, dh);
        dl = dot4((q.z & 0x0f0f0f0f) | lo5.z, lo.z, dl); dh = dot4(((q.z >> 4) & 0x0f0f0f0f) | hi5.z, hi.z, dh);
        dl = dot4((q.w & 0x0f0f0f0f) | lo5.w, lo.w, dl); dh = dot4(((q.w >> 4) & 0x0f0f0f0f) | hi5.w, hi.w, dh);
        isum = sc_lo * dl + sc_hi * dh;
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 4 * c + h;
        msum = m_lo * (int)bs[0] + m_hi * (int)bs[2];
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum - (dmin * yd) * (float)msum;
    }
};

// ---- Q6_K : device row planes [ql nb*128][qh nb*64][scales nb*16][d nb*2] ---------------------
template <> struct Item<T_Q6_K> {
    static constexpr int EPP = 2048;
    uint4 ql, qh;
    int sc_lo, sc_hi;
    float d;
    int sb;
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane, uint4 &hdr) {
        const int nb = K >> 8;
        sb = pass * 8 + (lane >> 3);
        valid = sb < nb;
        if (valid) {
            const int v = lane & 7, n = v >> 2, w = v & 3;
            ql = ld16(row + (size_t)sb * 128 + v * 16);
            qh = ld16(row + (size_t)nb * 128 + (size_t)sb * 64 + n * 32 + (w & 1) * 16);
            const int8_t *sc = reinterpret_cast<const int8_t *>(row + (size_t)nb * 192 + (size_t)sb * 16 + 8 * n + w);
            hdr.x = (uint32_t)(int)sc[0];
            hdr.y = (uint32_t)(int)sc[4];
            hdr.z = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2);
        }
    }
    __device__ __forceinline__ void prep(const uint4 &hdr, int lane) {
        if (!valid) return;
        sc_lo = (int)hdr.x; sc_hi = (int)hdr.y;
        d = h2f((uint16_t)hdr.z);
        const int w = lane & 3;
        const int s0 = 2 * (w >> 1), s1 = s0 + 4;
        uint4 a, b;
        a.x = (ql.x & 0x0f0f0f0f) | (((qh.x >> s0) & 0x03030303u) << 4); b.x = ((ql.x >> 4) & 0x0f0f0f0f) | (((qh.x >> s1) & 0x03030303u) << 4);
        a.y = (ql.y & 0x0f0f0f0f) | (((qh.y >> s0) & 0x03030303u) << 4); b.y = ((ql.y >> 4) & 0x0f0f0f0f) | (((qh.y >> s1) & 0x03030303u) << 4);
        a.z = (ql.z & 0x0f0f0f0f) | (((qh.z >> s0) & 0x03030303u) << 4); b.z = ((ql.z >> 4) & 0x0f0f0f0f) | (((qh.z >> s1) & 0x03030303u) << 4);
        a.w = (ql.w & 0x0f0f0f0f) | (((qh.w >> s0) & 0x03030303u) << 4); b.w = ((ql.w >> 4) & 0x0f0f0f0f) | (((qh.w >> s1) & 0x03030303u) << 4);
        ql = a; qh = b;   // now: ql = low-plane 6-bit codes, qh = high-plane 6-bit codes (0..63)
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        const int v = lane & 7, n = v >> 2, w = v & 3;
        const int8_t *a = A.qs + (size_t)t * A.K + sb * 256 + 128 * n + 16 * w;
        const uint4 lo = ld16(a), hi = ld16(a + 64);
        int dl = 0, dh = 0;
        dl = dot4(ql.x, lo.x, dl); dh = dot4(qh.x, hi.x, dh);
        dl = dot4(ql.y, lo.y, dl); dh = dot4(qh.y, hi.y, dh);
        dl = dot4(ql.z, lo.z, dl); dh = dot4(qh.z, hi.z, dh);
        dl = dot4(ql.w, lo.w, dl); dh = dot4(qh.w, hi.w, dh);
        const int16_t *bs = A.bs + (size_t)t * (A.K >> 4) + sb * 16 + 8 * n + w;
        isum = sc_lo * (dl - 32 * (int)bs[0]) + sc_hi * (dh - 32 * (int)bs[4]);
        msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        if (!valid) return 0.0f;
        int isum, msum;
        ints(A, t, lane, isum, msum);
        const float yd = A.d[(size_t)t * (A.K >> 8) + sb];
        return (d * yd) * (float)isum;
    }
};

// ---- Q8_0 : device row planes [qs K][d K/32 f16] -------------------------------------------------
template <> struct Item<T_Q8_0> {
    static constexpr int EPP = 1024;
    uint4 q;
    float d;
    int e;       // element offset of this lane's 16 codes
    bool valid;
    __device__ __forceinline__ void load(const uint8_t *row, int K, int pass, int lane, uint4 &hdr) {
        e = pass * 1024 + lane * 16;
        valid = e < K;
        if (valid) {
            q = ld16(row + e);
            hdr.x = *reinterpret_cast<const uint16_t *>(row + K + (e >> 5) * 2);
        }
    }
    __device__ __forceinline__ void prep(const uint4 &hdr, int lane) {
        if (valid) d = h2f((uint16_t)hdr.x);
    }
    __device__ __forceinline__ void ints(const ActLds &A, int t, int lane, int &isum, int &msum) const {
        int s = 0;
        if (valid) {
            const uint4 a = ld16(A.qs0 + (size_t)t * A.K + e);
            s = dot4(q.x, a.x, s); s = dot4(q.y, a.y, s); s = dot4(q.z, a.z, s); s = dot4(q.w, a.w, s);
        }
        isum = s + __shfl_xor(s, 1, 64);   // both halves of the 32-element block
        msum = 0;
    }
    __device__ __forceinline__ float dot(const ActLds &A, int t, int lane) const {
        int isum, msum;
        ints(A, t, lane, isum, msum);     // shuffle executed by all lanes
        if (!valid || (lane & 1)) return 0.0f;
        const float da = h2f(A.d0[(size_t)t * (A.K >> 5) + (e >> 5)]);
        return (float)isum * (d * da);
    }
};

// ------------------------------------------------------------------------------------------
template <int TYPE, int NT>
__device__ __forceinline__ void row_pair_dot(const uint8_t *r0, const uint8_t *r1, bool has1, int K,
                                             const ActLds &A, int lane, float (&acc)[2][NT]) {
    using It = Item<TYPE>;
    const int npass = (K + It::EPP - 1) / It::EPP;
    for (int p = 0; p < npass; p += 2) {
        It a0, a1, b0, b1;
        uint4 ha0, ha1, hb0, hb1;
        a0.load(r0, K, p, lane, ha0);
        if (has1) b0.load(r1, K, p, lane, hb0); else b0.valid = false;
        if (p + 1 < npass) {
            a1.load(r0, K, p + 1, lane, ha1);
            if (has1) b1.load(r1, K, p + 1, lane, hb1); else b1.valid = false;
        } else {
            a1.valid = false; b1.valid = false;
        }
        a0.prep(ha0, lane); b0.prep(hb0, lane); a1.prep(ha1, lane); b1.prep(hb1, lane);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            acc[0][t] += a0.dot(A, t, lane);
            acc[1][t] += b0.dot(A, t, lane);
            acc[0][t] += a1.dot(A, t, lane);
            acc[1][t] += b1.dot(A, t, lane);
        }
    }
}

// All rows of one segment handled by the waves of the blocks assigned to it (block-uniform TYPE).
template <int TYPE, int NT>
__device__ __forceinline__ void run_segment(const MMVQArgs &a, const MMVQSeg &sg, const ActLds &A, int gw, int nw) {
    const int lane = threadIdx.x & 63;
    const int K = a.K;
    // MoE: the expert is picked on the device (no host sync): weight base = W + expert_sel[0] * expert_stride
    const size_t eoff = sg.expert_sel ? (size_t)sg.expert_sel[0] * sg.expert_stride : 0;
    if (a.epi == EPI_SWIGLU) {
        const MMVQSeg &u = a.seg[1];
        const size_t uoff = u.expert_sel ? (size_t)u.expert_sel[0] * u.expert_stride : 0;
        for (int p = gw; p < sg.n_rows; p += nw) {
            float acc[2][NT];
#pragma unroll
            for (int t = 0; t < NT; t++) { acc[0][t] = 0.0f; acc[1][t] = 0.0f; }
            row_pair_dot<TYPE, NT>(sg.W + eoff + (size_t)p * sg.row_bytes, u.W + uoff + (size_t)p * u.row_bytes, true, K, A, lane, acc);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const float gv = wave_sum(acc[0][t]), uv = wave_sum(acc[1][t]);
                if (lane == 0) sg.out[(size_t)t * sg.ld_out + p] = (gv / (1.0f + expf(-gv))) * uv;
            }
        }
        return;
    }
    const int npairs = (sg.n_rows + 1) >> 1;
    for (int p = gw; p < npairs; p += nw) {
        float acc[2][NT];
#pragma unroll
        for (int t = 0; t < NT; t++) { acc[0][t] = 0.0f; acc[1][t] = 0.0f; }
        const int row0 = 2 * p;
        const bool has1 = row0 + 1 < sg.n_rows;
        const uint8_t *r0 = sg.W + eoff + (size_t)row0 * sg.row_bytes;
        row_pair_dot<TYPE, NT>(r0, r0 + sg.row_bytes, has1, K, A, lane, acc);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float v0 = wave_sum(acc[0][t]), v1 = wave_sum(acc[1][t]);
            if (lane == 0) {
                const size_t o = (size_t)t * sg.ld_out + row0;
                if (a.epi == EPI_ADD) {
                    sg.out[o] = sg.resid[o] + v0;
                    if (has1) sg.out[o + 1] = sg.resid[o + 1] + v1;
                } else {
                    sg.out[o] = v0;
                    if (has1) sg.out[o + 1] = v1;
                }
            }
        }
    }
}

// stage the quantised activations of NT tokens into LDS; returns the view
template <int NT>
__device__ __forceinline__ ActLds stage_act(const MMVQArgs &a, uint8_t *smem) {
    ActLds A;
    const int K = a.K, tid = threadIdx.x, nthr = blockDim.x;
    A.K = K;
    uint8_t *p = smem;
    A.qs = nullptr; A.d = nullptr; A.bs = nullptr; A.qs0 = nullptr; A.d0 = nullptr;
    if (a.need_q8k) {
        int8_t *qs = reinterpret_cast<int8_t *>(p);           p += (size_t)NT * K;
        float *d = reinterpret_cast<float *>(p);              p += (((size_t)NT * (K >> 8) * 4) + 15) & ~15;
        int16_t *bs = reinterpret_cast<int16_t *>(p);         p += (((size_t)NT * (K >> 4) * 2) + 15) & ~15;
        const uint4 *src = reinterpret_cast<const uint4 *>(a.aq);
        for (int i = tid; i < NT * K / 16; i += nthr) reinterpret_cast<uint4 *>(qs)[i] = src[i];
        for (int i = tid; i < NT * (K >> 8); i += nthr) d[i] = a.ad[i];
        const uint32_t *bsrc = reinterpret_cast<const uint32_t *>(a.abs);
        for (int i = tid; i < NT * (K >> 5); i += nthr) reinterpret_cast<uint32_t *>(bs)[i] = bsrc[i];
        A.qs = qs; A.d = d; A.bs = bs;
    }
    if (a.need_q80) {
        int8_t *qs0 = reinterpret_cast<int8_t *>(p);          p += (size_t)NT * K;
        uint16_t *d0 = reinterpret_cast<uint16_t *>(p);       p += (((size_t)NT * (K >> 5) * 2) + 15) & ~15;
        const uint4 *src = reinterpret_cast<const uint4 *>(a.aq0);
        for (int i = tid; i < NT * K / 16; i += nthr) reinterpret_cast<uint4 *>(qs0)[i] = src[i];
        for (int i = tid; i < NT * (K >> 5); i += nthr) d0[i] = a.ad0[i];
        A.qs0 = qs0; A.d0 = d0;
    }
    __syncthreads();
    return A;
}

// Blocks [seg_block0[s], seg_block0[s+1]) work on segment s, so the weight TYPE is block-uniform and
// each type's code path keeps its own (small) register footprint.
template <int NT>
__global__ __launch_bounds__(256) void mmvq_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ActLds A = stage_act<NT>(a, smem);
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    const int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = ((int)blockIdx.x - a.seg_block0[s]) * 4 + wave;
    const int nw = nblk * 4;
    switch (a.seg[s].type) {
        case T_Q4_K: run_segment<T_Q4_K, NT>(a, a.seg[s], A, gw, nw); break;
        case T_Q5_K: run_segment<T_Q5_K, NT>(a, a.seg[s], A, gw, nw); break;
        case T_Q6_K: run_segment<T_Q6_K, NT>(a, a.seg[s], A, gw, nw); break;
        case T_Q8_0: run_segment<T_Q8_0, NT>(a, a.seg[s], A, gw, nw); break;
        default: break;
    }
}

size_t mmvq_lds_bytes(const MMVQArgs &a, int NT) {
    size_t b = 0;
    const size_t K = (size_t)a.K;
    if (a.need_q8k) b += NT * K + ((NT * (K >> 8) * 4 + 15) & ~(size_t)15) + ((NT * (K >> 4) * 2 + 15) & ~(size_t)15);
    if (a.need_q80) b += NT * K + ((NT * (K >> 5) * 2 + 15) & ~(size_t)15);
    return b;
}

static int g_num_cu = 256;
void set_num_cu(int n) { if (n > 0) g_num_cu = n; }
int num_cu() { return g_num_cu; }

// host launcher: a.T tokens (1, 2 or 4 per launch; larger T is chunked by the caller).
// EPI_SWIGLU: seg[0] = ffn_gate, seg[1] = ffn_up (same type), out = silu(gate) * up into seg[0].out.
hipError_t launch_mmvq(MMVQArgs a, hipStream_t st) {
    a.need_q8k = 0; a.need_q80 = 0;
    for (int s = 0; s < a.n_seg; s++) {
        if (a.seg[s].type == T_Q8_0) a.need_q80 = 1; else a.need_q8k = 1;
    }
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return hipErrorInvalidValue;
    const int n_work_seg = a.epi == EPI_SWIGLU ? 1 : a.n_seg;
    const int max_blocks = g_num_cu * 8;
    size_t bytes[3] = {0, 0, 0}, total = 0;
    int want[3] = {0, 0, 0};
    for (int s = 0; s < n_work_seg; s++) {
        bytes[s] = (size_t)a.seg[s].n_rows * a.seg[s].row_bytes;
        total += bytes[s];
        const int units = a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
        want[s] = (units + 3) / 4;               // one unit (row pair) per wave at most
    }
    int sum_want = 0;
    for (int s = 0; s < n_work_seg; s++) sum_want += want[s];
    a.seg_block0[0] = 0;
    for (int s = 0; s < n_work_seg; s++) {
        int nb = want[s];
        if (sum_want > max_blocks) {            // share the block budget by bytes
            nb = (int)((double)max_blocks * (double)bytes[s] / (double)total);
            if (nb < 1) nb = 1;
            if (nb > want[s]) nb = want[s];
        }
        a.seg_block0[s + 1] = a.seg_block0[s] + nb;
    }
    for (int s = n_work_seg; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n_work_seg];
    const int blocks = a.seg_block0[n_work_seg];
    if (a.epi == EPI_SWIGLU) a.n_seg = 1;       // block->segment lookup sees only the gate segment; seg[1] read directly
    const size_t lds = mmvq_lds_bytes(a, a.T);
    switch (a.T) {
        case 1: hipLaunchKernelGGL(mmvq_kernel<1>, dim3(blocks), dim3(256), lds, st, a); break;
        case 2: hipLaunchKernelGGL(mmvq_kernel<2>, dim3(blocks), dim3(256), lds, st, a); break;
        case 4: hipLaunchKernelGGL(mmvq_kernel<4>, dim3(blocks), dim3(256), lds, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Debug / parity kernel: integer partial sums per (token, row, block), same Item code as above.
template <int TYPE>
__global__ __launch_bounds__(256) void mmvq_ints_kernel(const MMVQArgs a, int32_t *isum_out, int32_t *msum_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ActLds A = stage_act<1>(a, smem);
    using It = Item<TYPE>;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nw = gridDim.x * (blockDim.x >> 6);
    const int K = a.K;
    const MMVQSeg &sg = a.seg[0];
    const int nblk = TYPE == T_Q8_0 ? (K >> 5) : (K >> 8);
    const int npass = (K + It::EPP - 1) / It::EPP;
    for (int r = gw; r < sg.n_rows; r += nw) {
        const uint8_t *row = sg.W + (size_t)r * sg.row_bytes;
        for (int p = 0; p < npass; p++) {
            It it;
            uint4 hdr;
            it.load(row, K, p, lane, hdr);
            it.prep(hdr, lane);
            int is = 0, ms = 0;
            if constexpr (TYPE == T_Q8_0) {
                it.ints(A, 0, lane, is, ms);
                if (it.valid && !(lane & 1)) { isum_out[(size_t)r * nblk + (it.e >> 5)] = is; msum_out[(size_t)r * nblk + (it.e >> 5)] = 0; }
            } else {
                if (it.valid) it.ints(A, 0, lane, is, ms);
                is += __shfl_xor(is, 1, 64); ms += __shfl_xor(ms, 1, 64);
                is += __shfl_xor(is, 2, 64); ms += __shfl_xor(ms, 2, 64);
                is += __shfl_xor(is, 4, 64); ms += __shfl_xor(ms, 4, 64);
                const int sb = p * 8 + (lane >> 3);
                if (sb < nblk && (lane & 7) == 0) { isum_out[(size_t)r * nblk + sb] = is; msum_out[(size_t)r * nblk + sb] = ms; }
            }
        }
    }
}

hipError_t launch_mmvq_ints(MMVQArgs a, int32_t *isum, int32_t *msum, hipStream_t st) {
    a.need_q8k = a.seg[0].type != T_Q8_0;
    a.need_q80 = a.seg[0].type == T_Q8_0;
    const size_t lds = mmvq_lds_bytes(a, 1);
    const int blocks = 64;
    switch (a.seg[0].type) {
        case T_Q4_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q4_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q5_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q5_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q6_K: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q6_K>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        case T_Q8_0: hipLaunchKernelGGL(mmvq_ints_kernel<T_Q8_0>, dim3(blocks), dim3(256), lds, st, a, isum, msum); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mi355
