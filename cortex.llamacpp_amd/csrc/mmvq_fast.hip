// mmvq_fast.hip — single-token quantised mat-vec for the decode step, K % 2048 == 0 (every Llama-class shape).
//
// Same arithmetic as mmvq.hip (integer partial sums identical to ggml_vec_dot_q{4,5,6}_K_q8_K),
// restructured after measurement on MI355X (tools/bench_stream.hip, tools/bench_mmvq.hip):
//   * the 16-byte-per-lane load pattern alone streams at 5.3-5.7 TB/s; what slowed the first kernels down was the
//     serial  load -> wait -> decode/dot  chain inside each wave.  Here every wave is persistent and software-
//     pipelined: the loads of unit n+1 (2 rows x 2 passes = 8 x 16 B per lane) are in flight while unit n is decoded,
//     with two statically named register sets;
//   * the decode is branch-free (selects instead of divergent branches), uses 24-bit multiplies, and the Q8_K
//     activation slice of a pass is read once and applied to both rows of the pair;
//   * for K <= 4096 the lane's activation slices (2 passes) live in registers for the whole kernel, so the inner loop
//     touches neither LDS nor L2 for activations.
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldw(const void *p) {   // weight stream: non-temporal
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 lds16(const void *p) { return *reinterpret_cast<const uint4 *>(p); }
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// activation slice one lane needs for one pass (8 super-blocks per wave, lane -> (sb, 16-code piece))
struct ActSlice {
    uint4 lo, hi;      // 16 + 16 int8 codes
    float yd;          // Q8_K block scale
    int bs_lo, bs_hi;  // sums of the two 16-code groups
};

struct LaneRole {      // constants of this lane, computed once
    int sbl;           // super-block within the pass (0..7)
    int v, c, h;       // piece (0..7), chunk (0..3), half (0..1)
    int sh;            // bit shift selecting this chunk's 16-bit scale pair
    bool hi_scales;    // c >= 2: 6-bit scales split across bytes
    int n, w;          // Q6_K: half (0..1) and 16-code column (0..3)
};

template <int TYPE> struct Raw;

// ---------------------------------------------------------------- Q4_K
template <> struct Raw<T_Q4_K> {
    uint4 hdr, q;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const uint8_t *b = row + (size_t)sb * 144;
        hdr = ldw(b);
        q = ldw(b + 16 + L.v * 16);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
        const uint32_t a16 = (hdr.y >> L.sh) & 0xffff, b16 = (hdr.z >> L.sh) & 0xffff, c16 = (hdr.w >> L.sh) & 0xffff;
        const uint32_t sc = L.hi_scales ? ((c16 & 0x0f0f) | ((a16 >> 2) & 0x3030)) : (a16 & 0x3f3f);
        const uint32_t mn = L.hi_scales ? (((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030)) : (b16 & 0x3f3f);
        int dl = 0, dh = 0;
        dl = dot4(q.x & 0x0f0f0f0f, A.lo.x, dl); dh = dot4((q.x >> 4) & 0x0f0f0f0f, A.hi.x, dh);
        dl = dot4(q.y & 0x0f0f0f0f, A.lo.y, dl); dh = dot4((q.y >> 4) & 0x0f0f0f0f, A.hi.y, dh);
        dl = dot4(q.z & 0x0f0f0f0f, A.lo.z, dl); dh = dot4((q.z >> 4) & 0x0f0f0f0f, A.hi.z, dh);
        dl = dot4(q.w & 0x0f0f0f0f, A.lo.w, dl); dh = dot4((q.w >> 4) & 0x0f0f0f0f, A.hi.w, dh);
        const int isum = mul24((int)(sc & 0xff), dl) + mul24((int)(sc >> 8), dh);
        const int msum = mul24((int)(mn & 0xff), A.bs_lo) + mul24((int)(mn >> 8), A.bs_hi);
        return (d * A.yd) * (float)isum - (dmin * A.yd) * (float)msum;
    }
};

// ---------------------------------------------------------------- Q5_K
template <> struct Raw<T_Q5_K> {
    uint4 hdr, qh, q;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        const uint8_t *b = row + (size_t)sb * 176;
        hdr = ldw(b);
        qh = ldw(b + 16 + L.h * 16);
        q = ldw(b + 48 + L.v * 16);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)(hdr.x & 0xffff)), dmin = h2f((uint16_t)(hdr.x >> 16));
        const uint32_t a16 = (hdr.y >> L.sh) & 0xffff, b16 = (hdr.z >> L.sh) & 0xffff, c16 = (hdr.w >> L.sh) & 0xffff;
        const uint32_t sc = L.hi_scales ? ((c16 & 0x0f0f) | ((a16 >> 2) & 0x3030)) : (a16 & 0x3f3f);
        const uint32_t mn = L.hi_scales ? (((c16 >> 4) & 0x0f0f) | ((b16 >> 2) & 0x3030)) : (b16 & 0x3f3f);
        const int s0 = 2 * L.c, s1 = s0 + 1;
        int dl = 0, dh = 0;
#define Q5L(w, hw) (((w) & 0x0f0f0f0f) | ((((hw) >> s0) & 0x01010101u) << 4))
#define Q5H(w, hw) ((((w) >> 4) & 0x0f0f0f0f) | ((((hw) >> s1) & 0x01010101u) << 4))
        dl = dot4(Q5L(q.x, qh.x), A.lo.x, dl); dh = dot4(Q5H(q.x, qh.x), A.hi.x, dh);
        dl = dot4(Q5L(q.y, qh.y), A.lo.y, dl); dh = dot4(Q5H(q.y, qh.y), A.hi.y, dh);
        dl = dot4(Q5L(q.z, qh.z), A.lo.z, dl); dh = dot4(Q5H(q.z, qh.z), A.hi.z, dh);
        dl = dot4(Q5L(q.w, qh.w), A.lo.w, dl); dh = dot4(Q5H(q.w, qh.w), A.hi.w, dh);
#undef Q5L
#undef Q5H
        const int isum = mul24((int)(sc & 0xff), dl) + mul24((int)(sc >> 8), dh);
        const int msum = mul24((int)(mn & 0xff), A.bs_lo) + mul24((int)(mn >> 8), A.bs_hi);
        return (d * A.yd) * (float)isum - (dmin * A.yd) * (float)msum;
    }
};

// ---------------------------------------------------------------- Q6_K (device row planes: ql | qh | scales | d)
template <> struct Raw<T_Q6_K> {
    uint4 ql, qh;
    int sc_lo, sc_hi;
    uint32_t dh16;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, const LaneRole &L) {
        ql = ldw(row + (size_t)sb * 128 + L.v * 16);
        qh = ldw(row + (size_t)nb * 128 + (size_t)sb * 64 + L.n * 32 + (L.w & 1) * 16);
        const int8_t *sc = reinterpret_cast<const int8_t *>(row + (size_t)nb * 192 + (size_t)sb * 16 + 8 * L.n + L.w);
        sc_lo = sc[0];
        sc_hi = sc[4];
        dh16 = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2);
    }
    __device__ __forceinline__ float dot(const ActSlice &A, const LaneRole &L) const {
        const float d = h2f((uint16_t)dh16);
        const int s0 = 2 * (L.w >> 1), s1 = s0 + 4;
        int dl = 0, dh = 0;
#define Q6L(l, hh) (((l) & 0x0f0f0f0f) | ((((hh) >> s0) & 0x03030303u) << 4))
#define Q6H(l, hh) ((((l) >> 4) & 0x0f0f0f0f) | ((((hh) >> s1) & 0x03030303u) << 4))
        dl = dot4(Q6L(ql.x, qh.x), A.lo.x, dl); dh = dot4(Q6H(ql.x, qh.x), A.hi.x, dh);
        dl = dot4(Q6L(ql.y, qh.y), A.lo.y, dl); dh = dot4(Q6H(ql.y, qh.y), A.hi.y, dh);
        dl = dot4(Q6L(ql.z, qh.z), A.lo.z, dl); dh = dot4(Q6H(ql.z, qh.z), A.hi.z, dh);
        dl = dot4(Q6L(ql.w, qh.w), A.lo.w, dl); dh = dot4(Q6H(ql.w, qh.w), A.hi.w, dh);
#undef Q6L
#undef Q6H
        const int isum = mul24(sc_lo, dl - 32 * A.bs_lo) + mul24(sc_hi, dh - 32 * A.bs_hi);
        return (d * A.yd) * (float)isum;
    }
};

template <int TYPE> __device__ __forceinline__ LaneRole make_role(int lane) {
    LaneRole L;
    L.sbl = lane >> 3; L.v = lane & 7; L.c = L.v >> 1; L.h = L.v & 1;
    L.sh = (L.c & 1) * 16; L.hi_scales = L.c >= 2;
    L.n = L.v >> 2; L.w = L.v & 3;
    return L;
}

// LDS view of the staged Q8_K activation of the token
struct ActL { const int8_t *qs; const float *d; const int16_t *bs; };

template <int TYPE>
__device__ __forceinline__ ActSlice read_slice(const ActL &A, int sb, const LaneRole &L) {
    ActSlice s;
    if (TYPE == T_Q6_K) {
        const int8_t *a = A.qs + sb * 256 + 128 * L.n + 16 * L.w;
        s.lo = lds16(a); s.hi = lds16(a + 64);
        const int16_t *b = A.bs + sb * 16 + 8 * L.n + L.w;
        s.bs_lo = b[0]; s.bs_hi = b[4];
    } else {
        const int8_t *a = A.qs + sb * 256 + 64 * L.c + 16 * L.h;
        s.lo = lds16(a); s.hi = lds16(a + 32);
        const int16_t *b = A.bs + sb * 16 + 4 * L.c + L.h;
        s.bs_lo = b[0]; s.bs_hi = b[2];
    }
    s.yd = A.d[sb];
    return s;
}

// stage the activation (Q8_K planes) into LDS: copy (fuse_mode 0) or RMSNorm*w + quantise / quantise (1 / 2)
__device__ __forceinline__ ActL stage_q8k(const MMVQArgs &a, uint8_t *smem) {
    const int K = a.K, tid = threadIdx.x;
    int8_t *qs = reinterpret_cast<int8_t *>(smem);
    float *d = reinterpret_cast<float *>(smem + K);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + K + (((K >> 8) * 4 + 15) & ~15));
    if (a.fuse_mode == 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.aq);
        uint4 t[4];
        for (int i0 = 0; i0 < K / 16; i0 += 1024) {      // all loads of a sweep issued before any LDS write
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = i0 + j * 256 + tid; if (i < K / 16) t[j] = src[i]; }
#pragma unroll
            for (int j = 0; j < 4; j++) { const int i = i0 + j * 256 + tid; if (i < K / 16) reinterpret_cast<uint4 *>(qs)[i] = t[j]; }
        }
        for (int i = tid; i < (K >> 8); i += 256) d[i] = a.ad[i];
        const uint32_t *bsrc = reinterpret_cast<const uint32_t *>(a.abs);
        for (int i = tid; i < (K >> 5); i += 256) reinterpret_cast<uint32_t *>(bs)[i] = bsrc[i];
    } else {
        double *red = reinterpret_cast<double *>(smem + a.red_off);
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int MAXJ = 8;                  // K <= 8192
        float4 xv[MAXJ], wv[MAXJ];
        const int nj = K >> 10;
#pragma unroll
        for (int j = 0; j < MAXJ; j++)
            if (j < nj) {
                xv[j] = *reinterpret_cast<const float4 *>(a.nx + j * 1024 + tid * 4);
                if (a.fuse_mode == 1) wv[j] = *reinterpret_cast<const float4 *>(a.nw + j * 1024 + tid * 4);
            }
        float scale = 1.0f;
        if (a.fuse_mode == 1) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < MAXJ; j++)
                if (j < nj) {
                    const float4 v = xv[j];
                    s += (double)(v.x * v.x); s += (double)(v.y * v.y); s += (double)(v.z * v.z); s += (double)(v.w * v.w);
                }
            s = wave_sum(s);
            if (lane == 0) red[wave] = s;
            __syncthreads();
            const double tot = red[0] + red[1] + red[2] + red[3];
            const float mean = (float)(tot / (double)K);
            scale = 1.0f / sqrtf(mean + a.neps);
        }
#pragma unroll
        for (int j = 0; j < MAXJ; j++) {
            if (j >= nj) continue;
            const int b = wave + 4 * j;
            const int e0 = b * 256 + lane * 4;
            float4 v = xv[j];
            if (a.fuse_mode == 1) {
                const float4 ww = wv[j];
                v.x = (v.x * scale) * ww.x; v.y = (v.y * scale) * ww.y; v.z = (v.z * scale) * ww.z; v.w = (v.w * scale) * ww.w;
            }
            const float vv[4] = {v.x, v.y, v.z, v.w};
            uint32_t packed; int bsum; float dq;
            wave_quant_q8k(vv, lane, packed, bsum, dq);
            *reinterpret_cast<uint32_t *>(qs + e0) = packed;
            if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum;
            if (lane == 0) d[b] = dq;
        }
    }
    __syncthreads();
    ActL A{qs, d, bs};
    return A;
}

// One segment, persistent waves.  unit = (row pair, 2 passes); rows longer than 2 passes are walked chunk by chunk.
template <int TYPE, bool ACT_REGS>
__device__ __forceinline__ void run_fast(const MMVQArgs &a, const MMVQSeg &sg, uint8_t *smem, int gw, int nw) {
    using R = Raw<TYPE>;
    const int lane = threadIdx.x & 63;
    const LaneRole L = make_role<TYPE>(lane);
    const int K = a.K, nb = K >> 8;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const uint8_t *W0 = sg.W + (sg.expert_sel ? (size_t)sg.expert_sel[0] * sg.expert_stride : 0);
    const MMVQSeg &ug = a.seg[1];
    const uint8_t *W1 = swiglu ? ug.W + (ug.expert_sel ? (size_t)ug.expert_sel[0] * ug.expert_stride : 0) : nullptr;
    const size_t rb0 = sg.row_bytes, rb1 = swiglu ? ug.row_bytes : sg.row_bytes;
    const int npass = nb >> 3;                    // K % 2048 == 0
    const int nchunk = (npass + 1) >> 1;
    const int npairs = swiglu ? sg.n_rows : (sg.n_rows + 1) >> 1;

    auto rows_of = [&](int pair, const uint8_t *&ra, const uint8_t *&rbp) {
        if (swiglu) { ra = W0 + (size_t)pair * rb0; rbp = W1 + (size_t)pair * rb1; }
        else {
            ra = W0 + (size_t)(2 * pair) * rb0;
            rbp = (2 * pair + 1 < sg.n_rows) ? ra + rb0 : ra;   // odd tail: row b re-reads row a, result discarded
        }
    };
    // unit u of this wave: pair = gw + (u / nchunk) * nw, chunk = u % nchunk
    const int my_pairs = gw < npairs ? (npairs - gw + nw - 1) / nw : 0;
    const int n_units = my_pairs * nchunk;

    R A0, A1, A2, A3, B0, B1, B2, B3;             // two statically named register sets
    auto load_unit = [&](int u, R &r0, R &r1, R &r2, R &r3) {
        const int pi = u / nchunk, ch = u - pi * nchunk;
        const uint8_t *ra, *rbp;
        rows_of(gw + pi * nw, ra, rbp);
        const int p0 = 2 * ch;
        const int sb0 = p0 * 8 + L.sbl;
        r0.load(ra, nb, sb0, L);
        r1.load(rbp, nb, sb0, L);
        if (p0 + 1 < npass) {                       // wave-uniform
            r2.load(ra, nb, sb0 + 8, L);
            r3.load(rbp, nb, sb0 + 8, L);
        }
    };
    if (n_units > 0) load_unit(0, A0, A1, A2, A3);

    const ActL AL = stage_q8k(a, smem);             // workgroup barrier inside: reached by every wave
    ActSlice S0, S1;                                 // K <= 4096: this lane's slices of pass 0 / pass 1, kept in registers
    if (ACT_REGS) {
        S0 = read_slice<TYPE>(AL, L.sbl, L);
        if (npass > 1) S1 = read_slice<TYPE>(AL, 8 + L.sbl, L);
    }

    float acc0 = 0.0f, acc1 = 0.0f;
    auto compute_unit = [&](int u, const R &r0, const R &r1, const R &r2, const R &r3) {
        const int pi = u / nchunk, ch = u - pi * nchunk;
        const int p0 = 2 * ch;
        {
            const ActSlice s = ACT_REGS ? S0 : read_slice<TYPE>(AL, p0 * 8 + L.sbl, L);
            acc0 += r0.dot(s, L);
            acc1 += r1.dot(s, L);
        }
        if (p0 + 1 < npass) {
            const ActSlice s = ACT_REGS ? S1 : read_slice<TYPE>(AL, (p0 + 1) * 8 + L.sbl, L);
            acc0 += r2.dot(s, L);
            acc1 += r3.dot(s, L);
        }
        if (ch == nchunk - 1) {                      // pair finished
            const float v0 = wave_sum(acc0), v1 = wave_sum(acc1);
            acc0 = 0.0f; acc1 = 0.0f;
            if (lane == 0) {
                const int pair = gw + pi * nw;
                if (swiglu) {
                    sg.out[pair] = (v0 / (1.0f + expf(-v0))) * v1;
                } else {
                    const int row0 = 2 * pair;
                    const bool has1 = row0 + 1 < sg.n_rows;
                    if (a.epi == EPI_ADD) {
                        sg.out[row0] = sg.resid[row0] + v0;
                        if (has1) sg.out[row0 + 1] = sg.resid[row0 + 1] + v1;
                    } else {
                        sg.out[row0] = v0;
                        if (has1) sg.out[row0 + 1] = v1;
                    }
                }
            }
        }
    };
    for (int u = 0; u < n_units; u += 2) {
        if (u + 1 < n_units) load_unit(u + 1, B0, B1, B2, B3);
        compute_unit(u, A0, A1, A2, A3);
        if (u + 1 >= n_units) break;
        if (u + 2 < n_units) load_unit(u + 2, A0, A1, A2, A3);
        compute_unit(u + 1, B0, B1, B2, B3);
    }
}

template <bool ACT_REGS>
__global__ __launch_bounds__(256, 3) void mmvq_fast_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    const int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = ((int)blockIdx.x - a.seg_block0[s]) * 4 + wave;
    const int nw = nblk * 4;
    switch (a.seg[s].type) {
        case T_Q4_K: run_fast<T_Q4_K, ACT_REGS>(a, a.seg[s], smem, gw, nw); break;
        case T_Q5_K: run_fast<T_Q5_K, ACT_REGS>(a, a.seg[s], smem, gw, nw); break;
        case T_Q6_K: run_fast<T_Q6_K, ACT_REGS>(a, a.seg[s], smem, gw, nw); break;
        default: break;
    }
}

}  // namespace

bool mmvq_fast_applicable(const MMVQArgs &a) {
    if (a.T != 1 || (a.K % 2048) != 0) return false;
    const int n = a.epi == EPI_SWIGLU ? 2 : a.n_seg;
    for (int s = 0; s < n; s++)
        if (a.seg[s].type != T_Q4_K && a.seg[s].type != T_Q5_K && a.seg[s].type != T_Q6_K) return false;
    if (a.fuse_mode != 0 && (a.K > 8192 || (a.K & 1023))) return false;
    return true;
}

hipError_t launch_mmvq_fast(MMVQArgs a, hipStream_t st) {
    if (a.epi == EPI_SWIGLU && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type)) return hipErrorInvalidValue;
    const int n_work_seg = a.epi == EPI_SWIGLU ? 1 : a.n_seg;
    // persistent grid: k workgroups per CU (k <= 3, launch bound 3 waves / SIMD); pick the k whose unit count per wave
    // divides most evenly (e.g. 14336 pairs over 2 x 256 x 4 waves = exactly 7 units each), ties -> larger k
    int total_pairs_all = 0, nchunk_all = 1;
    for (int s = 0; s < (a.epi == EPI_SWIGLU ? 1 : a.n_seg); s++) total_pairs_all += a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
    nchunk_all = ((a.K >> 11) + 1) >> 1;
    int best_k = 3;
    double best_cost = 1e30;
    for (int k = 3; k >= 1; k--) {
        const long waves = (long)num_cu() * k * 4;
        const long rounds = (total_pairs_all + waves - 1) / waves;          // pairs per wave, rounded up
        // time ~ rounds * chunks-per-pair, with fewer resident waves costing a little latency hiding
        const double cost = (double)rounds * nchunk_all * (k == 3 ? 1.0 : k == 2 ? 1.04 : 1.15);
        if (cost < best_cost - 1e-9) { best_cost = cost; best_k = k; }
    }
    const int max_blocks = num_cu() * best_k;
    size_t bytes[3] = {0, 0, 0}, total = 0;
    int want[3] = {0, 0, 0}, sum_want = 0;
    for (int s = 0; s < n_work_seg; s++) {
        bytes[s] = (size_t)a.seg[s].n_rows * a.seg[s].row_bytes;
        total += bytes[s];
        const int pairs = a.epi == EPI_SWIGLU ? a.seg[s].n_rows : (a.seg[s].n_rows + 1) / 2;
        want[s] = (pairs + 3) / 4;
        sum_want += want[s];
    }
    a.seg_block0[0] = 0;
    for (int s = 0; s < n_work_seg; s++) {
        int nb = want[s];
        if (sum_want > max_blocks) {
            nb = (int)((double)max_blocks * (double)bytes[s] / (double)total + 0.5);
            if (nb < 1) nb = 1;
            if (nb > want[s]) nb = want[s];
        }
        a.seg_block0[s + 1] = a.seg_block0[s] + nb;
    }
    for (int s = n_work_seg; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n_work_seg];
    const int blocks = a.seg_block0[n_work_seg];
    if (a.epi == EPI_SWIGLU) a.n_seg = 1;
    const size_t K = (size_t)a.K;
    size_t lds = K + (((K >> 8) * 4 + 15) & ~(size_t)15) + (((K >> 4) * 2 + 15) & ~(size_t)15);
    lds = (lds + 15) & ~(size_t)15;
    a.red_off = (int)lds;
    lds += 64;
    if (a.K <= 4096) hipLaunchKernelGGL(mmvq_fast_kernel<true>, dim3(blocks), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(mmvq_fast_kernel<false>, dim3(blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

}  // namespace mi355
