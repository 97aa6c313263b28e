// mmvq_stream.hip — single-token quantised mat-vec as a weight STREAM: every wave copies its contiguous share of the
// weight rows HBM -> LDS with the DMA form of the global load (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per
// instruction, no register round trip), into a private 16 KiB ring, and decodes rows out of the ring while the next
// groups are in flight.  Same arithmetic, lane roles and summation order as mmvq_fast.hip (ggml_vec_dot_q{4,5,6}_K_q8_K /
// q8_0_q8_0): the two kernels are bit-identical (tools/exp_stream.hip, tests/test_gpu_ops.py).
//
// Why (measured): a decode step is ~160 dependent launches of 9-66 MB; each paid 3-5 us on top of its
// bytes because the first weight request of a wave waited for the activation prologue, the register ring held at most
// 9 KB per wave, and the drain of one unit gated the request of the next.  Here
//   * every byte of a wave's first 16 KiB is requested in the first few hundred cycles of the kernel, before the
//     activation exists (128 KiB per CU in flight: all of attn_output / Q/K/V, half of gate/up);
//   * the DMA is issued from inline asm, so hipcc's waitcnt pass does not see it: with the builtin form it puts
//     s_waitcnt vmcnt(0) before every ds_read that follows (it cannot prove the LDS ranges differ), which serialises
//     request and decode — the 4.1 TB/s "LDS-DMA ring" of round 1 (tools/bench_stream.hip) was that, not the hardware;
//     the ring is drained with counted waits (vmcnt(4 k): groups complete in order);
//   * the f32 inputs of the fused prologues (RMSNorm / Q8_K quantise) are requested first, by asm loads into registers,
//     and waited for with the same counter, so the prologue costs no extra round trip;
//   * a row never leaves its wave: no barrier after the prologue, results are collected one per lane and stored with
//     one coalesced store per wave (+ residual, prefetched by DMA; or SwiGLU of a gate/up pair).
// Weight stream policy: non-temporal (each byte is read once per token).
#include "mmvq_fast_dev.h"

namespace mi355 {

namespace {

constexpr int ST_NT = 512, ST_NW = ST_NT / 64;
constexpr int ST_RING = 16384;                  // per wave, power of two (ring offsets wrap with a mask)
constexpr int ST_GC = 4;                        // DMA instructions (1 KiB each) per group
constexpr int ST_GB = ST_GC * 1024;             // the unit of issue and of the counted wait
constexpr int ST_RG = ST_RING / ST_GB;          // groups the ring holds
constexpr unsigned ST_MASK = ST_RING - 1;
constexpr int ST_MAX_STEP = ST_RING - ST_GB;    // bytes one decode step may span (row or row pair): always issuable

// ---- the DMA forms.  M0 = LDS byte address of the 64-lane destination (lane l lands at M0 + 16 l / 4 l); saved and
// restored inside the statement (hipcc does not preserve M0 around asm and does not expect it changed).
template <bool NTL> __device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NTL) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                          : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// at most n of this wave's vector-memory operations still outstanding (n wave-uniform, 0..20; they complete in order)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define C(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
#ifdef MI355_STREAM_PROBE
// tools/exp_stream.hip: per wave, six 100 MHz wall-clock stamps (entry, first groups issued, activation ready, first row
// landed, last row decoded, outputs stored)
__device__ unsigned long long *g_stream_probe = nullptr;
#define ST_SLOT(i) g_stream_probe[(((size_t)(a.nck >> 2) * 256 + blockIdx.x) * ST_NW + wave) * 8 + (i)]
#define ST_STAMP(i) do { if (g_stream_probe && lane == 0) ST_SLOT(i) = wall_clock64(); } while (0)
// accumulated phase times of the decode loop: slots 3 (waiting for data), 6 (decoding), 7 (refilling)
#define ST_ACC_DECL unsigned long long st_t0 = 0, st_acc_w = 0, st_acc_d = 0, st_acc_r = 0
#define ST_T0() do { st_t0 = wall_clock64(); } while (0)
#define ST_ACC(x) do { const unsigned long long t_ = wall_clock64(); x += t_ - st_t0; st_t0 = t_; } while (0)
#define ST_ACC_OUT() do { if (g_stream_probe && lane == 0) { ST_SLOT(3) = st_acc_w; ST_SLOT(6) = st_acc_d; ST_SLOT(7) = st_acc_r; } } while (0)
#else
#define ST_STAMP(i) do { } while (0)
#define ST_ACC_DECL do { } while (0)
#define ST_T0() do { } while (0)
#define ST_ACC(x) do { } while (0)
#define ST_ACC_OUT() do { } while (0)
#endif
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // low half of a flat LDS address = LDS byte offset
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- decoders reading a row out of the ring (row_off = ring offset of the row's first byte; everything is 16-B granular,
// so a piece never straddles the wrap).  Same fields as Raw<TYPE>::load of mmvq_fast_dev.h.
#define RO(x) (ring + (((x)) & ST_MASK))
template <int TYPE> __device__ __forceinline__ void ring_load(Raw<TYPE> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L);
template <> __device__ __forceinline__ void ring_load<T_Q4_K>(Raw<T_Q4_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 144u;
    r.hdr = lds16(RO(b));
    r.q = lds16(RO(b + 16u + (unsigned)L.v * 16u));
}
template <> __device__ __forceinline__ void ring_load<T_Q5_K>(Raw<T_Q5_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 176u;
    r.hdr = lds16(RO(b));
    r.qh = lds16(RO(b + 16u + (unsigned)L.h * 16u));
    r.q = lds16(RO(b + 48u + (unsigned)L.v * 16u));
}
template <> __device__ __forceinline__ void ring_load<T_Q6_K>(Raw<T_Q6_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    r.ql = lds16(RO(row_off + (unsigned)sb * 128u + (unsigned)L.v * 16u));
    r.qh = lds16(RO(row_off + (unsigned)nb * 128u + (unsigned)sb * 64u + (unsigned)L.n * 32u + (unsigned)(L.w & 1) * 16u));
    const unsigned so = row_off + (unsigned)nb * 192u + (unsigned)sb * 16u + 8u * (unsigned)L.n + (unsigned)L.w;
    r.sc_lo = *reinterpret_cast<const int8_t *>(RO(so));
    r.sc_hi = *reinterpret_cast<const int8_t *>(RO(so + 4u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 208u + (unsigned)sb * 2u));
}
template <> __device__ __forceinline__ void ring_load<T_Q8_0>(Raw<T_Q8_0> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 256u + (unsigned)L.v * 32u;
    r.q0 = lds16(RO(b));
    r.q1 = lds16(RO(b + 16u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 256u + ((unsigned)sb * 8u + (unsigned)L.v) * 2u));
}
#undef RO

// ---- LDS layout (bytes from smem): activation | reduction scratch | residual prefetch | rings
struct StLayout {
    int qs, d, bs, red, resid, ring, total;
};
__host__ __device__ inline StLayout st_layout(int kb) {
    const int Kp = kb * 2048;
    StLayout l;
    l.qs = 0;
    l.d = Kp;                                   // one DMA piece (1 KiB) of room: nb * 4 B of scales
    l.bs = Kp + 1024;                           // Kp / 8 B of block sums (Q8_0: f32 block scales, Kp / 8 B too), in whole DMA pieces
    l.red = l.bs + ((Kp / 8 + 1023) & ~1023);
    l.resid = l.red + 128;
    l.ring = l.resid + ST_NW * 256;
    l.total = l.ring + ST_NW * ST_RING;
    return l;
}

// ---- fused prologue inputs: asm loads (hipcc must not count them: its own wait would drain the weight DMA behind them)
template <int KB, int FUSE>
__device__ __forceinline__ void pre_issue(const MMVQArgs &a, f32x4_t (&rxv)[StageDims<KB, ST_NT>::NJW], f32x4_t (&rwv)[StageDims<KB, ST_NT>::NJW]) {
    using S = StageDims<KB, ST_NT>;
    const int tid = tid_now();
    const int lane = tid & 63, wave = tid >> 6;
    const int nbt = a.K >> 8;
#pragma unroll
    for (int j = 0; j < S::NJW; j++) {
        const int b = wave + S::NW * j;
        const int bc = b < nbt ? b : nbt - 1;   // clamped: always a valid address, result unused when b is out of range
        const float *px = a.nx + bc * 256 + lane * 4;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rxv[j]) : "v"(px) : "memory");
        if (FUSE == 1) {
            const float *pw = a.nw + bc * 256 + lane * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rwv[j]) : "v"(pw) : "memory");
        }
    }
}
template <int N> __device__ __forceinline__ void pin_regs(f32x4_t (&r)[N]) {
#pragma unroll
    for (int j = 0; j < N; j++) asm volatile("" : "+v"(r[j]));
}

// RMSNorm * w and / or the Q8_K (Q8_0) quantisation of the token, result in LDS: stage_finish of mmvq_fast_dev.h on this
// kernel's layout (wave w owns 256-blocks w, w + 8, ..; sum of squares in double, fixed order)
template <int KB, int FUSE, bool Q80>
__device__ __forceinline__ void pre_finish(const MMVQArgs &a, const f32x4_t (&rxv)[StageDims<KB, ST_NT>::NJW], const f32x4_t (&rwv)[StageDims<KB, ST_NT>::NJW],
                                           uint8_t *smem, const StLayout &lay) {
    using S = StageDims<KB, ST_NT>;
    const int nbt = a.K >> 8;
    const int tid = tid_now();
    int8_t *qs = reinterpret_cast<int8_t *>(smem + lay.qs);
    float *d = reinterpret_cast<float *>(smem + lay.d);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + lay.bs);
    double *red = reinterpret_cast<double *>(smem + lay.red);
    const int lane = tid & 63, wave = tid >> 6;
    float scale = 1.0f;
    if (FUSE == 1) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < S::NJW; j++) {
            const f32x4_t v = rxv[j];
            double t = 0.0;
            t += (double)(v.x * v.x); t += (double)(v.y * v.y); t += (double)(v.z * v.z); t += (double)(v.w * v.w);
            if (wave + S::NW * j < nbt) sum += t;
        }
        sum = wave_sum(sum);
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < S::NW; w++) tot += red[w];
        const float mean = (float)(tot / (double)a.K);
        scale = 1.0f / sqrtf(mean + a.neps);
    }
#pragma unroll
    for (int j = 0; j < S::NJW; j++) {
        const int b = wave + S::NW * j;
        if (b >= nbt) continue;                                // wave-uniform
        const int e0 = b * 256 + lane * 4;
        f32x4_t v = rxv[j];
        if (FUSE == 1) {
            const f32x4_t ww = rwv[j];
            v.x = (v.x * scale) * ww.x; v.y = (v.y * scale) * ww.y; v.z = (v.z * scale) * ww.z; v.w = (v.w * scale) * ww.w;
        }
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if (Q80) {
            uint32_t packed; float dd;
            wave_quant_q80(vv, packed, dd);
            *reinterpret_cast<uint32_t *>(qs + e0) = packed;
            if ((lane & 7) == 0) reinterpret_cast<float *>(bs)[b * 8 + (lane >> 3)] = h2f(f2h(dd));
            continue;
        }
        uint32_t packed; int bsum; float dq;
        wave_quant_q8k(vv, lane, packed, bsum, dq);
        *reinterpret_cast<uint32_t *>(qs + e0) = packed;
        if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum;
        if (lane == 0) d[b] = dq;
    }
    __syncthreads();
}

// One wave: n_out outputs; output j is row  row0 + (j / UO) * ustride + j % UO  of the segment, UO = 2 for row pairs, 1
// otherwise (SWIGLU: output j = gate row j and up row j, adjacent in the stream).  Returns after its outputs are stored.
template <int TYPE, int KB, int FUSE, bool SWIGLU, bool PAIR>
__device__ __forceinline__ void run_stream(const MMVQArgs &a, const MMVQSeg &sg, uint8_t *smem, const StLayout &lay, int row0, int ustride, int n_out) {
    constexpr int UO = (PAIR && !SWIGLU) ? 2 : 1;
    auto phys = [&](int j) { return row0 + (UO == 2 ? (j >> 1) * ustride + (j & 1) : j * ustride); };
    using R = Raw<TYPE>;
    constexpr bool ACT_REGS = KB <= 2;
    constexpr int STEP = (SWIGLU || PAIR) ? 2 : 1;                // logical rows decoded together
    const int lane = tid_now() & 63;
    const int wave = uni(tid_now() >> 6);
    const LaneRole L = make_role<TYPE>(lane);
    const int nb = a.K >> 8;
    const unsigned rb = (unsigned)sg.row_bytes;
    const uint8_t *W0 = sg.W, *W1 = SWIGLU ? a.seg[1].W : sg.W;
    const int n_lr = SWIGLU ? 2 * n_out : n_out;                  // logical rows of the stream
    const unsigned share = (unsigned)n_lr * rb;
    const int NG = uni((int)((share + ST_GB - 1) / ST_GB));
    uint8_t *ring = smem + lay.ring + wave * ST_RING;
    const unsigned ring_lds = lds_addr(ring);
    ST_STAMP(0);

    // ---- 0. prologue requests that must land before the weights: f32 inputs of the fused modes (registers), the
    //         residual of this wave's rows and (FUSE 0) this wave's pieces of the quantised activation planes (DMA)
    f32x4_t rxv[StageDims<KB, ST_NT>::NJW], rwv[StageDims<KB, ST_NT>::NJW];
    if (FUSE != 0) pre_issue<KB, FUSE>(a, rxv, rwv);
    if (!SWIGLU && a.epi == EPI_ADD) {
        const int j = lane < n_out ? lane : (n_out > 0 ? n_out - 1 : 0);
        if (n_out > 0) dma4(sg.resid + phys(j), lds_addr(smem + lay.resid + wave * 256));
    }
    if (FUSE == 0) {
        // pieces of 1 KiB: qs [K], then d [nb * 4], then bs [K / 8] (Q8_0 segments: not supported in this mode)
        const int nq = a.K >> 10, nbs = ((a.K >> 3) + 1023) >> 10;
        for (int c = wave; c < nq + 1 + nbs; c += ST_NW) {
            const uint8_t *src; int size; unsigned dst;
            if (c < nq) { src = reinterpret_cast<const uint8_t *>(a.aq) + c * 1024; size = 1024; dst = lay.qs + c * 1024; }
            else if (c == nq) { src = reinterpret_cast<const uint8_t *>(a.ad); size = nb * 4; dst = lay.d; }
            else { const int i = c - nq - 1; src = reinterpret_cast<const uint8_t *>(a.abs) + i * 1024; size = (a.K >> 3) - i * 1024; if (size > 1024) size = 1024; dst = lay.bs + i * 1024; }
            const int o = lane * 16 < size ? lane * 16 : size - 16;
            dma16<false>(src + o, lds_addr(smem + dst));
        }
    }

    // ---- 1. the weight stream.  Per lane: (lr, within) = logical row and byte inside it of this lane's 16 B of the NEXT
    //         piece; pieces are issued strictly in order, so the cursor only moves forward.
    int lr = 0;
    unsigned within = (unsigned)lane * 16u;
    while (within >= rb) { within -= rb; lr++; }
    unsigned piece_off = (unsigned)lane * 16u;                    // logical byte offset of this lane in the next piece
    const int last_lr = n_lr - 1;
    auto issue_group = [&](int g) {
        const unsigned dst = ring_lds + (unsigned)(g & (ST_RG - 1)) * ST_GB;
#pragma unroll
        for (int k = 0; k < ST_GC; k++) {
            const bool valid = piece_off < share;                 // past the share: every such lane re-reads the share's last 16 B
            const int lr_u = valid ? lr : last_lr;
            const unsigned wi_u = valid ? within : rb - 16u;
            const int row = phys(SWIGLU ? (lr_u >> 1) : lr_u);
            const uint8_t *base = (SWIGLU && (lr_u & 1)) ? W1 : W0;
            dma16<true>(base + (size_t)row * rb + wi_u, dst + k * 1024);
            piece_off += 1024u;
            within += 1024u;
            while (within >= rb) { within -= rb; lr++; }
        }
    };
    int issued = 0;
    if (n_lr > 0)
        for (; issued < NG && issued < ST_RG; issued++) issue_group(issued);
    ST_STAMP(1);

    // ---- 2. activation into LDS / registers.  Everything requested in step 0 precedes the `issued` groups in this wave's queue.
    if (FUSE != 0) {
        wait_vm(issued * ST_GC);
        pin_regs(rxv);
        if (FUSE == 1) pin_regs(rwv);
        pre_finish<KB, FUSE, TYPE == T_Q8_0>(a, rxv, rwv, smem, lay);
    } else {
        wait_vm(issued * ST_GC);
        __syncthreads();
    }
    ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
    ActSlice S0, S1;
    if (ACT_REGS) {
        S0 = read_slice_t<TYPE>(AL, L.sbl, nb, L);
        if (KB > 1) S1 = read_slice_t<TYPE>(AL, 8 + L.sbl, nb, L);
    }

    ST_STAMP(2);
    // ---- 3. decode rows out of the ring
    float res = 0.0f;                                             // lane i: output i of this wave (64 per flush)
    int n_done = 0, flushed = 0;
    auto flush = [&](int upto) {                                  // outputs [flushed, upto) are in lanes 0 ..
        const int cnt = upto - flushed;
        if (lane < cnt) {
            const int row = phys(flushed + lane);
            float v = res;
            if (!SWIGLU && a.epi == EPI_ADD) {
                const float rsd = flushed == 0 ? reinterpret_cast<const float *>(smem + lay.resid + wave * 256)[lane] : sg.resid[row];
                v = rsd + v;
            }
            sg.out[row] = v;
        }
        flushed = upto;
    };
    ST_ACC_DECL;
    ST_T0();
    for (int i = 0; i < n_lr; i += STEP) {
        const bool two = STEP == 2 && i + 1 < n_lr;
        const unsigned off0 = (unsigned)i * rb;
        const unsigned end = off0 + (two ? 2u : 1u) * rb;
        const int g_need = (int)((end - 1u) / ST_GB);
        wait_vm((issued - 1 - g_need) * ST_GC);
        ST_ACC(st_acc_w);
        float acc0 = 0.0f, acc1 = 0.0f;
        // K <= 4096: both passes in flight at once; longer rows: two passes at a time (fully unrolled, the compiler hoists
        // every pass's ring reads to the top: 246+ registers at 7 passes)
#pragma unroll KB <= 2 ? KB : 2
        for (int p = 0; p < KB; p++) {
            const ActSlice sl = ACT_REGS ? (p == 0 ? S0 : S1) : read_slice_t<TYPE>(AL, p * 8 + L.sbl, nb, L);
            int sb = p * 8 + L.sbl;
            if (sb >= nb) sb = nb - 1;                            // tail of a partial last pass: any valid block, its slice scale is zero
            R w0, w1;
            ring_load<TYPE>(w0, ring, off0, nb, sb, L);
            if (STEP == 2) ring_load<TYPE>(w1, ring, two ? off0 + rb : off0, nb, sb, L);
            acc0 += w0.dot(sl, L);
            if (STEP == 2) acc1 += w1.dot(sl, L);
        }
        const float v0 = wave_sum(acc0);
        const float v1 = STEP == 2 ? wave_sum(acc1) : 0.0f;
        asm volatile("" ::: "memory");                            // the ring reads above stay above the refill below
        ST_ACC(st_acc_d);
        // refill: group `issued` lands on the slot of group issued - RG, free once every byte of it is consumed
        while (issued < NG && (unsigned)(issued - ST_RG + 1) * ST_GB <= end) { issue_group(issued); issued++; }
        ST_ACC(st_acc_r);
        if (SWIGLU) {
            const float y = (v0 / (1.0f + expf(-v0))) * v1;
            if (lane == (n_done & 63)) res = y;
            n_done++;
        } else {
            if (lane == (n_done & 63)) res = v0;
            n_done++;
            if (two) {
                if ((n_done & 63) == 0) flush(n_done);
                if (lane == (n_done & 63)) res = v1;
                n_done++;
            }
        }
        if ((n_done & 63) == 0) flush(n_done);
    }
    ST_STAMP(4);
    ST_ACC_OUT();
    if (flushed < n_done) flush(n_done);
#ifdef MI355_STREAM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ST_STAMP(5);
#endif
}

template <int KB, int FUSE>
__global__ __launch_bounds__(ST_NT) void mmvq_stream_kernel(const MMVQArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const StLayout lay = st_layout(KB);
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg_block0[2]) s = 2;
    const int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
    const int bl = (int)blockIdx.x - a.seg_block0[s];
    const MMVQSeg &sg = a.seg[s];
    const int wave = uni(tid_now() >> 6);
    // contiguous rows per workgroup, then per wave
    const int rpb = (sg.n_rows + nblk - 1) / nblk;
    int b0 = bl * rpb, b1 = b0 + rpb;
    if (b0 > sg.n_rows) b0 = sg.n_rows;
    if (b1 > sg.n_rows) b1 = sg.n_rows;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const bool pair = 2 * sg.row_bytes <= (size_t)ST_MAX_STEP;
    // which rows a wave takes (a.nck, experiments): 0 = a contiguous run of its workgroup's rows; 1 = every 8th unit of its
    // workgroup's rows (the workgroup reads one moving window); 2 = every (8 * workgroups)th unit of the segment
    const int uo = (pair && !swiglu) ? 2 : 1;
    int r0, ustride, n_out;
    {
        int lo, hi, first, stride;                                 // units [first, first + stride, ..) of rows [lo, hi)
        if ((a.nck & 3) == 2) { lo = 0; hi = sg.n_rows; first = bl * ST_NW + wave; stride = nblk * ST_NW; }
        else if ((a.nck & 3) == 1) { lo = b0; hi = b1; first = wave; stride = ST_NW; }
        else {
            const int upb = (b1 - b0 + uo - 1) / uo, upw = (upb + ST_NW - 1) / ST_NW;
            lo = b0 + wave * upw * uo; hi = lo + upw * uo;
            if (lo > b1) lo = b1;
            if (hi > b1) hi = b1;
            first = 0; stride = 1;
        }
        r0 = lo + first * uo; ustride = stride * uo;
        const int avail = hi - r0;                                 // rows from the first unit to the end of the range
        if (avail <= 0) n_out = 0;
        else {
            const int nu = (avail + ustride - 1) / ustride;        // units that start inside the range
            const int last = avail - (nu - 1) * ustride;           // rows of the last one
            n_out = (nu - 1) * uo + (last < uo ? last : uo);
        }
    }
    // forms that exist: SwiGLU pairs only with the fused RMSNorm prologue (gate/up), row pairs up to K = 8192, single rows
    // from K = 6144 (mmvq_stream_applicable agrees)
#define RUN(TY)                                                                                               \
    do {                                                                                                      \
        if (swiglu) { if constexpr (FUSE == 1) run_stream<TY, KB, FUSE, true, true>(a, sg, smem, lay, r0, ustride, n_out); } \
        else if (pair) { if constexpr (KB <= 4) run_stream<TY, KB, FUSE, false, true>(a, sg, smem, lay, r0, ustride, n_out); } \
        else { if constexpr (KB >= 3) run_stream<TY, KB, FUSE, false, false>(a, sg, smem, lay, r0, ustride, n_out); } \
    } while (0)
    switch (sg.type) {
        case T_Q4_K: RUN(T_Q4_K); break;
        case T_Q5_K: RUN(T_Q5_K); break;
        case T_Q6_K: RUN(T_Q6_K); break;
        case T_Q8_0: if constexpr (FUSE != 0) RUN(T_Q8_0); break;
        default: break;
    }
#undef RUN
}

}  // namespace

#ifdef MI355_STREAM_PROBE
void mmvq_stream_set_probe(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_probe), &p, sizeof(p)); }
#endif

bool mmvq_stream_applicable(const MMVQArgs &a) {
    if (a.T != 1 || a.K <= 0 || (a.K % 1024) != 0) return false;
    if (a.n_sel > 1) return false;
    const int kb = (a.K + 2047) >> 11;
    if (kb != 1 && kb != 2 && kb != 3 && kb != 4 && kb != 6 && kb != 7) return false;
    if (a.fuse_mode < 0 || a.fuse_mode > 2) return false;
    if (a.fuse_mode == 1 && a.K > 8192) return false;
    const bool swiglu = a.epi == EPI_SWIGLU;
    if (swiglu && a.fuse_mode != 1) return false;
    if (swiglu && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type || a.seg[0].n_rows != a.seg[1].n_rows || a.seg[0].row_bytes != a.seg[1].row_bytes)) return false;
    const int n = swiglu ? 1 : a.n_seg;
    for (int s = 0; s < (swiglu ? 2 : n); s++) {
        const MMVQSeg &g = a.seg[s];
        const int t = g.type;
        if (t != T_Q4_K && t != T_Q5_K && t != T_Q6_K && t != T_Q8_0) return false;
        if (g.expert_sel) return false;
        if ((g.row_bytes % 16) != 0 || g.row_bytes < 1024) return false;
        if ((swiglu ? 2 : 1) * g.row_bytes > (size_t)ST_MAX_STEP) return false;
        if (kb <= 2 && 2 * g.row_bytes > (size_t)ST_MAX_STEP) return false;      // K <= 4096 only has the row-pair form
        if (kb >= 6 && 2 * g.row_bytes <= (size_t)ST_MAX_STEP) return false;     // K >= 10240 only the single-row form
        if ((reinterpret_cast<uintptr_t>(g.W) & 15) != 0) return false;
        if (a.fuse_mode == 0 && (t == T_Q8_0 || !a.aq || !a.ad || !a.abs)) return false;
    }
    if (a.fuse_mode == 0 && (((uintptr_t)a.aq | (uintptr_t)a.ad | (uintptr_t)a.abs) & 15) != 0) return false;
    if (a.fuse_mode != 0 && (reinterpret_cast<uintptr_t>(a.nx) & 15) != 0) return false;
    if (a.fuse_mode == 1 && (reinterpret_cast<uintptr_t>(a.nw) & 15) != 0) return false;
    return true;
}

// workgroups per segment in proportion to its bytes (every workgroup streams about the same number of bytes)
static void stream_plan(MMVQArgs &a, int max_blocks) {
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int n = swiglu ? 1 : a.n_seg;
    double bytes[3] = {0, 0, 0}, total = 0;
    for (int s = 0; s < n; s++) { bytes[s] = (double)a.seg[s].n_rows * (double)a.seg[s].row_bytes; total += bytes[s]; }
    int left = max_blocks;
    a.seg_block0[0] = 0;
    for (int s = 0; s < n; s++) {
        int nb = s == n - 1 ? left : (int)((double)max_blocks * bytes[s] / total + 0.5);
        if (nb < 1) nb = 1;
        if (nb > left - (n - 1 - s)) nb = left - (n - 1 - s);
        if (nb > a.seg[s].n_rows) nb = a.seg[s].n_rows;
        a.seg_block0[s + 1] = a.seg_block0[s] + nb;
        left -= nb;
    }
    for (int s = n; s < 3; s++) a.seg_block0[s + 1] = a.seg_block0[n];
    if (swiglu) a.n_seg = 1;
}

hipError_t launch_mmvq_stream(MMVQArgs a, hipStream_t st) {
    if (!mmvq_stream_applicable(a)) return hipErrorInvalidValue;
    const int kb = (a.K + 2047) >> 11;
    stream_plan(a, num_cu());
    const int blocks = a.seg_block0[3];
    const size_t lds = (size_t)st_layout(kb).total;
#define STREAM(KBV, FZ)                                                                                                  \
    do {                                                                                                                 \
        static bool attr_set = false;                                                                                    \
        if (!attr_set) {                                                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mmvq_stream_kernel<KBV, FZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                               \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL((mmvq_stream_kernel<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, a);                       \
    } while (0)
#define STREAM_F(KBV) do { if (a.fuse_mode == 0) STREAM(KBV, 0); else if (a.fuse_mode == 1) STREAM(KBV, 1); else STREAM(KBV, 2); } while (0)
    switch (kb) {
        case 1: STREAM_F(1); break;
        case 2: STREAM_F(2); break;
        case 3: STREAM_F(3); break;
        case 4: STREAM_F(4); break;
        case 6: if (a.fuse_mode == 2) STREAM(6, 2); else if (a.fuse_mode == 0) STREAM(6, 0); else return hipErrorInvalidValue; break;
        case 7: if (a.fuse_mode == 2) STREAM(7, 2); else if (a.fuse_mode == 0) STREAM(7, 0); else return hipErrorInvalidValue; break;
        default: return hipErrorInvalidValue;
    }
#undef STREAM_F
#undef STREAM
    return hipGetLastError();
}

}  // namespace mi355
