// mmvq_stream.hip — single-token quantised mat-vec as a weight STREAM.  One workgroup of 8 waves per CU owns a contiguous
// run of weight rows.  Wave 0 is the LOADER: it copies that run HBM -> LDS with the DMA form of the global load
// (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction, no register round trip) into a 128 KiB ring, 4 KiB
// slots, up to 12 slots in flight, and publishes a slot once it has landed.  Waves 1..7 are CONSUMERS: they quantise the
// activation (RMSNorm * w -> Q8_K, the fused prologues) while the ring fills, then decode row pairs out of the ring as
// the slots are published, round-robin in stream order, and report their progress so that the loader can reuse the space.
// Same arithmetic, lane roles and summation order as mmvq_fast.hip (ggml_vec_dot_q{4,5,6}_K_q8_K / q8_0_q8_0, SURVEY.md
// §8a a8): the two kernels are bit-identical (tools/exp_stream.hip, tests/test_gpu_ops.py).
//
// Why (measured on MI355X, tools/exp_stream.hip):
//   * a DMA-only kernel reads these tensors at the chip's streaming rate (33 MB in 6.4 us in a graph chain = 6.4 TB/s behind
//     a 1.2 us boundary) — wave-private, workgroup-window or chip-wide interleaved alike; a register-load kernel of the same
//     shape needs 7.5 us.  hipcc serialises the BUILTIN form: it cannot prove that a later ds_read misses the DMA's LDS
//     range and puts s_waitcnt vmcnt(0) in front of every one (the 4.1 TB/s "LDS-DMA ring" of round 1 was that).  Issued
//     from inline asm the DMA is invisible to that pass; completion is tracked by hand (vmcnt counts, in-order return);
//   * a wave that issues its own DMA (v1 of this file: a 16 KiB ring per wave) is BLOCKED in the issue for microseconds:
//     the vector-memory queue of a CU is shallow and drains at the HBM rate, so the 16 requests of a ring take 3-5 us to
//     be accepted, then the wave does its prologue (2-3 us, ring full, HBM idle), then decodes (VALU-bound: 0.66 us per
//     row pair) before it may request again.  Request, prologue and decode in series per wave: no faster than the
//     register ring (42 vs 42 us per layer).  With the roles split the loader is the only wave that ever blocks on the
//     memory queue, which is exactly the flow control wanted; nobody else issues vector memory in the loop.
// Weight stream policy: non-temporal (each byte is read once per token).
#include "mmvq_fast_dev.h"

namespace mi355 {

namespace {

#ifndef MI355_STREAM_NL
#define MI355_STREAM_NL 2
#endif
constexpr int ST_NL = MI355_STREAM_NL;          // loader waves (the first waves of the workgroup): 2 or 4
constexpr int ST_NC = 8;                        // consumer waves: the wave count the prologue of mmvq_fast is cut for
constexpr int ST_NW = ST_NL + ST_NC, ST_NT = ST_NW * 64;
constexpr int ST_RING = 131072;                 // bytes, power of two (ring offsets wrap with a mask)
constexpr int ST_SI = 4;                        // DMA instructions (1 KiB each) per slot
constexpr int ST_SLOT = ST_SI * 1024;           // unit of publication
constexpr int ST_D = ST_NL == 2 ? 15 : 8;       // slots in flight per loader (at most 60 of the 63 countable vector-memory operations of a wave)
constexpr unsigned ST_MASK = ST_RING - 1;      // one stream in the whole ring
constexpr unsigned ST_MASK2 = ST_RING / 2 - 1;  // SwiGLU: gate rows in the lower half, up rows in the upper half
constexpr int ST_MAX_STEP = 16384;              // bytes one decode step may span (a row, a row pair or a gate/up pair)
constexpr int ST_PAIR_MAX = 12288;              // rows are decoded two at a time up to this many bytes per pair

// ---- the DMA.  M0 = LDS byte address of the 64-lane destination (lane l lands at M0 + 16 l); saved and restored inside
// the statement (hipcc does not preserve M0 around asm and does not expect it changed).
template <bool NTL> __device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NTL) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                          : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                      : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// one whole slot: lane l copies 16 B from g, g + 1 KiB, g + 2 KiB, g + 3 KiB; the instruction offset applies to the global
// AND to the LDS address (piece i lands at M0 + i KiB + 16 l)
__device__ __forceinline__ void dma_slot(const void *g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048 nt\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072 nt\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm_c() { asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(N) : "memory"); }
#ifdef MI355_STREAM_PROBE
// tools/exp_stream.hip: per wave, 100 MHz wall-clock stamps and accumulated waits
__device__ unsigned long long *g_stream_probe = nullptr;
#define ST_SLOTP(i) g_stream_probe[(((size_t)(a.nck >> 2) * 256 + blockIdx.x) * ST_NW + wave) * 8 + (i)]
#define ST_STAMP(i) do { if (g_stream_probe && lane == 0) ST_SLOTP(i) = wall_clock64(); } while (0)
#define ST_ACC_DECL unsigned long long st_t0 = 0, st_acc_w = 0, st_acc_d = 0
#define ST_T0() do { st_t0 = wall_clock64(); } while (0)
#define ST_ACC(x) do { const unsigned long long t_ = wall_clock64(); x += t_ - st_t0; st_t0 = t_; } while (0)
#define ST_ACC_OUT() do { if (g_stream_probe && lane == 0) { ST_SLOTP(3) = st_acc_w; ST_SLOTP(6) = st_acc_d; } } while (0)
#else
#define ST_STAMP(i) do { } while (0)
#define ST_ACC_DECL do { } while (0)
#define ST_T0() do { } while (0)
#define ST_ACC(x) do { } while (0)
#define ST_ACC_OUT() do { } while (0)
#endif
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(uintptr_t)p; }   // low half of a flat LDS address = LDS byte offset
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// words shared by the waves of the workgroup (LDS is coherent inside a CU; every access is a real ds instruction)
__device__ __forceinline__ int ld_sync(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_sync(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ---- decoders reading a row out of the ring (row_off = stream offset of the row's first byte; everything is 16-B
// granular, so a piece never straddles the wrap).  Same fields as Raw<TYPE>::load of mmvq_fast_dev.h.
#define RO(x) (ring + (((x)) & MASK))
template <unsigned MASK> __device__ __forceinline__ void ring_load(Raw<T_Q4_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 144u;
    r.hdr = lds16(RO(b));
    r.q = lds16(RO(b + 16u + (unsigned)L.v * 16u));
}
template <unsigned MASK> __device__ __forceinline__ void ring_load(Raw<T_Q5_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 176u;
    r.hdr = lds16(RO(b));
    r.qh = lds16(RO(b + 16u + (unsigned)L.h * 16u));
    r.q = lds16(RO(b + 48u + (unsigned)L.v * 16u));
}
template <unsigned MASK> __device__ __forceinline__ void ring_load(Raw<T_Q6_K> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    r.ql = lds16(RO(row_off + (unsigned)sb * 128u + (unsigned)L.v * 16u));
    r.qh = lds16(RO(row_off + (unsigned)nb * 128u + (unsigned)sb * 64u + (unsigned)L.n * 32u + (unsigned)(L.w & 1) * 16u));
    const unsigned so = row_off + (unsigned)nb * 192u + (unsigned)sb * 16u + 8u * (unsigned)L.n + (unsigned)L.w;
    r.sc_lo = *reinterpret_cast<const int8_t *>(RO(so));
    r.sc_hi = *reinterpret_cast<const int8_t *>(RO(so + 4u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 208u + (unsigned)sb * 2u));
}
template <unsigned MASK> __device__ __forceinline__ void ring_load(Raw<T_Q8_0> &r, const uint8_t *ring, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 256u + (unsigned)L.v * 32u;
    r.q0 = lds16(RO(b));
    r.q1 = lds16(RO(b + 16u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 256u + ((unsigned)sb * 8u + (unsigned)L.v) * 2u));
}
#undef RO

// ---- LDS layout (bytes from smem): activation | reduction scratch | sync words | ring
struct StLayout {
    int qs, d, bs, red, sync, ring, total;
};
__host__ __device__ inline StLayout st_layout(int kb) {
    const int Kp = kb * 2048;
    StLayout l;
    l.qs = 0;
    l.d = Kp;                                   // one DMA piece (1 KiB) of room: nb * 4 B of scales
    l.bs = Kp + 1024;                           // Kp / 8 B of block sums (Q8_0: f32 block scales, Kp / 8 B too), in whole DMA pieces
    l.red = l.bs + ((Kp / 8 + 1023) & ~1023);
    l.sync = l.red + 128;
    l.ring = l.sync + 128;
    l.total = l.ring + ST_RING;
    return l;
}
// sync words (ints at smem + lay.sync): [q] slots published by loader q, [4] / [5] arrivals at the two prologue
// rendezvous, [6] consumers whose own global requests (prologue inputs, residual) are in the memory queue: the loaders
// start after that, [8 + c] consumer c: the first of its steps it has NOT finished
enum { SY_LANDED = 0, SY_PRO1 = 4, SY_PRO2 = 5, SY_GO = 6, SY_DONE = 8 };

// every wait on another wave of the workgroup is bounded: the waves of a workgroup are always co-resident, so a healthy
// wait ends within microseconds; if a protocol bug ever broke that, the launch still ends (with wrong numbers, which the
// parity tests catch) instead of hanging the device.  ~0.2 s of polls.
constexpr int ST_SPIN_LIMIT = 1 << 21;
#define ST_SPIN_WHILE(cond, sleep_arg) do { int spins_ = 0; while ((cond) && ++spins_ < ST_SPIN_LIMIT) __builtin_amdgcn_s_sleep(sleep_arg); } while (0)
__device__ __forceinline__ void consumers_rendezvous(int *word, int lane) {
    if (lane == 0) (void)__hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ST_SPIN_WHILE(ld_sync(word) < ST_NC, 1);
}

// What the device code needs of MMVQArgs, read ONCE at the top of the kernel in straight-line code: the kernel argument
// segment is cold at every launch (~0.7 us per miss, tools/exp_stream.hip timeline) and hipcc loads a field where it is
// first used, so `a.seg[s]` behind the segment choice and a.nx behind the role choice were two more serial misses.
struct StArgs {
    int K, epi, nck;
    float neps;
    const int8_t *aq; const float *ad; const int16_t *abs;
    const float *nx, *nw;
    const uint8_t *W1;            // SwiGLU: the up tensor
};
struct StSeg {
    const uint8_t *W; float *out; const float *resid;
    int type, n_rows;
    unsigned row_bytes;
};

// RMSNorm * w and / or the Q8_K (Q8_0) quantisation of the token by the 8 consumer waves, result in LDS: stage_finish of
// mmvq_fast_dev.h (consumer c owns 256-blocks c, c + 8, ..; sum of squares in double, fixed order), with the workgroup
// barrier replaced by a rendezvous of the consumers (the loaders never join: they sit in the memory queue).
template <int KB, int FUSE, bool Q80>
__device__ __forceinline__ void consumer_prologue(const StArgs &a, uint8_t *smem, const StLayout &lay, int c, int lane) {
    constexpr int NJW = (KB * 8 + ST_NC - 1) / ST_NC;
    const int nbt = a.K >> 8;
    int8_t *qs = reinterpret_cast<int8_t *>(smem + lay.qs);
    float *d = reinterpret_cast<float *>(smem + lay.d);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + lay.bs);
    double *red = reinterpret_cast<double *>(smem + lay.red);
    int *sy = reinterpret_cast<int *>(smem + lay.sync);
    f32x4_t rxv[NJW], rwv[NJW];
#pragma unroll
    for (int j = 0; j < NJW; j++) {
        const int b = c + ST_NC * j;
        const int bc = b < nbt ? b : nbt - 1;                  // clamped: always a valid address, result unused when b is out of range
        rxv[j] = *reinterpret_cast<const f32x4_t *>(a.nx + bc * 256 + lane * 4);
        if (FUSE == 1) rwv[j] = *reinterpret_cast<const f32x4_t *>(a.nw + bc * 256 + lane * 4);
    }
    asm volatile("" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    float scale = 1.0f;
    if (FUSE == 1) {
        double sum = 0.0;
#pragma unroll
        for (int j = 0; j < NJW; j++) {
            const f32x4_t x = rxv[j];
            double t = 0.0;
            t += (double)(x.x * x.x); t += (double)(x.y * x.y); t += (double)(x.z * x.z); t += (double)(x.w * x.w);
            if (c + ST_NC * j < nbt) sum += t;
        }
        sum = wave_sum(sum);
        if (lane == 0) red[c] = sum;
        consumers_rendezvous(sy + SY_PRO1, lane);
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < ST_NC; w++) tot += red[w];
        const float mean = (float)(tot / (double)a.K);
        scale = 1.0f / sqrtf(mean + a.neps);
    }
#pragma unroll
    for (int j = 0; j < NJW; j++) {
        const int b = c + ST_NC * j;
        if (b >= nbt) continue;                                // wave-uniform
        const int e0 = b * 256 + lane * 4;
        f32x4_t x = rxv[j];
        if (FUSE == 1) {
            const f32x4_t ww = rwv[j];
            x.x = (x.x * scale) * ww.x; x.y = (x.y * scale) * ww.y; x.z = (x.z * scale) * ww.z; x.w = (x.w * scale) * ww.w;
        }
        const float vv[4] = {x.x, x.y, x.z, x.w};
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
    consumers_rendezvous(sy + SY_PRO2, lane);
}

// ---- a loader wave (q < ST_NL).  The workgroup's rows [b0, b0 + n) are one contiguous run of `total` bytes per tensor.
//   one tensor : the run is cut into 4 KiB slots, slot k lives at ring offset k * 4096 mod 128 KiB, loader q copies the
//                slots k = q mod ST_NL;
//   SwiGLU     : even loaders copy the gate run into the lower half of the ring, odd loaders the up run into the upper
//                half (slot k of either at k * 4096 mod 64 KiB; loader q takes k = q / 2 mod ST_NL / 2).
// FUSE 0: loader 0 first copies the quantised activation planes of the token.
template <int FUSE, bool SWIGLU>
__device__ __forceinline__ void run_loader(const StArgs &a, const StSeg &sg, uint8_t *smem, const StLayout &lay, int b0, unsigned total, unsigned step_bytes, int q) {
    const int lane = tid_now() & 63;
    const int wave = q; (void)wave;
    int *sy = reinterpret_cast<int *>(smem + lay.sync);
    const unsigned rb = (unsigned)sg.row_bytes;
    constexpr int NLT = SWIGLU ? ST_NL / 2 : ST_NL;              // loaders per run
    const int t = SWIGLU ? (q & 1) : 0, u = SWIGLU ? (q >> 1) : q;
    const uint8_t *W = (t == 1 ? a.W1 : sg.W) + (size_t)b0 * rb;
    const unsigned ring_lds = lds_addr(smem + lay.ring) + (t == 1 ? ST_RING / 2 : 0);
    if (FUSE == 0 && q == 0) {
        // pieces of 1 KiB: qs [K], then d [nb * 4], then bs [K / 8]
        const int nb = a.K >> 8;
        const int nq = a.K >> 10, nbs = ((a.K >> 3) + 1023) >> 10;
        for (int c = 0; c < nq + 1 + nbs; c++) {
            const uint8_t *src; int size; unsigned dst;
            if (c < nq) { src = reinterpret_cast<const uint8_t *>(a.aq) + c * 1024; size = 1024; dst = lay.qs + c * 1024; }
            else if (c == nq) { src = reinterpret_cast<const uint8_t *>(a.ad); size = nb * 4; dst = lay.d; }
            else { const int i = c - nq - 1; src = reinterpret_cast<const uint8_t *>(a.abs) + i * 1024; size = (a.K >> 3) - i * 1024; if (size > 1024) size = 1024; dst = lay.bs + i * 1024; }
            const int o = lane * 16 < size ? lane * 16 : size - 16;
            dma16<false>(src + o, lds_addr(smem + dst));
        }
    }
    // the consumers' own requests go first: a load queued behind this wave's 60 KiB waits for all of it
    ST_SPIN_WHILE(ld_sync(sy + SY_GO) < ST_NC, 0);
    ST_STAMP(2);
    const int NS_all = uni((int)((total + ST_SLOT - 1) / ST_SLOT));                 // slots of the run
    const int NS = NS_all > u ? (NS_all - u + NLT - 1) / NLT : 0;                    // slots of this loader
    const unsigned last = total - 16u;
    const unsigned ring_bytes = SWIGLU ? ST_RING / 2 : ST_RING;
    const uint8_t *pl = W + (size_t)u * ST_SLOT + (size_t)lane * 16;                 // this lane's 16 B of the next slot
    auto issue_slot = [&](int j) {
        const unsigned k = (unsigned)NLT * (unsigned)j + (unsigned)u;               // slot of the run
        const unsigned dst = ring_lds + ((k * ST_SLOT) & (ring_bytes - 1));
        if ((k + 1u) * ST_SLOT <= total) dma_slot(pl, dst);
        else {
            const unsigned off0 = k * ST_SLOT + (unsigned)lane * 16u;
#pragma unroll
            for (int i = 0; i < ST_SI; i++) {
                const unsigned off = off0 + i * 1024u;            // past the end: every such lane re-reads the run's last 16 B
                dma16<true>(W + (off < total ? off : last), dst + i * 1024);
            }
        }
        pl += (size_t)NLT * ST_SLOT;
    };
    int issued = 0, published = 0, blocked_polls = 0;
    unsigned free_until = ring_bytes;                             // bytes of the run that may be in the ring: consumed frontier + ring size
    const unsigned consumed_per_step = SWIGLU ? rb : step_bytes;  // bytes of THIS run a decode step retires
    while (issued < NS) {
        const unsigned k = (unsigned)NLT * (unsigned)issued + (unsigned)u;
        if ((k + 1u) * ST_SLOT > free_until) {
            // the ring is full of unconsumed rows: nothing to issue, so everything in flight may as well be waited for
            if (published < issued) { wait_vm_c<0>(); published = issued; st_sync(sy + SY_LANDED + q, published); }
            int f = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < ST_NC; c++) { const int x = uni(ld_sync(sy + SY_DONE + c)); f = x < f ? x : f; }
            const unsigned long long fb = (unsigned long long)(unsigned)f * (unsigned long long)consumed_per_step;
            free_until = fb > 0xffffffffull - ring_bytes ? 0xffffffffu : (unsigned)fb + ring_bytes;
            if ((k + 1u) * ST_SLOT > free_until) {
                if (++blocked_polls > ST_SPIN_LIMIT) break;       // (never on a healthy run)
                __builtin_amdgcn_s_sleep(2);
            }
            continue;
        }
        issue_slot(issued);
        issued++;
        if (issued - published > ST_D) { wait_vm_c<ST_D * ST_SI>(); published = issued - ST_D; st_sync(sy + SY_LANDED + q, published); }
    }
    ST_STAMP(1);
    // drain in order
#define DR(r) if (issued - published > r) { wait_vm_c<(r) * ST_SI>(); published = issued - r; st_sync(sy + SY_LANDED + q, published); }
    if constexpr (ST_D > 8) { DR(14) DR(13) DR(12) DR(11) DR(10) DR(9) DR(8) }
    DR(7) DR(6) DR(5) DR(4) DR(3) DR(2) DR(1) DR(0)
#undef DR
    ST_STAMP(5);
}

// ---- a consumer wave.  Step s of the workgroup = a row pair (UO = 2 outputs), one row (big K) or gate row s + up row s
// (one output); consumer c decodes steps c, c + 8, ..
template <int TYPE, int KB, int FUSE, bool SWIGLU, bool PAIR>
__device__ __forceinline__ void run_consumer(const StArgs &a, const StSeg &sg, uint8_t *smem, const StLayout &lay, int b0, int n_rows_wg, int c) {
    using R = Raw<TYPE>;
    constexpr bool ACT_REGS = KB <= 2;
    constexpr int STEP = (SWIGLU || PAIR) ? 2 : 1;                // rows decoded together
    constexpr int UO = (PAIR && !SWIGLU) ? 2 : 1;                 // outputs per step
    constexpr unsigned MASK = SWIGLU ? ST_MASK2 : ST_MASK;
    const int lane = tid_now() & 63;
    const int wave = c + ST_NL; (void)wave;
    const LaneRole L = make_role<TYPE>(lane);
    const int nb = a.K >> 8;
    const unsigned rb = (unsigned)sg.row_bytes;
    int *sy = reinterpret_cast<int *>(smem + lay.sync);
    const uint8_t *ring = smem + lay.ring;
    const uint8_t *ring1 = SWIGLU ? ring + ST_RING / 2 : ring;    // where the second row of a step is read from
    const int n_steps = (n_rows_wg + UO - 1) / UO;
    auto phys = [&](int j) { return b0 + (UO == 2 ? ((j >> 1) * ST_NC + c) * 2 + (j & 1) : j * ST_NC + c); };
    // outputs of this consumer
    int n_out = 0;
    if (c < n_steps) {
        const int my_steps = (n_steps - c + ST_NC - 1) / ST_NC;
        const int last_step = c + (my_steps - 1) * ST_NC;
        const int rows_last = n_rows_wg - last_step * UO < UO ? n_rows_wg - last_step * UO : UO;
        n_out = (my_steps - 1) * UO + rows_last;
    }
    float rsd = 0.0f;                                             // residual of output `lane` (first 64 outputs), requested now
    if (!SWIGLU && a.epi == EPI_ADD && lane < n_out) rsd = sg.resid[phys(lane)];

    // ---- activation into LDS (fused modes: by the consumers themselves; planes: DMA'd by loader 0 ahead of its slot 0)
    if (FUSE != 0) consumer_prologue<KB, FUSE, TYPE == T_Q8_0>(a, smem, lay, c, lane);
    else {
        asm volatile("" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ST_SPIN_WHILE(ld_sync(sy + SY_LANDED) < 1, 1);
    }
    ST_STAMP(2);
    ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
    ActSlice S0, S1;
    if (ACT_REGS) {
        S0 = read_slice_t<TYPE>(AL, L.sbl, nb, L);
        if (KB > 1) S1 = read_slice_t<TYPE>(AL, 8 + L.sbl, nb, L);
    }

    float res = 0.0f;                                             // lane i: output i of this wave (64 per flush)
    int n_done = 0, flushed = 0;
    auto flush = [&](int upto) {                                  // outputs [flushed, upto) are in lanes 0 ..
        const int cnt = upto - flushed;
        if (lane < cnt) {
            const int row = phys(flushed + lane);
            float v = res;
            if (!SWIGLU && a.epi == EPI_ADD) v = (flushed == 0 ? rsd : sg.resid[row]) + v;
            sg.out[row] = v;
        }
        flushed = upto;
    };
    ST_ACC_DECL;
    ST_T0();
    for (int s = c; s < n_steps; s += ST_NC) {
        const bool two = STEP == 2 && (SWIGLU || s * 2 + 1 < n_rows_wg);
        // ring offsets of the step's rows and the slots that must have landed
        const unsigned off0 = SWIGLU ? (unsigned)s * rb : (unsigned)s * (unsigned)STEP * rb;
        const unsigned off1 = SWIGLU ? off0 : (two ? off0 + rb : off0);
        const unsigned end = SWIGLU ? off0 + rb : off0 + (two ? 2u : 1u) * rb;
        const int n = (int)((end + ST_SLOT - 1) / ST_SLOT);       // slots [0, n) of the run(s)
        {
            constexpr int NLT = SWIGLU ? ST_NL / 2 : ST_NL;
            auto missing = [&]() {
                bool m = false;
#pragma unroll
                for (int q = 0; q < ST_NL; q++) {
                    const int u = SWIGLU ? (q >> 1) : q;
                    m = m || ld_sync(sy + SY_LANDED + q) < (n > u ? (n - u + NLT - 1) / NLT : 0);
                }
                return m;
            };
            ST_SPIN_WHILE(missing(), 1);
        }
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
            ring_load<MASK>(w0, ring, off0, nb, sb, L);
            if (STEP == 2) ring_load<MASK>(w1, ring1, off1, nb, sb, L);
            acc0 += w0.dot(sl, L);
            if (STEP == 2) acc1 += w1.dot(sl, L);
        }
        const float v0 = wave_sum(acc0);
        const float v1 = STEP == 2 ? wave_sum(acc1) : 0.0f;
        asm volatile("" ::: "memory");                            // the ring reads above stay above the release below
        if (lane == 0) st_sync(sy + SY_DONE + c, s + ST_NC);
        ST_ACC(st_acc_d);
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
__global__ __launch_bounds__(ST_NT) void mmvq_stream_kernel(const MMVQArgs ka) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef MI355_STREAM_PROBE
    const unsigned long long t_top = wall_clock64();
#endif
    const StLayout lay = st_layout(KB);
    // every kernel argument the launch can need, unconditionally (one batch of scalar loads)
    const int n_seg = ka.n_seg, sb0 = ka.seg_block0[0], sb1 = ka.seg_block0[1], sb2 = ka.seg_block0[2], sb3 = ka.seg_block0[3];
    const StArgs a{ka.K, ka.epi, ka.nck, ka.neps, ka.aq, ka.ad, ka.abs, ka.nx, ka.nw, ka.seg[1].W};
    const StSeg g0{ka.seg[0].W, ka.seg[0].out, ka.seg[0].resid, ka.seg[0].type, ka.seg[0].n_rows, (unsigned)ka.seg[0].row_bytes};
    const StSeg g1{ka.seg[1].W, ka.seg[1].out, ka.seg[1].resid, ka.seg[1].type, ka.seg[1].n_rows, (unsigned)ka.seg[1].row_bytes};
    const StSeg g2{ka.seg[2].W, ka.seg[2].out, ka.seg[2].resid, ka.seg[2].type, ka.seg[2].n_rows, (unsigned)ka.seg[2].row_bytes};
    // (pinned: without a use here hipcc sinks each load to its first use again)
#define PIN(x) asm volatile("" :: "s"(x))
    PIN(n_seg); PIN(sb0); PIN(sb1); PIN(sb2); PIN(sb3);
    PIN(a.K); PIN(a.epi); PIN(a.nck); PIN(a.neps); PIN(a.aq); PIN(a.ad); PIN(a.abs); PIN(a.nx); PIN(a.nw); PIN(a.W1);
    PIN(g0.W); PIN(g0.out); PIN(g0.resid); PIN(g0.type); PIN(g0.n_rows); PIN(g0.row_bytes);
    PIN(g1.W); PIN(g1.out); PIN(g1.resid); PIN(g1.type); PIN(g1.n_rows); PIN(g1.row_bytes);
    PIN(g2.W); PIN(g2.out); PIN(g2.resid); PIN(g2.type); PIN(g2.n_rows); PIN(g2.row_bytes);
#undef PIN
    int s = 0;
    if (n_seg > 1 && (int)blockIdx.x >= sb1) s = 1;
    if (n_seg > 2 && (int)blockIdx.x >= sb2) s = 2;
    const int lo = s == 0 ? sb0 : s == 1 ? sb1 : sb2, hi = s == 0 ? sb1 : s == 1 ? sb2 : sb3;
    const int nblk = hi - lo;
    const int bl = (int)blockIdx.x - lo;
    StSeg sg;
    sg.W = s == 0 ? g0.W : s == 1 ? g1.W : g2.W;
    sg.out = s == 0 ? g0.out : s == 1 ? g1.out : g2.out;
    sg.resid = s == 0 ? g0.resid : s == 1 ? g1.resid : g2.resid;
    sg.type = s == 0 ? g0.type : s == 1 ? g1.type : g2.type;
    sg.n_rows = s == 0 ? g0.n_rows : s == 1 ? g1.n_rows : g2.n_rows;
    sg.row_bytes = s == 0 ? g0.row_bytes : s == 1 ? g1.row_bytes : g2.row_bytes;
    const int wave = uni(tid_now() >> 6);
    // sync words: published slots 0, rendezvous 0, every consumer at its first step
    if (threadIdx.x < 32) reinterpret_cast<int *>(smem + lay.sync)[threadIdx.x] = threadIdx.x >= SY_DONE ? (int)threadIdx.x - SY_DONE : 0;
    __syncthreads();
    // contiguous rows per workgroup
    const int rpb = (sg.n_rows + nblk - 1) / nblk;
    int b0 = bl * rpb, b1 = b0 + rpb;
    if (b0 > sg.n_rows) b0 = sg.n_rows;
    if (b1 > sg.n_rows) b1 = sg.n_rows;
    const int n_rows_wg = b1 - b0;
    if (n_rows_wg <= 0) return;
    const bool swiglu = a.epi == EPI_SWIGLU;
    const bool pair = 2 * sg.row_bytes <= (size_t)ST_PAIR_MAX;
    const unsigned total = (unsigned)n_rows_wg * (unsigned)sg.row_bytes;      // per tensor
#ifdef MI355_STREAM_PROBE
    {
        unsigned dep = uni((int)total); asm volatile("" : "+s"(dep));
        const unsigned long long t_args = wall_clock64() + (dep & 0);
        if (g_stream_probe && (threadIdx.x & 63) == 0) {
            unsigned long long *pp = &g_stream_probe[(((size_t)(a.nck >> 2) * 256 + blockIdx.x) * ST_NW + wave) * 8];
            pp[0] = t_top; pp[7] = t_args;
        }
    }
#endif
    if (wave < ST_NL) {
        const unsigned step_bytes = (unsigned)sg.row_bytes * ((swiglu || pair) ? 2u : 1u);
        if (swiglu) { if constexpr (FUSE == 1) run_loader<FUSE, true>(a, sg, smem, lay, b0, total, step_bytes, wave); }
        else run_loader<FUSE, false>(a, sg, smem, lay, b0, total, step_bytes, wave);
        return;
    }
    const int c = wave - ST_NL;
    // forms that exist: SwiGLU pairs only with the fused RMSNorm prologue (gate/up), row pairs up to K = 8192, single rows
    // from K = 6144 (mmvq_stream_applicable agrees)
#define RUN(TY)                                                                                               \
    do {                                                                                                      \
        if (swiglu) { if constexpr (FUSE == 1) run_consumer<TY, KB, FUSE, true, true>(a, sg, smem, lay, b0, n_rows_wg, c); } \
        else if (pair) { if constexpr (KB <= 4) run_consumer<TY, KB, FUSE, false, true>(a, sg, smem, lay, b0, n_rows_wg, c); } \
        else { if constexpr (KB >= 3) run_consumer<TY, KB, FUSE, false, false>(a, sg, smem, lay, b0, n_rows_wg, c); } \
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
        if (g.row_bytes > (size_t)ST_MAX_STEP) return false;
        if (kb <= 2 && 2 * g.row_bytes > (size_t)ST_PAIR_MAX) return false;      // K <= 4096 only has the row-pair form
        if (kb >= 6 && 2 * g.row_bytes <= (size_t)ST_PAIR_MAX) return false;     // K >= 10240 only the single-row form
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
