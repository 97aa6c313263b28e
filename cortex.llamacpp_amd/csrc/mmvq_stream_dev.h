// mmvq_stream_dev.h — device side of the weight-stream mat-vec (see mmvq_stream.hip for the design notes and measurements):
// the LDS-DMA loader waves, the ring decoders and the consumer waves.  Shared by mmvq_stream.hip (one mat-vec per launch) and
// decode_engine.hip (the mat-vecs of a layer inside one persistent launch, hand-overs through tagged granules).
#pragma once
#include "mmvq_fast_dev.h"
#include <hip/hip_ext.h>

namespace mi355 {

namespace {

constexpr int ST_NL = 2;                        // loader waves (the first waves of the workgroup)
#ifndef MI355_ST_NC
#define MI355_ST_NC 8
#endif
constexpr int ST_NC = MI355_ST_NC;              // consumer waves (8: the wave count the prologue of mmvq_fast is cut for; up to 14 = a 1024-thread workgroup)
static_assert(ST_NC >= 1 && ST_NC <= 14, "consumer waves");
constexpr int ST_NW = ST_NL + ST_NC, ST_NT = ST_NW * 64;
#ifndef MI355_ST_RING
#define MI355_ST_RING 131072                    // (tools/exp_stream.hip builds a 64 KiB variant: two workgroups per CU)
#endif
constexpr int ST_RING = MI355_ST_RING;          // bytes, a power of two (ring offsets wrap with a mask)
constexpr unsigned ST_MASK = ST_RING - 1;
constexpr int ST_SI = 4;                        // DMA instructions (1 KiB each) per slot
constexpr int ST_SLOT = ST_SI * 1024;           // unit of publication; global slot g lives at ring offset g * 4 KiB mod ring,
constexpr int ST_RING_SLOTS = ST_RING / ST_SLOT;   // loader q copies the slots g = q mod 2
#ifndef MI355_STREAM_DEPTH
#define MI355_STREAM_DEPTH 12
#endif
constexpr int ST_D = MI355_STREAM_DEPTH;        // slots in flight per loader (4 .. 15 measured alike); together with the activation-plane
                                                // pieces never more than 60 of the 63 vector-memory operations a wave's counter can count
#ifndef MI355_STREAM_THIN_DEPTH
#define MI355_STREAM_THIN_DEPTH 2
#endif
#ifndef MI355_STREAM_EARLY
#define MI355_STREAM_EARLY 0
#endif
constexpr int ST_EARLY = MI355_STREAM_EARLY;    // slots a loader of a stream-bound launch requests before the consumers' own requests are queued (0 = none)
constexpr int ST_THIN_D = MI355_STREAM_THIN_DEPTH;   // slots in flight per loader while a wave of the CU gathers a hand-over (decode_engine.hip)
constexpr int ST_MAX_STEP = 16384;              // bytes one decode step may span (a row, a row pair or one row of a gate/up pair)
#ifndef MI355_ST_PAIR_MAX
#define MI355_ST_PAIR_MAX 12288
#endif
constexpr int ST_PAIR_MAX = MI355_ST_PAIR_MAX;  // rows are decoded two at a time up to this many bytes per pair

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
// tools/exp_stream.hip: per wave, 100 MHz wall-clock stamps and accumulated waits (single-op launches)
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

// ---- decoders reading a row out of the ring.  `base` = ring byte offset of the mat-vec's first slot; `x` = byte offset in
// the workgroup's run of the tensor.  One tensor: run byte x is ring byte base + x.  IL (gate/up): gate slot j and up slot
// j alternate in the ring (loader 0 streams the gate run, loader 1 the up run), so run byte x of tensor t is ring byte
// base + 4096 * (2 * (x / 4096) + t) + x % 4096; t is folded into base.  Everything is 16-B granular: a piece never
// straddles a slot or the wrap.  Same fields as Raw<TYPE>::load of mmvq_fast_dev.h.
template <bool IL> __device__ __forceinline__ const uint8_t *ring_at(const uint8_t *ring, unsigned base, unsigned x) {
    const unsigned o = IL ? base + ((x >> 12) << 13) + (x & 4095u) : base + x;
    return ring + (o & ST_MASK);
}
#define RO(x) ring_at<IL>(ring, base, (x))
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q4_K> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 144u;
    r.hdr = lds16(RO(b));
    r.q = lds16(RO(b + 16u + (unsigned)L.c * 32u));
    r.q1 = lds16(RO(b + 32u + (unsigned)L.c * 32u));
}
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q5_K> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 176u;
    r.hdr = lds16(RO(b));
    r.qh = lds16(RO(b + 16u));
    r.qh1 = lds16(RO(b + 32u));
    r.q = lds16(RO(b + 48u + (unsigned)L.c * 32u));
    r.q1 = lds16(RO(b + 64u + (unsigned)L.c * 32u));
}
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q6_K> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    r.ql = lds16(RO(row_off + (unsigned)sb * 128u + (unsigned)L.v * 16u));
    r.qh = lds16(RO(row_off + (unsigned)nb * 128u + (unsigned)sb * 64u + (unsigned)L.n * 32u + (unsigned)(L.w & 1) * 16u));
    const unsigned so = row_off + (unsigned)nb * 192u + (unsigned)sb * 16u + 8u * (unsigned)L.n + (unsigned)L.w;
    r.sc_lo = *reinterpret_cast<const int8_t *>(RO(so));
    r.sc_hi = *reinterpret_cast<const int8_t *>(RO(so + 4u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 208u + (unsigned)sb * 2u));
}
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q8_0> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned b = row_off + (unsigned)sb * 256u + (unsigned)L.v * 32u;
    r.q0 = lds16(RO(b));
    r.q1 = lds16(RO(b + 16u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 256u + ((unsigned)sb * 8u + (unsigned)L.v) * 2u));
}
// Q2_K / Q3_K device rows (planes: qs | scales | d, dmin  and  hmask | qs | scales | d) and the 32-element formats (nibbles | [qh] | scales): the fields of
// Raw<>::load in mmvq_fast_dev.h, read out of the ring
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q2_K> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    r.q = lds16(RO(row_off + (unsigned)sb * 64u + (unsigned)(L.c >> 1) * 32u + (unsigned)L.h * 16u));
    const unsigned so = row_off + (unsigned)nb * 64u + (unsigned)sb * 16u + 4u * (unsigned)L.c + (unsigned)L.h;
    r.sc_lo = *RO(so);
    r.sc_hi = *RO(so + 2u);
    r.dd = *reinterpret_cast<const uint32_t *>(RO(row_off + (unsigned)nb * 80u + (unsigned)sb * 4u));
}
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q3_K> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    r.hm = lds16(RO(row_off + (unsigned)sb * 32u + (unsigned)L.h * 16u));
    r.q = lds16(RO(row_off + (unsigned)nb * 32u + (unsigned)sb * 64u + (unsigned)(L.c >> 1) * 32u + (unsigned)L.h * 16u));
    const unsigned so = row_off + (unsigned)nb * 96u + (unsigned)sb * 12u;
    r.s0w = *reinterpret_cast<const uint32_t *>(RO(so));
    r.s1w = *reinterpret_cast<const uint32_t *>(RO(so + 4u));
    r.s2w = *reinterpret_cast<const uint32_t *>(RO(so + 8u));
    r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + (unsigned)nb * 108u + (unsigned)sb * 2u));
}
template <bool IL, int TYPE> __device__ __forceinline__ void ring_load_nib32(RawNib32<TYPE> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) {
    const unsigned blk = (unsigned)sb * 8u + (unsigned)L.v, half = (unsigned)nb * 128u, nblk = (unsigned)nb * 8u;
    r.q = lds16(RO(row_off + blk * 16u));
    if (TYPE == T_Q5_0) {
        r.qh = *reinterpret_cast<const uint32_t *>(RO(row_off + half + blk * 4u));
        r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + half + nblk * 4u + blk * 2u));
    } else {
        r.qh = 0;
        r.dh16 = *reinterpret_cast<const uint16_t *>(RO(row_off + half + blk * 2u));
    }
}
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q4_0> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) { ring_load_nib32<IL, T_Q4_0>(r, ring, base, row_off, nb, sb, L); }
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_Q5_0> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) { ring_load_nib32<IL, T_Q5_0>(r, ring, base, row_off, nb, sb, L); }
template <bool IL> __device__ __forceinline__ void ring_load(Raw<T_IQ4_NL> &r, const uint8_t *ring, unsigned base, unsigned row_off, int nb, int sb, const LaneRole &L) { ring_load_nib32<IL, T_IQ4_NL>(r, ring, base, row_off, nb, sb, L); }
#undef RO

// ---- LDS layout (bytes from smem): sync words | reduction scratch | ring | activation of the current mat-vec
constexpr int ST_OFF_SYNC = 0, ST_OFF_RED = 256, ST_OFF_RING = 512, ST_OFF_ACT = ST_OFF_RING + ST_RING;   // (64 sync words, 16 doubles of reduction scratch)
struct StLayout { int qs, d, bs, total; };
__host__ __device__ inline StLayout st_layout(int kb) {
    const int Kp = kb * 2048;
    StLayout l;
    l.qs = ST_OFF_ACT;
    l.d = l.qs + Kp;                            // one DMA piece (1 KiB) of room: nb * 4 B of scales
    l.bs = l.d + 1024;                          // Kp / 8 B of block sums (Q8_0: f32 block scales, Kp / 8 B too), in whole DMA pieces
    l.total = l.bs + ((Kp / 8 + 1023) & ~1023);
    return l;
}
// sync words (ints at smem): [q] slots published by loader q; [2] consumers whose own global requests are in the memory
// queue (the loaders start after that); [8] / [12] arrivals at the two prologue rendezvous; [24 + c] consumer c: the first
// slot it still needs
enum { SY_LANDED = 0, SY_GO = 2, SY_THIN = 3, SY_PRO1 = 8, SY_PRO2 = 12, SY_DONE = 16, SY_READY = 17, SY_ABORT = 18, SY_GATHER = 19, SY_GDONE = 20, SY_STEP = 21, SY_FRONT = 24 };
// (decode_engine.hip: [SY_THIN] != 0: a wave of this CU is gathering a hand-over, the loaders keep ST_THIN_D slots in flight; [SY_DONE] consumers that have
// finished their steps, counted over the launch; [SY_READY] hand-overs gathered into LDS so far; the rendezvous words count over the launch too)

// every wait on another wave is bounded: the waves of a workgroup are always co-resident, so a healthy wait ends within microseconds; if a protocol bug (or a
// debugger, or time-slicing) ever broke that, the launch still ends instead of hanging the device - and says so: a wait that gives up raises the sticky error
// word (pinned host memory, set once per process with stream_set_error_word; Context::decode checks it where it synchronises anyway and fails the step), so
// wrong numbers are never returned silently.  After every wait an empty asm with a memory clobber keeps the compiler from hoisting the reads of the data the
// wait guards (ring slots, DMA'd planes) above it: the acquire side of the hand-over (the release side has the same in front of its store).
#ifndef MI355_STREAM_SPIN_LIMIT
#define MI355_STREAM_SPIN_LIMIT (1 << 21)
#endif
constexpr int ST_SPIN_LIMIT = MI355_STREAM_SPIN_LIMIT;
__device__ unsigned *g_st_err_word = nullptr;       // (one copy per translation unit that includes this header; each has its setter)
__device__ __attribute__((noinline)) void st_timeout(unsigned code) {
    unsigned *w = g_st_err_word;
    if (w) __hip_atomic_fetch_or(w, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
enum { ST_ERR_SPIN = 1u, ST_ERR_LOADER = 2u, ST_ERR_DRAIN = 4u, ST_ERR_GATHER = 8u, ST_ERR_ENGINE_WAIT = 16u };
#define ST_SPIN_WHILE(cond, sleep_arg) do { int spins_ = 0; while (cond) { if (++spins_ >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_SPIN); break; } __builtin_amdgcn_s_sleep(sleep_arg); } \
                                            asm volatile("" ::: "memory"); } while (0)
__device__ __forceinline__ void consumers_rendezvous(int *word, int lane, int round = 1) {   // round: the word counts over the launch (engine: several mat-vecs)
    asm volatile("" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ST_SPIN_WHILE(ld_sync(word) < ST_NC * round, 1);
}

// What the device code needs of one MMVQArgs, read in ONE batch of scalar loads: the kernel argument segment is cold at
// every launch (~0.7 us per miss, tools/exp_stream.hip timeline) and hipcc loads a field where it is first used, so
// `a.seg[s]` behind the segment choice and a.nx behind the role choice were two more serial misses.
struct StOp {
    int K, epi, nck, type, n_rows, b0, n_rows_wg;
    float neps;
    const int8_t *aq; const float *ad; const int16_t *abs;
    const float *nx, *nw;
    const uint8_t *W, *W1;        // this workgroup's segment; SwiGLU: W1 = the up tensor
    float *out; const float *resid;
    float *out2;                  // a second copy of every result (segment 0 of a one-segment launch: the logits row into pinned host memory), or nullptr
    unsigned row_bytes, total;    // total = bytes of the workgroup's run of rows, per tensor
    bool swiglu, pair;
    int ns_raw, ns_pad;           // ring slots of the workgroup's share (gate / up slots alternate), and padded to an even count
};
// MEM: the MMVQArgs sits in device memory (decode_engine.hip) instead of the kernel argument segment: hipcc reads it with vector loads there, and what
// feeds a scalar operand (the DMA's M0, a segment choice) has to be made wave-uniform explicitly
template <bool MEM, class T> __device__ __forceinline__ T st_uni(T v) {
    if constexpr (!MEM) return v;
    else if constexpr (sizeof(T) == 4) { int b = __builtin_bit_cast(int, v); b = __builtin_amdgcn_readfirstlane(b); return __builtin_bit_cast(T, b); }
    else {
        static_assert(sizeof(T) == 8, "st_uni");
        unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
        b = ((unsigned long long)hi << 32) | lo;
        return __builtin_bit_cast(T, b);
    }
}
// MEM (the description sits in LDS, decode_engine.hip): every lane reads two words of it in ONE pair of LDS reads and the fields are picked out of those
// two registers with v_readlane - a field-by-field read (ds_read -> wait -> v_readfirstlane, forty times) cost 0.8 us per mat-vec on the critical path of
// every hand-over
struct StDescRegs { unsigned r0, r1; };
static_assert(sizeof(MMVQArgs) <= 512, "MMVQArgs no longer fits two words per lane");
__device__ __forceinline__ StDescRegs st_desc_load(const MMVQArgs &ka) {
    const unsigned *p = reinterpret_cast<const unsigned *>(&ka);
    const int lane = tid_now() & 63;
    constexpr int NW = (int)(sizeof(MMVQArgs) / 4);
    StDescRegs r;
    r.r0 = p[lane < NW ? lane : NW - 1];
    r.r1 = p[64 + lane < NW ? 64 + lane : NW - 1];
    return r;
}
template <class T> __device__ __forceinline__ T st_desc_get(const StDescRegs &r, size_t byte_off) {
    const int d = (int)(byte_off >> 2);
    auto word = [&](int i) { return (unsigned)__builtin_amdgcn_readlane((int)(i < 64 ? r.r0 : r.r1), i & 63); };
    if constexpr (sizeof(T) == 4) { const unsigned w = word(d); return __builtin_bit_cast(T, w); }
    else {
        static_assert(sizeof(T) == 8, "st_desc_get");
        const unsigned long long w = (unsigned long long)word(d) | ((unsigned long long)word(d + 1) << 32);
        return __builtin_bit_cast(T, w);
    }
}
// MOE: a mixture-of-experts step's launch (the token's n_sel selected experts share it, kernels.h MMVQArgs::n_sel); the dense kernels are built without it - its
// seven extra argument words and the dependent read of the expert index cost every launch 0.1 - 0.25 us when they were unconditional (same-box A/B: 577 -> 570 tok/s)
template <bool MEM = false, bool MOE = false, class KA = MMVQArgs>
__device__ __forceinline__ void op_setup(const KA &ka, StOp &o) {
    StDescRegs dr{0u, 0u};
    if constexpr (MEM) dr = st_desc_load(*reinterpret_cast<const MMVQArgs *>(&ka));
#define U(x) (MEM ? st_desc_get<decltype(st_uni<false>(x))>(dr, (size_t)((uintptr_t)&(x) - (uintptr_t)&ka)) : (x))
    const int n_seg = U(ka.n_seg), sb0 = U(ka.seg_block0[0]), sb1 = U(ka.seg_block0[1]), sb2 = U(ka.seg_block0[2]), sb3 = U(ka.seg_block0[3]);
    const int K = U(ka.K), epi = U(ka.epi), nck = U(ka.nck);
    const float neps = U(ka.neps);
    const int8_t *aq = U(ka.aq); const float *ad = U(ka.ad); const int16_t *abs = U(ka.abs); const float *nx = U(ka.nx), *nw = U(ka.nw);
    const uint8_t *w0 = U(ka.seg[0].W), *w1 = U(ka.seg[1].W), *w2 = U(ka.seg[2].W);
    float *o0 = U(ka.seg[0].out), *o1 = U(ka.seg[1].out), *o2 = U(ka.seg[2].out), *oh = U(ka.out_host);
    const float *r0 = U(ka.seg[0].resid), *r1 = U(ka.seg[1].resid), *r2 = U(ka.seg[2].resid);
    const int t0 = U(ka.seg[0].type), t1 = U(ka.seg[1].type), t2 = U(ka.seg[2].type);
    const int n0 = U(ka.seg[0].n_rows), n1 = U(ka.seg[1].n_rows), n2 = U(ka.seg[2].n_rows);
    const unsigned rb0 = (unsigned)U(ka.seg[0].row_bytes), rb1 = (unsigned)U(ka.seg[1].row_bytes), rb2 = (unsigned)U(ka.seg[2].row_bytes);
    // mixture-of-experts step: the token's n_sel selected experts share the launch (kernels.h MMVQArgs::n_sel); segment 0 (and the up tensor of a SwiGLU pair) carry them
    int n_sel = 0, sel_os = 0, sel_ns = 0;
    const int32_t *es0 = nullptr, *es1 = nullptr;
    size_t est0 = 0, est1 = 0;
    if constexpr (MOE) {
        n_sel = U(ka.n_sel); sel_os = U(ka.sel_out_stride); sel_ns = U(ka.sel_nx_stride);
        es0 = U(ka.seg[0].expert_sel); es1 = U(ka.seg[1].expert_sel);
        est0 = U(ka.seg[0].expert_stride); est1 = U(ka.seg[1].expert_stride);
    }
#undef U
    // (pinned: without a use here hipcc sinks each load to its first use again)
#define PIN(x) asm volatile("" :: "s"(x))
    PIN(n_seg); PIN(sb0); PIN(sb1); PIN(sb2); PIN(sb3); PIN(K); PIN(epi); PIN(nck); PIN(neps);
    PIN(aq); PIN(ad); PIN(abs); PIN(nx); PIN(nw);
    PIN(w0); PIN(w1); PIN(w2); PIN(o0); PIN(o1); PIN(o2); PIN(oh); PIN(r0); PIN(r1); PIN(r2);
    PIN(t0); PIN(t1); PIN(t2); PIN(n0); PIN(n1); PIN(n2); PIN(rb0); PIN(rb1); PIN(rb2);
    if constexpr (MOE) { PIN(n_sel); PIN(sel_os); PIN(sel_ns); PIN(es0); PIN(es1); PIN(est0); PIN(est1); }
#undef PIN
    int s = 0;
    if (n_seg > 1 && (int)blockIdx.x >= sb1) s = 1;
    if (n_seg > 2 && (int)blockIdx.x >= sb2) s = 2;
    const int lo = s == 0 ? sb0 : s == 1 ? sb1 : sb2, hi = s == 0 ? sb1 : s == 1 ? sb2 : sb3;
    int nblk = hi - lo, bl = (int)blockIdx.x - lo;
    int sel_j = 0;
    if (MOE && n_sel > 1 && nblk >= n_sel && bl >= 0 && bl < nblk) {   // an even share of the segment's workgroups per selected expert (the last one takes the remainder)
        const int per = nblk / n_sel;
        sel_j = bl / per < n_sel ? bl / per : n_sel - 1;
        bl -= sel_j * per;
        nblk = sel_j == n_sel - 1 ? nblk - sel_j * per : per;
    }
    o.K = K; o.epi = epi; o.nck = nck; o.neps = neps; o.aq = aq; o.ad = ad; o.abs = abs; o.nw = nw;
    o.W = s == 0 ? w0 : s == 1 ? w1 : w2; o.W1 = w1;
    if (MOE && es0 && s == 0) {                                        // the expert's index is read on the device (written by the router launch before this one)
        o.W += (size_t)es0[sel_j] * est0;
        if (es1) o.W1 += (size_t)es1[sel_j] * est1;
    }
    o.nx = nx; o.out = s == 0 ? o0 : s == 1 ? o1 : o2;
    o.out2 = n_seg == 1 ? oh : nullptr;
    if constexpr (MOE) { o.nx = nx + (size_t)sel_j * (size_t)sel_ns; o.out += (size_t)sel_j * (size_t)sel_os; }
    o.resid = s == 0 ? r0 : s == 1 ? r1 : r2;
    o.type = s == 0 ? t0 : s == 1 ? t1 : t2;
    o.n_rows = s == 0 ? n0 : s == 1 ? n1 : n2;
    o.row_bytes = s == 0 ? rb0 : s == 1 ? rb1 : rb2;
    // contiguous rows per workgroup (workgroups past the last row of the segment have none)
    int b0 = 0, b1 = 0;
    if (nblk > 0 && bl >= 0 && (int)blockIdx.x < sb3) {
        const int rpb = (o.n_rows + nblk - 1) / nblk;
        b0 = bl * rpb; b1 = b0 + rpb;
        if (b0 > o.n_rows) b0 = o.n_rows;
        if (b1 > o.n_rows) b1 = o.n_rows;
    }
    o.b0 = b0; o.n_rows_wg = b1 - b0;
    o.swiglu = epi == EPI_SWIGLU;
    o.pair = 2 * o.row_bytes <= (unsigned)ST_PAIR_MAX;
    o.total = (unsigned)o.n_rows_wg * o.row_bytes;
    const int per_tensor = (int)((o.total + ST_SLOT - 1) / ST_SLOT);
    o.ns_raw = o.swiglu ? 2 * per_tensor : per_tensor;
    o.ns_pad = (o.ns_raw + 1) & ~1;
}

// RMSNorm * w and / or the Q8_K (Q8_0) quantisation of the token by the 8 consumer waves, result in LDS: stage_finish of
// mmvq_fast_dev.h (consumer c owns 256-blocks c, c + 8, ..; sum of squares in double, fixed order), with the workgroup
// barrier replaced by a rendezvous of the consumers (the loaders never join: they sit in the memory queue).
// decode_engine.hip: where a mat-vec's activation comes from and where its results go when it is one of several inside ONE launch
struct EngIO {
    const float *xf = nullptr;            // LDS: the token's f32 vector, gathered from the producers' granules (source of the RMSNorm / quantisation)
    int ready_round = 0;                  // [SY_READY] >= this: the hand-over this mat-vec consumes is in LDS
    int round = 1;                        // rendezvous round of this mat-vec's prologue (the words count over the launch)
    unsigned long long *gran = nullptr;   // results as tagged granules {tag, f32 bits}, one 8-byte device-coherent store each (nullptr: plain stores only)
    unsigned tag = 0;
    float *plain = nullptr;               // with gran: the results also as plain stores (read by a LATER launch)
    const float *rs = nullptr;            // LDS: residual of this workgroup's rows, rs[row - b0] (nullptr: a.resid in global memory)
    unsigned long long *probe = nullptr;  // diagnosis: [0] activation ready (wall clock), [1] time spent waiting for ring slots, [2] time spent decoding
    int step_base = 0;                    // value of the step ticket counter [SY_STEP] at which this mat-vec's step 0 sits (the counter runs over the launch)
};
__device__ __forceinline__ void st_store_granule(unsigned long long *g, unsigned tag, unsigned value) {
    __hip_atomic_store(g, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Round 5 (the stream-bound roles, mmvq_stream.hip stream_body_fast): the consumers' activation / norm-weight requests issued in the kernel's first instructions
// from PRELOADED kernel arguments (scalars that arrive in SGPRs with the wave), before the argument struct has been read
template <int KB> struct EarlyAct {
    static constexpr int NJW = (KB * 8 + ST_NC - 1) / ST_NC;
    f32x4_t x[NJW], w[NJW];
};
template <int KB, int FUSE>
__device__ __forceinline__ void early_issue(const float *nx, const float *nw, int K, int c, int lane, EarlyAct<KB> &ea) {
    const int nbt = K >> 8;
#pragma unroll
    for (int i = 0; i < EarlyAct<KB>::NJW; i++) {
        const int b = c + ST_NC * i;
        const int bc = b < nbt ? b : (nbt > 0 ? nbt - 1 : 0);  // clamped as in consumer_prologue
        ea.x[i] = *reinterpret_cast<const f32x4_t *>(nx + bc * 256 + lane * 4);
        if (FUSE == 1) ea.w[i] = *reinterpret_cast<const f32x4_t *>(nw + bc * 256 + lane * 4);
    }
}
// XLDS (engine): the f32 vector is read from LDS (io.xf) once [SY_READY] says it has been gathered; the norm weights are requested before that wait
template <int KB, int FUSE, bool Q80, bool XLDS = false, bool EARLY = false>
__device__ __forceinline__ void consumer_prologue(const StOp &a, uint8_t *smem, const StLayout &lay, int c, int lane, const EngIO &io = EngIO(), const EarlyAct<KB> *ea = nullptr) {
    constexpr int NJW = (KB * 8 + ST_NC - 1) / ST_NC;
    const int nbt = a.K >> 8;
    int8_t *qs = reinterpret_cast<int8_t *>(smem + lay.qs);
    float *d = reinterpret_cast<float *>(smem + lay.d);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + lay.bs);
    double *red = reinterpret_cast<double *>(smem + ST_OFF_RED);
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    f32x4_t rxv[NJW], rwv[NJW];
#pragma unroll
    for (int i = 0; i < NJW; i++) {
        const int b = c + ST_NC * i;
        const int bc = b < nbt ? b : nbt - 1;                  // clamped: always a valid address, result unused when b is out of range
        if (EARLY) { rxv[i] = ea->x[i]; if (FUSE == 1) rwv[i] = ea->w[i]; continue; }     // requested at the top of the kernel (early_issue)
        if (!XLDS) rxv[i] = *reinterpret_cast<const f32x4_t *>(a.nx + bc * 256 + lane * 4);
        if (FUSE == 1) rwv[i] = *reinterpret_cast<const f32x4_t *>(a.nw + bc * 256 + lane * 4);
    }
    asm volatile("" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (XLDS) {
        ST_SPIN_WHILE(ld_sync(sy + SY_READY) < io.ready_round && ld_sync(sy + SY_ABORT) == 0, 1);
#pragma unroll
        for (int i = 0; i < NJW; i++) {
            const int b = c + ST_NC * i;
            const int bc = b < nbt ? b : nbt - 1;
            rxv[i] = *reinterpret_cast<const f32x4_t *>(io.xf + bc * 256 + lane * 4);
        }
    }
    float scale = 1.0f;
    if (FUSE == 1) {
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < NJW; i++) {
            const f32x4_t x = rxv[i];
            double t = 0.0;
            t += (double)(x.x * x.x); t += (double)(x.y * x.y); t += (double)(x.z * x.z); t += (double)(x.w * x.w);
            if (c + ST_NC * i < nbt) sum += t;
        }
        sum = wave_sum(sum);
        if (lane == 0) red[c] = sum;
        consumers_rendezvous(sy + SY_PRO1, lane, io.round);
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < ST_NC; w++) tot += red[w];
        // tot / K: a power-of-two K divides exactly by a multiplication - the same bits as the division, without the f64 divide on this serial path
        const double dk = (double)a.K;
        const float mean = (a.K & (a.K - 1)) == 0 ? (float)(tot * (1.0 / dk)) : (float)(tot / dk);
        scale = 1.0f / sqrtf(mean + a.neps);
    }
    if (Q80) {
#pragma unroll
        for (int i = 0; i < NJW; i++) {
            const int b = c + ST_NC * i;
            if (b >= nbt) continue;                            // wave-uniform
            const int e0 = b * 256 + lane * 4;
            f32x4_t x = rxv[i];
            if (FUSE == 1) {
                const f32x4_t ww = rwv[i];
                x.x = (x.x * scale) * ww.x; x.y = (x.y * scale) * ww.y; x.z = (x.z * scale) * ww.z; x.w = (x.w * scale) * ww.w;
            }
            const float vv[4] = {x.x, x.y, x.z, x.w};
            uint32_t packed; float dd;
            wave_quant_q80(vv, packed, dd);
            *reinterpret_cast<uint32_t *>(qs + e0) = packed;
            if ((lane & 7) == 0) reinterpret_cast<float *>(bs)[b * 8 + (lane >> 3)] = h2f(f2h(dd));
        }
    } else {
        // the wave's blocks side by side in ONE straight line (a clamped block where the last one does not exist; only its stores are skipped): a
        // block's quantisation is a chain of wave-level steps (max -> first lane holding it -> its value -> 1 / scale -> codes -> sums) that waits on
        // itself, so two or more of them interleave (decode_engine.hip's probe: 0.84 us for two blocks, most of it latency)
        uint32_t packed[NJW]; int bsum[NJW]; float dq[NJW];
        float vv[NJW][4];
#pragma unroll
        for (int i = 0; i < NJW; i++) {
            f32x4_t x = rxv[i];
            if (FUSE == 1) {
                const f32x4_t ww = rwv[i];
                x.x = (x.x * scale) * ww.x; x.y = (x.y * scale) * ww.y; x.z = (x.z * scale) * ww.z; x.w = (x.w * scale) * ww.w;
            }
            vv[i][0] = x.x; vv[i][1] = x.y; vv[i][2] = x.z; vv[i][3] = x.w;
        }
        wave_quant_q8k_batch<NJW>(vv, lane, packed, bsum, dq);      // (the blocks' two divisions each done once, lane-parallel: quant_dev.h)
#pragma unroll
        for (int i = 0; i < NJW; i++) {
            const int b = c + ST_NC * i;
            if (b < nbt) {
                *reinterpret_cast<uint32_t *>(qs + b * 256 + lane * 4) = packed[i];
                if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum[i];
                if (lane == 0) d[b] = dq[i];
            }
        }
    }
    consumers_rendezvous(sy + SY_PRO2, lane, io.round);
}

// ---- a loader wave (q = 0, 1).  The workgroup's rows of a mat-vec are one contiguous run of `total` bytes per tensor, cut
// into 4 KiB slots; the mat-vec's slots follow the previous mat-vec's in the ring (g0 = its first global slot, even).
//   one tensor : local slot k = run bytes [4096 k, ..); loader q copies the slots k = q mod 2;
//   SwiGLU     : local slot 2 i = gate slot i, 2 i + 1 = up slot i: loader 0 streams the gate run, loader 1 the up run.
// Every slot is four DMA instructions (the tail re-reads the run's last 16 B, the padding slot of an odd count likewise),
// so "at most 60 outstanding" always means "everything but my last 15 slots has landed".
// Nothing here blocks: what has landed is read off the wave's own vector-memory counter (the one s_waitcnt vmcnt waits on,
// HW_REG_IB_STS) after every step, so a slot is published as soon as it is in LDS — with counted waits a slot was only
// published once 15 more had been issued (5 us at the streaming rate), and a loader blocked on ring space had to drain
// everything (vmcnt(0)) before it could publish anything, one slot per round trip.
struct LoaderState {
    int issued, published;        // slots of THIS loader, over the whole launch (its i-th slot is global slot 2 i + q)
    int pre;                      // DMA instructions issued before the first slot (activation planes)
    unsigned free_until;          // global slots below this index may be in the ring
    int idle_polls;
    int w_depth = 0, w_space = 0; // diagnosis: polls spent waiting for landings (in-flight limit) / for ring space, since last cleared
};
__device__ __forceinline__ int vm_outstanding() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_IB_STS)" : "=s"(x) :: "memory");
    return (int)((x & 0xfu) | ((x >> 18) & 0x30u));              // VM_CNT = {bits 23:22, bits 3:0}
}
__device__ __forceinline__ void loader_publish(LoaderState &st, int *sy, int q) {
    const int done = st.pre + ST_SI * st.issued - vm_outstanding();   // DMA instructions complete (they complete in order)
    const int landed = done <= st.pre ? 0 : (done - st.pre) / ST_SI;
    if (landed > st.published) { st.published = landed; st_sync(sy + SY_LANDED + q, landed); }
}
// early_slots > 0 (a launch whose share of the weights exceeds the ring: it is bound by its stream, not by its prologue): this loader requests that many
// slots BEFORE the consumers' own requests are queued ([SY_GO]) - a head start of the stream worth what it moves, for a bounded delay of those requests -
// and waits for [SY_GO] only then
template <bool THIN = false>
__device__ __forceinline__ void loader_op(LoaderState &st, const StOp &a, uint8_t *smem, int q, unsigned g0, int lane, int early_slots = 0) {
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    bool go = early_slots <= 0;
    if (a.ns_pad == 0) return;
    const unsigned ring_lds = lds_addr(smem + ST_OFF_RING);
    const uint8_t *W = (a.swiglu && q == 1 ? a.W1 : a.W) + (size_t)a.b0 * a.row_bytes;
    const unsigned total = a.total, last = total - 16u;
    const unsigned run_stride = a.swiglu ? ST_SLOT : 2u * ST_SLOT;              // run bytes between two slots of this loader
    unsigned run_off = a.swiglu ? 0u : (unsigned)q * ST_SLOT;                    // run offset of this loader's next slot
    const uint8_t *pl = W + run_off + (size_t)lane * 16;                         // this lane's 16 B of it
    const int n_mine = a.ns_pad / 2;                                             // slots of this loader in the mat-vec
    for (int i = 0; i < n_mine;) {
        if (!go && i >= early_slots) { ST_SPIN_WHILE(ld_sync(sy + SY_GO) < ST_NC, 0); go = true; }
        loader_publish(st, sy, q);
        const unsigned g = g0 + 2u * (unsigned)i + (unsigned)q;                  // global slot
        int depth = ST_D * ST_SI + st.pre <= 60 ? ST_D : (60 - st.pre) / ST_SI;   // (the counter read by vm_outstanding() saturates at 63)
        if (THIN && uni(ld_sync(sy + SY_THIN)) != 0) depth = ST_THIN_D;          // a consumer of this CU gathers a hand-over: stay out of its way
        bool wait = st.issued - st.published >= depth;                           // in flight: at most `depth` slots
        if (wait) st.w_depth++;
        if (!wait && g + 1u > st.free_until) {                                   // ring space: the slowest consumer's front + the ring
            int f = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < ST_NC; c++) { const int x = uni(ld_sync(sy + SY_FRONT + c)); f = x < f ? x : f; }
            st.free_until = (unsigned)f + ST_RING_SLOTS;
            wait = g + 1u > st.free_until;
            if (wait) st.w_space++;
        }
        if (wait) {
            if (++st.idle_polls > ST_SPIN_LIMIT) { st_timeout(ST_ERR_LOADER); return; }   // (never on a healthy run)
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        st.idle_polls = 0;
        const unsigned dst = ring_lds + ((g * ST_SLOT) & ST_MASK);
        if (run_off + ST_SLOT <= total) dma_slot(pl, dst);
        else {
#pragma unroll
            for (int p = 0; p < ST_SI; p++) {
                const unsigned off = run_off + (unsigned)lane * 16u + p * 1024u;  // past the end: every such lane re-reads the run's last 16 B
                dma16<true>(W + (off < total ? off : last), dst + p * 1024);
            }
        }
        pl += run_stride; run_off += run_stride;
        i++;
        st.issued++;
    }
}
__device__ __forceinline__ void loader_drain(LoaderState &st, uint8_t *smem, int q) {
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    int polls = 0;
    while (st.published < st.issued) {
        if (++polls >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_DRAIN); break; }
        loader_publish(st, sy, q); __builtin_amdgcn_s_sleep(1);
    }
}
// the quantised activation planes of the token (pre-quantised by the attention's merge): pieces of 1 KiB by DMA, ahead of
// the first weight slot of the launch: qs [K], then d [nb * 4], then bs [K / 8]
__device__ __forceinline__ int loader_planes(const StOp &a, uint8_t *smem, const StLayout &lay, int lane) {   // returns the DMA instructions issued
    const int nb = a.K >> 8;
    const int nq = (a.K + 1023) >> 10, nbs = ((a.K >> 3) + 1023) >> 10;      // (K a multiple of 256: the last piece of the codes may be short)
    for (int c = 0; c < nq + 1 + nbs; c++) {
        const uint8_t *src; int size; unsigned dst;
        if (c < nq) { src = reinterpret_cast<const uint8_t *>(a.aq) + c * 1024; size = a.K - c * 1024; if (size > 1024) size = 1024; dst = lay.qs + c * 1024; }
        else if (c == nq) { src = reinterpret_cast<const uint8_t *>(a.ad); size = nb * 4; dst = lay.d; }
        else { const int i = c - nq - 1; src = reinterpret_cast<const uint8_t *>(a.abs) + i * 1024; size = (a.K >> 3) - i * 1024; if (size > 1024) size = 1024; dst = lay.bs + i * 1024; }
        const int o = lane * 16 < size ? lane * 16 : size - 16;
        dma16<false>(src + o, lds_addr(smem + dst));
    }
    return nq + 1 + nbs;
}

// ---- a consumer wave, one mat-vec.  Step s = a row pair (UO = 2 outputs), one row (big K) or gate row s + up row s (one
// output).  Steps are handed out FIRST COME, in stream order, through a ticket counter in LDS: the eight consumers of a workgroup do not run at one
// speed - two of them share each SIMD and the older wave wins the issue slots (measured: the same seven gate|up steps take 5.0 us on consumer 0 and 8.1
// us on consumer 7, tools/engine_probe.py), so with the static deal (consumer c: steps c, c + 8, ..) the mat-vec ended when the slowest wave did, 3 us
// after the fastest had gone idle.  Which wave decodes a row does not change a bit of its result.  g0 = the first ring slot of the mat-vec (0: one mat-vec per launch).
// ENG (decode_engine.hip): 0 = one mat-vec per launch; 1 = the activation arrives through LDS as io says (FUSE 1: f32 vector -> RMSNorm -> Q8_K by the
// consumers; FUSE 2: f32 vector -> Q8_K; FUSE 3: the Q8_K planes are already in LDS at `lay`), 2 = FUSE 0 as in a launch of its own (planes by DMA); with
// ENG != 0 the results go where io says.  The arithmetic is the same in every form.
template <int TYPE, int KB, int FUSE, bool SWIGLU, bool PAIR, int ENG = 0, bool EARLY = false>
__device__ __forceinline__ void consumer_op(const StOp &a, uint8_t *smem, int c, unsigned g0, const StLayout &lay, const EngIO &io = EngIO(), const EarlyAct<KB> *ea = nullptr) {
    using R = Raw<TYPE>;
    constexpr int NP = role_passes<TYPE, KB>(), SBP = role_sbp<TYPE>();   // passes over a row and super-blocks per pass of this type's lane role
    constexpr bool ACT_REGS = NP <= 2;
    constexpr int STEP = (SWIGLU || PAIR) ? 2 : 1;                // rows decoded together
    constexpr int UO = (PAIR && !SWIGLU) ? 2 : 1;                 // outputs per step
    const int lane = tid_now() & 63;
    const int wave = c + ST_NL; (void)wave;
    const LaneRole L = make_role<TYPE>(lane);
    const int nb = a.K >> 8;
    const unsigned rb = a.row_bytes;
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    const uint8_t *ring = smem + ST_OFF_RING;
    const unsigned base0 = (g0 * ST_SLOT) & ST_MASK, base1 = SWIGLU ? (base0 + ST_SLOT) & ST_MASK : base0;
    const int n_rows_wg = a.n_rows_wg, b0 = a.b0;
    const int n_steps = (n_rows_wg + UO - 1) / UO;
    // the first global slot step s needs (what this consumer frees the ring up to)
    auto front_of = [&](int s) { return (int)(g0 + (SWIGLU ? 2u * (((unsigned)s * rb) >> 12) : ((unsigned)s * (unsigned)STEP * rb) >> 12)); };
    if (lane == 0) st_sync(sy + SY_FRONT + c, (int)g0);           // any step may become this wave's: the mat-vec's first slot stays until it holds a ticket
    // residual of the workgroup's rows, requested now: lane i holds row b0 + i (the rows a wave will decode are not known yet); more than 64 rows
    // per workgroup: read when the results are stored
    const bool rsd_regs = !SWIGLU && a.epi == EPI_ADD && !(ENG != 0 && io.rs) && n_rows_wg <= 64;
    float rsd = 0.0f;
    if (rsd_regs && lane < n_rows_wg) rsd = a.resid[b0 + lane];

    // ---- activation into LDS (fused modes: by the consumers themselves; planes: DMA'd by loader 0 ahead of its first slot)
    if (FUSE == 1 || FUSE == 2) consumer_prologue<KB, FUSE, act_is_q80(TYPE), ENG == 1, EARLY>(a, smem, lay, c, lane, io, ea);
    else if (FUSE == 3) {
        ST_SPIN_WHILE(ld_sync(sy + SY_READY) < io.ready_round && ld_sync(sy + SY_ABORT) == 0, 1);
    } else if (ENG == 0) {
        // ready-made planes (codes [K] | scales [K / 256] | block sums [K / 16]): the eight consumer waves fetch them themselves, 16 bytes per lane and
        // request, all requests in flight before the loaders are let go.  (Round 4: as 17 LDS-DMA pieces in front of loader 0's first weight slot - and
        // counted as landed with that slot - they held that loader back by 1.8 us and were usable 5.4 us into an ffn_down-sized launch, no sooner than
        // the quantising prologue's; tools/exp_stream.hip time line.)
        const int tix = c * 64 + lane, nq16 = a.K >> 4, nbs16 = a.K >> 7, nb4 = a.K >> 8;
        constexpr int NQ = (KB * 2048 / 16 + ST_NC * 64 - 1) / (ST_NC * 64);
        u32x4_t cq[NQ], cb = {0u, 0u, 0u, 0u};
        float cd = 0.0f;
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            int j = tix + i * ST_NC * 64;
            if (j >= nq16) j = nq16 - 1;                                   // clamped: a valid address, stored only where it belongs
            cq[i] = *reinterpret_cast<const u32x4_t *>(a.aq + (size_t)j * 16);
        }
        cb = *reinterpret_cast<const u32x4_t *>(reinterpret_cast<const uint8_t *>(a.abs) + (size_t)(tix < nbs16 ? tix : nbs16 - 1) * 16);
        cd = a.ad[tix < nb4 ? tix : nb4 - 1];
        asm volatile("" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            const int j = tix + i * ST_NC * 64;
            if (j < nq16) *reinterpret_cast<u32x4_t *>(smem + lay.qs + (size_t)j * 16) = cq[i];
        }
        if (tix < nbs16) *reinterpret_cast<u32x4_t *>(smem + lay.bs + (size_t)tix * 16) = cb;
        if (tix < nb4) reinterpret_cast<float *>(smem + lay.d)[tix] = cd;
        consumers_rendezvous(sy + SY_PRO2, lane, io.round);
    } else {
        asm volatile("" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ST_SPIN_WHILE(ld_sync(sy + SY_LANDED) < 1, 1);
    }
    ST_STAMP(2);
    const bool eprobe = ENG != 0 && io.probe != nullptr;
    unsigned long long ep_t0 = 0, ep_w = 0, ep_d = 0;
    if (eprobe) { ep_t0 = wall_clock64(); if (lane == 0) io.probe[0] = ep_t0; }
    ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
    ActSlice S0, S1;
    if (ACT_REGS) {
        S0 = read_slice_t<TYPE>(AL, L.sbl, nb, L);
        if (NP > 1) S1 = read_slice_t<TYPE>(AL, SBP + L.sbl, nb, L);
    }

    float res = 0.0f;                                             // lane i: output i of this wave (64 per flush) ...
    int rrow = b0;                                                // ... and the row it belongs to
    int n_done = 0, flushed = 0;
    auto flush = [&](int upto) {                                  // outputs [flushed, upto) are in lanes 0 ..
        const int cnt = upto - flushed;
        const int row = rrow;
        float rv = 0.0f;
        if (rsd_regs) rv = __int_as_float(__builtin_amdgcn_ds_bpermute((row - b0) << 2, __float_as_int(rsd)));   // (every lane takes part)
        if (lane < cnt) {
            float v = res;
            if (!SWIGLU && a.epi == EPI_ADD) v = (ENG != 0 && io.rs ? io.rs[row - b0] : rsd_regs ? rv : a.resid[row]) + v;
            if (ENG != 0 && io.gran) {
                st_store_granule(io.gran + row, io.tag, __float_as_uint(v));
                if (io.plain) io.plain[row] = v;
            } else {
                a.out[row] = v;
                if (ENG == 0 && a.out2) a.out2[row] = v;
            }
        }
        flushed = upto;
    };
    ST_ACC_DECL;
    ST_T0();
#ifndef MI355_ST_TICKETS
#define MI355_ST_TICKETS 0       // 1: the one-mat-vec-per-launch form takes its steps first-come too (the engine always does)
#endif
    int s_static = c;
    auto claim = [&]() {                                          // the next step of the mat-vec nobody has taken (>= n_steps: none left)
        if (ENG == 0 && !MI355_ST_TICKETS) { const int t = s_static; s_static += ST_NC; return t; }   // a launch: consumer c takes steps c, c + 8, ...
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(sy + SY_STEP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return uni(t) - io.step_base;
    };
    for (int s = claim(); s < n_steps; s = claim()) {
        if (lane == 0) st_sync(sy + SY_FRONT + c, front_of(s));
        const bool two = STEP == 2 && (SWIGLU || s * 2 + 1 < n_rows_wg);
        // run offsets of the step's rows and the slots that must have landed
        const unsigned off0 = SWIGLU ? (unsigned)s * rb : (unsigned)s * (unsigned)STEP * rb;
        const unsigned off1 = SWIGLU ? off0 : (two ? off0 + rb : off0);
        const unsigned end = SWIGLU ? off0 + rb : off0 + (two ? 2u : 1u) * rb;
        const unsigned n = (end + ST_SLOT - 1) / ST_SLOT;         // slots [0, n) of the run
        const unsigned gn = g0 + (SWIGLU ? 2u * n : n);           // global slots below gn
        const int need0 = (int)((gn + 1u) >> 1), need1 = (int)(gn >> 1);
        ST_SPIN_WHILE(ld_sync(sy + SY_LANDED) < need0 || ld_sync(sy + SY_LANDED + 1) < need1, 1);
        ST_ACC(st_acc_w);
        if (eprobe) { const unsigned long long t_ = wall_clock64(); ep_w += t_ - ep_t0; ep_t0 = t_; }
        float acc0 = 0.0f, acc1 = 0.0f;
        // K <= 4096: both passes in flight at once; longer rows: two passes at a time (fully unrolled, the compiler hoists
        // every pass's ring reads to the top: 246+ registers at 7 passes)
#pragma unroll NP <= 2 ? NP : 2
        for (int p = 0; p < NP; p++) {
            const ActSlice sl = ACT_REGS ? (p == 0 ? S0 : S1) : read_slice_t<TYPE>(AL, p * SBP + L.sbl, nb, L);
            int sb = p * SBP + L.sbl;
            if (sb >= nb) sb = nb - 1;                            // tail of a partial last pass: any valid block, its slice scale is zero
            R w0, w1;
            ring_load<SWIGLU>(w0, ring, base0, off0, nb, sb, L);
            if (STEP == 2) ring_load<SWIGLU>(w1, ring, base1, off1, nb, sb, L);
            acc0 += w0.dot(sl, L);
            if (STEP == 2) acc1 += w1.dot(sl, L);
        }
        const float v0 = wave_sum(acc0);
        const float v1 = STEP == 2 ? wave_sum(acc1) : 0.0f;
        asm volatile("" ::: "memory");                            // the ring reads above stay above the next ticket (whose front releases the slots)
        ST_ACC(st_acc_d);
        if (eprobe) { const unsigned long long t_ = wall_clock64(); ep_d += t_ - ep_t0; ep_t0 = t_; }
        const int row0 = b0 + s * UO;
        if (SWIGLU) {
            const float y = (v0 / (1.0f + expf(-v0))) * v1;
            if (lane == (n_done & 63)) { res = y; rrow = row0; }
            n_done++;
        } else {
            if (lane == (n_done & 63)) { res = v0; rrow = row0; }
            n_done++;
            if (two) {
                if ((n_done & 63) == 0) flush(n_done);
                if (lane == (n_done & 63)) { res = v1; rrow = row0 + 1; }
                n_done++;
            }
        }
        if ((n_done & 63) == 0) flush(n_done);
    }
    asm volatile("" ::: "memory");
    if (lane == 0) st_sync(sy + SY_FRONT + c, (int)(g0 + (unsigned)a.ns_pad));   // no step left for this wave: nothing of this mat-vec is held back by it
    ST_STAMP(4);
    ST_ACC_OUT();
    if (eprobe && lane == 0) { io.probe[1] = ep_w; io.probe[2] = ep_d; }
    if (flushed < n_done) flush(n_done);
#ifdef MI355_STREAM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ST_STAMP(5);
#endif
}
// forms that exist: SwiGLU pairs only with the fused RMSNorm prologue (gate/up), row pairs up to K = 8192, single rows
// from K = 6144 (mmvq_stream_applicable agrees)
template <int KB, int FUSE, int ENG = 0, bool EARLY = false>
__device__ __forceinline__ void consumer_dispatch(const StOp &a, uint8_t *smem, int c, unsigned g0, const StLayout &lay, const EngIO &io = EngIO(), const EarlyAct<KB> *ea = nullptr) {
#define RUN(TY)                                                                                               \
    do {                                                                                                      \
        if (a.swiglu) { if constexpr (FUSE == 1 || (FUSE == 3 && KB <= 4)) consumer_op<TY, KB, FUSE, true, true, ENG, EARLY>(a, smem, c, g0, lay, io, ea); } \
        else if (a.pair) { if constexpr (KB <= 4 || ENG == 0) consumer_op<TY, KB, FUSE, false, true, ENG, EARLY>(a, smem, c, g0, lay, io, ea); } \
        else { if constexpr (KB >= 3) consumer_op<TY, KB, FUSE, false, false, ENG, EARLY>(a, smem, c, g0, lay, io, ea); } \
    } while (0)
    switch (a.type) {
        case T_Q4_K: RUN(T_Q4_K); break;
        case T_Q5_K: RUN(T_Q5_K); break;
        case T_Q6_K: RUN(T_Q6_K); break;
        case T_Q8_0: if constexpr (FUSE == 1 || FUSE == 2) RUN(T_Q8_0); break;
        // round 4 (one mat-vec per launch only: the layer engine keeps to the types it was built with)
        case T_Q2_K: if constexpr (ENG == 0) RUN(T_Q2_K); break;
        case T_Q3_K: if constexpr (ENG == 0) RUN(T_Q3_K); break;
        case T_Q4_0: if constexpr (ENG == 0 && (FUSE == 1 || FUSE == 2)) RUN(T_Q4_0); break;
        case T_Q5_0: if constexpr (ENG == 0 && (FUSE == 1 || FUSE == 2)) RUN(T_Q5_0); break;
        case T_IQ4_NL: if constexpr (ENG == 0 && (FUSE == 1 || FUSE == 2)) RUN(T_IQ4_NL); break;
        default: break;
    }
#undef RUN
}

__device__ __forceinline__ void sync_init(uint8_t *smem) {
    if (threadIdx.x < 64) reinterpret_cast<int *>(smem + ST_OFF_SYNC)[threadIdx.x] = 0;
    __syncthreads();
}

}  // namespace

}  // namespace mi355
