// mmvq_stream.hip — single-token quantised mat-vec as a weight STREAM.  One workgroup of 10 waves per CU owns a contiguous
// run of weight rows.  Waves 0-1 are LOADERS: they copy that run HBM -> LDS with the DMA form of the global load
// (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per instruction, no register round trip) into a 128 KiB ring, 4 KiB
// slots, up to 15 slots in flight each, and publish a slot once it has landed.  Waves 2-9 are CONSUMERS: they quantise the
// activation (RMSNorm * w -> Q8_K, the fused prologues) while the ring fills, then decode row pairs out of the ring as
// the slots are published, round-robin in stream order, and report their progress so that the loaders can reuse the space.
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
//     register ring (42 vs 42 us per layer).  With the roles split the loaders are the only waves that ever block on the
//     memory queue, which is exactly the flow control wanted; nobody else issues vector memory in the loop;
//   * a loader never blocks on completion either: what has landed is read off the wave's own vector-memory counter
//     (HW_REG_IB_STS, the counter s_waitcnt vmcnt waits on) between two requests, so a slot is published as soon as it is in
//     LDS.  With counted waits (vmcnt(60)) a slot was published once 15 more had been requested, 5 us later at the streaming
//     rate, and a loader out of ring space had to drain everything before it could publish anything: one slot per memory
//     round trip (gate/up 16.8 -> 14.5 us, down 11.9 -> 10.7);
//   * a launch still pays ~4 us that it cannot hide from itself: cold kernel arguments (0.4-0.7 us per miss), cold
//     instruction cache (~1 us to the first request), ~1.2 us to the first byte, and the decode tail.
// Tried and dropped: attn_output -> gate/up -> down -> next Q/K/V as ONE launch, the loaders streaming
// from one mat-vec's weights into the next while only the consumers wait (sharded arrival counters, write-through
// outputs): bit-identical, and 52 us per layer against 37 for four launches — every small global access of a consumer
// (the next activation, its outputs, the counter polls) queues behind the bulk stream in its CU's memory pipeline and in
// HBM, so each hand-over costs 6-8 us where a kernel boundary costs 1.5.  Also dropped: launches chained across two
// streams with in-kernel waits (HIP promises no dispatch order between graph branches: the graph form deadlocks).
// Weight stream policy: non-temporal (each byte is read once per token).
#include "mmvq_stream_dev.h"

namespace mi355 {

namespace {

// ---- one mat-vec per launch
template <int KB, int FUSE, bool MOE>
__device__ __forceinline__ void stream_body(const MMVQArgs &ka) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef MI355_STREAM_PROBE
    const unsigned long long t_top = wall_clock64();
#endif
    StOp a;
    op_setup<false, MOE>(ka, a);
    const int wave = uni(tid_now() >> 6);
    sync_init(smem);
#ifdef MI355_STREAM_PROBE
    {
        unsigned dep = uni((int)a.total); asm volatile("" : "+s"(dep));
        const unsigned long long t_args = wall_clock64() + (dep & 0);
        if (g_stream_probe && (threadIdx.x & 63) == 0) {
            unsigned long long *pp = &g_stream_probe[(((size_t)(a.nck >> 2) * 256 + blockIdx.x) * ST_NW + wave) * 8];
            pp[0] = t_top; pp[7] = t_args;
        }
    }
#endif
    if (a.n_rows_wg <= 0) return;                               // (more workgroups than rows)
    if (wave < ST_NL) {
        const int lane = tid_now() & 63;
        int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
        LoaderState st{0, 0, 0, (unsigned)ST_RING_SLOTS, 0};
        // (FUSE == 0: the consumers fetch the ready-made activation planes themselves, mmvq_stream_dev.h consumer_op)
        // the consumers' own requests go first: a load queued behind this wave's 60 KiB waits for all of it.  A launch bound by its stream (its share
        // of the weights exceeds the ring, and the prologue reads two 16 KB vectors: gate | up, the output head) lets the stream have a head start of
        // ST_EARLY slots per loader first
        const int early = (FUSE == 1 && a.ns_pad > ST_RING_SLOTS) ? ST_EARLY : 0;
        if (early == 0) ST_SPIN_WHILE(ld_sync(sy + SY_GO) < ST_NC, 0);
        ST_STAMP(2);
        loader_op(st, a, smem, wave, 0u, lane, early);
        ST_STAMP(1);
        loader_drain(st, smem, wave);
        ST_STAMP(5);
        return;
    }
    consumer_dispatch<KB, FUSE>(a, smem, wave - ST_NL, 0u, st_layout(KB));
}
// ---- the STREAM-BOUND roles (gate | up, the output head): the launch ends when its stream does, so the stream's first request is what counts (round 5).
// The launch's geometry is packed once more into the kernel's FIRST 14 argument dwords - scalars the hardware preloads into SGPRs (build.py:
// -amdgpu-kernarg-preload-count=14; tools/preload_probe shows the preload is live on this box), usable in the wave's first instruction while the argument
// struct is cold (0.45 us per round trip):
//   * every wave requests the consumers' activation and norm weights at once (early_issue; the loaders' copies are never looked at: under a condition hipcc
//     waits for the loads where the branches join);
//   * behind the workgroup's first barrier the loaders work out their run of rows from the packed geometry and start the stream - no argument read, no wait
//     for the consumers; the consumers read the argument struct THEN (its round trip is hidden behind a stream that would not have been decoded yet anyway).
// Measured in the harness (tools/exp_stream.hip): gate | up 13.7 -> 13.1 us.  Only for these roles: a launch that is bound by its prologue (ffn_down) or ends
// one decode step behind a short stream (Q | K | V) LOSES when its consumers' argument read and residual request queue behind the stream (6.25 -> 6.8 us and
// 9.97 -> 11.4 us: profiles/r5_exp_fast_start_harness.txt) - those keep the order "all small requests, then the stream".
// kf: K >> 8 (8 bits) | first workgroup of segment 1 (10) | of segment 2 (10) | segments (2) | SwiGLU pair (1) | valid (1).  sg[s]: rows per workgroup (11
// bits) | workgroups that take that many (10) | rows of the next one (11).  rb01 / rb2: row bytes of the segments (16 bits each).  The weights and the norm
// weights as 16-byte units above `wbase` (one arena: 32 bits reach 64 GB).
struct StFast { const float *nx; const uint8_t *wbase; unsigned ow0, ow1, ow2, onw, kf, sg0, sg1, sg2, rb01, rb2; };
// the loader's view of the workgroup's run from the packed geometry: the fields loader_op reads (W, W1, b0, row_bytes, total, swiglu, ns_pad) - the same
// numbers op_setup derives from the argument struct (a consumer checks that: a mismatch raises the sticky error word)
__device__ __forceinline__ void st_fast_setup(StFast f, StOp &o) {
    // (every packed word through an opaque copy: hipcc otherwise sees adjacent kernel arguments selected by one index and builds the selection as a table in
    // scratch memory - 32 bytes per lane and a scratch round trip on the loaders' way to their first request)
    asm volatile("" : "+s"(f.ow0), "+s"(f.ow1), "+s"(f.ow2), "+s"(f.sg0), "+s"(f.sg1), "+s"(f.sg2), "+s"(f.rb01), "+s"(f.rb2));
    const unsigned kf = f.kf;
    const int nseg = (int)((kf >> 28) & 3u), lo1 = (int)((kf >> 8) & 1023u), lo2 = (int)((kf >> 18) & 1023u);
    const bool swiglu = ((kf >> 30) & 1u) != 0;
    const int b = (int)blockIdx.x;
    unsigned g = f.sg0, rb = f.rb01 & 0xffffu, ow = f.ow0;
    int lo = 0;
    if (nseg > 1 && b >= lo1) { g = f.sg1; rb = f.rb01 >> 16; ow = f.ow1; lo = lo1; }
    if (nseg > 2 && b >= lo2) { g = f.sg2; rb = f.rb2 & 0xffffu; ow = f.ow2; lo = lo2; }
    const int rpb = (int)(g & 2047u), full = (int)((g >> 11) & 1023u), rem = (int)(g >> 21);
    const int bl = b - lo;
    const int rows = bl < full ? rpb : bl == full ? rem : 0;
    o.W = f.wbase + ((size_t)ow << 4);
    o.W1 = f.wbase + ((size_t)f.ow1 << 4);
    o.b0 = bl * rpb; o.n_rows_wg = rows; o.row_bytes = rb;
    o.swiglu = swiglu;
    o.total = (unsigned)rows * rb;
    const int per_tensor = (int)((o.total + ST_SLOT - 1) / ST_SLOT);
    o.ns_raw = swiglu ? 2 * per_tensor : per_tensor;
    o.ns_pad = (o.ns_raw + 1) & ~1;
}

template <int KB>
__device__ __forceinline__ void stream_body_fast(const StFast &fa) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = uni(tid_now() >> 6), lane = tid_now() & 63;
    EarlyAct<KB> ea;
    early_issue<KB, 1>(fa.nx, reinterpret_cast<const float *>(fa.wbase + ((size_t)fa.onw << 4)), (int)(fa.kf & 255u) << 8, wave >= ST_NL ? wave - ST_NL : 0, lane, ea);
    sync_init(smem);
    if (wave < ST_NL) {                                         // the stream starts here
        StOp a;
        st_fast_setup(fa, a);
        if (a.n_rows_wg <= 0) return;
        LoaderState st{0, 0, 0, (unsigned)ST_RING_SLOTS, 0};
        loader_op(st, a, smem, wave, 0u, lane, 0);
        loader_drain(st, smem, wave);
        return;
    }
    // the argument struct, read through the segment pointer laundered BEHIND the barrier: hipcc otherwise hoists op_setup's scalar loads (and the wait between
    // its batches) to the top of the kernel, in front of the early requests
    typedef const __attribute__((address_space(4))) MMVQArgs *KArgP;
    static_assert(alignof(MMVQArgs) == 8, "kernel argument layout: 14 preloaded dwords, then MMVQArgs at byte 56");
    const __attribute__((address_space(4))) char *kseg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kseg) :: "memory");
    const __attribute__((address_space(4))) MMVQArgs &ka = *(KArgP)(kseg + 56);
    StOp a;
    op_setup<false, false>(ka, a);
    if (wave == ST_NL && lane == 0) {                           // the packed geometry must describe the run op_setup describes
        StOp la;
        st_fast_setup(fa, la);
        const uint8_t *w_arg = a.W;
        if (la.n_rows_wg != a.n_rows_wg || (a.n_rows_wg > 0 && (la.W != w_arg || la.b0 != a.b0 || la.total != a.total || la.row_bytes != a.row_bytes || la.ns_pad != a.ns_pad ||
                                                                 (a.swiglu && la.W1 != a.W1)))) st_timeout(ST_ERR_LOADER);
    }
    if (a.n_rows_wg <= 0) return;
    consumer_dispatch<KB, 1, 0, true>(a, smem, wave - ST_NL, 0u, st_layout(KB), EngIO(), &ea);
}
#define MI355_ST_KERNEL_FAST(NAME)                                                                            \
    template <int KB>                                                                                         \
    __global__ __launch_bounds__(ST_NT) void NAME(const float *f_nx, const uint8_t *f_wbase, unsigned f_ow0, unsigned f_ow1, unsigned f_ow2, unsigned f_onw, unsigned f_kf, \
                                                  unsigned f_sg0, unsigned f_sg1, unsigned f_sg2, unsigned f_rb01, unsigned f_rb2, const MMVQArgs ka_by_value) {            \
        stream_body_fast<KB>(StFast{f_nx, f_wbase, f_ow0, f_ow1, f_ow2, f_onw, f_kf, f_sg0, f_sg1, f_sg2, f_rb01, f_rb2});                                                \
    }
MI355_ST_KERNEL_FAST(mmvq_stream_gate_up_fast)
MI355_ST_KERNEL_FAST(mmvq_stream_head_fast)
#undef MI355_ST_KERNEL_FAST

// ---- ffn_down (round 6): a PROLOGUE-bound role - its launch ends one decode pass behind "activation ready", and the activation's requests (57 KB of SwiGLU outputs per
// workgroup) are the head of that chain.  They are issued in the kernel's first instructions from two preloaded argument words (the vector's address, K >> 8), 0.45 us
// before the cold argument segment answers its first read; everything else keeps the order of stream_body ("all small requests, then the stream": the loaders wait for
// [SY_GO] as before - the early STREAM start of the roles above loses here, profiles/r5_exp_fast_start_harness.txt).
template <int KB>
__device__ __forceinline__ void stream_body_early(const float *f_nx, unsigned f_k8) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int wave = uni(tid_now() >> 6), lane = tid_now() & 63;
    EarlyAct<KB> ea;
    early_issue<KB, 2>(f_nx, nullptr, (int)f_k8 << 8, wave >= ST_NL ? wave - ST_NL : 0, lane, ea);   // (the loaders' copies are never looked at)
    sync_init(smem);
    typedef const __attribute__((address_space(4))) MMVQArgs *KArgP;
    const __attribute__((address_space(4))) char *kseg = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kseg) :: "memory");                   // (behind the barrier: hipcc otherwise hoists op_setup's scalar loads in front of the early requests)
    const __attribute__((address_space(4))) MMVQArgs &ka = *(KArgP)(kseg + 16);
    StOp a;
    op_setup<false, false>(ka, a);
    if (a.n_rows_wg <= 0) return;
    if (wave < ST_NL) {
        int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
        LoaderState st{0, 0, 0, (unsigned)ST_RING_SLOTS, 0};
        ST_SPIN_WHILE(ld_sync(sy + SY_GO) < ST_NC, 0);
        loader_op(st, a, smem, wave, 0u, lane, 0);
        loader_drain(st, smem, wave);
        return;
    }
    if (wave == ST_NL && lane == 0 && (a.nx != f_nx || (unsigned)(a.K >> 8) != f_k8)) st_timeout(ST_ERR_LOADER);   // the preloaded words must describe the launch's activation
    consumer_dispatch<KB, 2, 0, true>(a, smem, wave - ST_NL, 0u, st_layout(KB), EngIO(), &ea);
}
template <int KB>
__global__ __launch_bounds__(ST_NT) void mmvq_stream_ffn_down_early(const float *f_nx, unsigned f_k8, const MMVQArgs ka_by_value) {
    static_assert(alignof(MMVQArgs) == 8, "kernel argument layout: two preloaded words (12 bytes), then MMVQArgs at byte 16");
    stream_body_early<KB>(f_nx, f_k8);
}

// The same body under one kernel NAME per role a decode step launches it in (round 5): a kernel trace (rocprofv3 --kernel-trace --stats) then carries one row
// per role - Q | K | V, gate | up, ffn_down, the output head - and the achieved bytes per second of each can be worked out from profiles/ alone
// (tools/assemble_profiles_r5.py); launches of any other shape (attn_output where it is a launch of its own, expert launches, ..) keep the plain name.
#define MI355_ST_KERNEL(NAME)                                                                                 \
    template <int KB, int FUSE, bool MOE = false>                                                             \
    __global__ __launch_bounds__(ST_NT) void NAME(const MMVQArgs ka) { stream_body<KB, FUSE, MOE>(ka); }
MI355_ST_KERNEL(mmvq_stream_kernel)
MI355_ST_KERNEL(mmvq_stream_qkv)
MI355_ST_KERNEL(mmvq_stream_gate_up)
MI355_ST_KERNEL(mmvq_stream_ffn_down)
MI355_ST_KERNEL(mmvq_stream_head)
#undef MI355_ST_KERNEL

}  // namespace

#ifdef MI355_STREAM_PROBE
void mmvq_stream_set_probe(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stream_probe), &p, sizeof(p)); }
#endif

#ifndef MI355_STREAM_Q80_MIN_MB
#define MI355_STREAM_Q80_MIN_MB 16
#endif
static constexpr size_t STREAM_Q80_MIN_BYTES = (size_t)MI355_STREAM_Q80_MIN_MB << 20;   // launches of Q8_0 tensors only: below this the register ring (TinyLlama: 850 vs 816 tok/s)

bool mmvq_stream_applicable(const MMVQArgs &a) {
    if (a.T != 1 || a.K <= 0 || (a.K % 256) != 0) return false;      // (whole super-blocks; a partial last pass decodes zero-scale slices, as on the register ring)
    if (a.out_host && (a.n_seg != 1 || a.epi != EPI_STORE || a.n_sel > 1 || a.seg[0].expert_sel)) return false;   // the host copy: one plain segment only
    // the selected experts of one token in one launch (n_sel): one tensor or one gate | up pair, results stored per expert (no residual epilogue)
    if (a.n_sel > 1 && (a.n_sel > 8 || (a.epi == EPI_SWIGLU ? 1 : a.n_seg) != 1 || a.epi == EPI_ADD)) return false;
    if (a.n_sel > 1 || a.seg[0].expert_sel) {                   // (the forms launch_mmvq_stream instantiates for expert launches)
        const int kbm = (a.K + 2047) >> 11;
        if (!((a.fuse_mode == 1 && (kbm == 1 || kbm == 2)) || (a.fuse_mode == 2 && (kbm == 2 || kbm == 4 || kbm == 7)))) return false;
    }
    const int kb = (a.K + 2047) >> 11;
    if (kb != 1 && kb != 2 && kb != 3 && kb != 4 && kb != 6 && kb != 7 && kb != 10) return false;    // (10: 18944 / 20480 - Qwen2-7B's and Yi-34B's feed-forward widths, single-row steps)
    if (a.fuse_mode < 0 || a.fuse_mode > 2) return false;
    if (a.fuse_mode == 1 && a.K > 8192) return false;
    const bool swiglu = a.epi == EPI_SWIGLU;
    if (swiglu && a.fuse_mode != 1) return false;
    if (swiglu && (a.n_seg != 2 || a.seg[0].type != a.seg[1].type || a.seg[0].n_rows != a.seg[1].n_rows || a.seg[0].row_bytes != a.seg[1].row_bytes)) return false;
    const int n = swiglu ? 1 : a.n_seg;
    // Q8_0 tensors: 1.06 B per weight through LDS and no shorter decode; measured on TinyLlama-1.1B Q8_0 the register ring is 4 %
    // faster (850 vs 816 tok/s), so launches made of Q8_0 tensors only stay there (a Q8_0 attn_k / attn_v beside a K-quant attn_q
    // of an 8-expert file streams with it)
    // (round 4: from STREAM_Q80_MIN_BYTES on they stream too - at Llama-3-8B's sizes the launch is bound by its bytes like any other)
    bool all_q80 = true;
    size_t launch_bytes = 0;
    for (int s = 0; s < (swiglu ? 2 : n); s++) { all_q80 = all_q80 && a.seg[s].type == T_Q8_0; launch_bytes += (size_t)a.seg[s].n_rows * a.seg[s].row_bytes; }
    if (all_q80 && launch_bytes < STREAM_Q80_MIN_BYTES) return false;
    for (int s = 0; s < (swiglu ? 2 : n); s++) {
        const MMVQSeg &g = a.seg[s];
        const int t = g.type;
        if (t != T_Q4_K && t != T_Q5_K && t != T_Q6_K && t != T_Q8_0 && t != T_Q2_K && t != T_Q3_K && t != T_Q4_0 && t != T_Q5_0 && t != T_IQ4_NL) return false;
        if (g.expert_sel && (s != 0 && !(swiglu && s == 1))) return false;      // experts on segment 0 (and the up tensor of its pair) only
        if ((g.row_bytes % 16) != 0 || g.row_bytes < 512) return false;         // (512: a Q2_K row of K = 2048 - TinyLlama's hidden size in the smoke model's file - is 672 B)
        if (g.row_bytes > (size_t)ST_MAX_STEP) return false;
        if (kb <= 2 && 2 * g.row_bytes > (size_t)ST_PAIR_MAX) return false;      // K <= 4096 only has the row-pair form
        // (K >= 10240: rows of the 2- and 3-bit formats are short enough for the row-pair form too - Q2_K at K = 14336: 4704 B)
        if ((reinterpret_cast<uintptr_t>(g.W) & 15) != 0) return false;
        if (a.fuse_mode == 0 && (act_is_q80(t) || !a.aq || !a.ad || !a.abs)) return false;
    }
    if (a.fuse_mode == 0 && (((uintptr_t)a.aq | (uintptr_t)a.ad | (uintptr_t)a.abs) & 15) != 0) return false;
    if (a.fuse_mode != 0 && (reinterpret_cast<uintptr_t>(a.nx) & 15) != 0) return false;
    if (a.fuse_mode == 1 && (reinterpret_cast<uintptr_t>(a.nw) & 15) != 0) return false;
    return true;
}

void mmvq_stream_set_error_word(unsigned *w) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_err_word), &w, sizeof(w)); }

// workgroups per segment in proportion to its bytes (every workgroup streams about the same number of bytes)
void mmvq_stream_plan(MMVQArgs &a, int max_blocks) {
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

// tools/exp_stream.hip ONLY: dispatch without the barrier behind the previous launch of the stream.  Nothing makes a launch wait for the
// results of its predecessor then - the harness uses it to price what overlapping a launch's ramp with its predecessor's tail could gain
// at most (the numbers it computes that way are garbage).
static KernelTimer *g_kernel_timer = nullptr;
void set_kernel_timer(KernelTimer *t) { g_kernel_timer = t; }
KernelTimer *kernel_timer() { return g_kernel_timer; }
static bool g_stream_anyorder = false;
void mmvq_stream_set_anyorder_for_timing(bool on) { g_stream_anyorder = on; }

// the packed geometry of a planned launch (StFast); kf == 0 (not valid) where a field does not fit - the launch then takes the plain kernel
static StFast st_fast_pack(const MMVQArgs &a, int blocks) {
    StFast f{};
    const bool swiglu = a.epi == EPI_SWIGLU;
    const int nseg = a.n_seg;                                    // (after mmvq_stream_plan: 1 for a SwiGLU pair)
    if (a.n_sel > 1 || a.seg[0].expert_sel || a.fuse_mode != 1 || !a.nx || !a.nw || nseg < 1 || nseg > 3 || (a.K >> 8) > 255 || a.out_host) return f;
    uintptr_t lo = ~(uintptr_t)0, hi = 0;
    const void *ptrs[4] = {a.seg[0].W, nseg > 1 || swiglu ? (const void *)a.seg[1].W : (const void *)a.seg[0].W, nseg > 2 ? a.seg[2].W : a.seg[0].W, (const void *)a.nw};
    for (int i = 0; i < 4; i++) { const uintptr_t p = (uintptr_t)ptrs[i]; if (p & 15) return f; lo = p < lo ? p : lo; hi = p > hi ? p : hi; }
    if (((hi - lo) >> 4) > 0xffffffffull) return f;
    f.nx = a.nx; f.wbase = reinterpret_cast<const uint8_t *>(lo);
    f.ow0 = (unsigned)(((uintptr_t)ptrs[0] - lo) >> 4); f.ow1 = (unsigned)(((uintptr_t)ptrs[1] - lo) >> 4);
    f.ow2 = (unsigned)(((uintptr_t)ptrs[2] - lo) >> 4); f.onw = (unsigned)(((uintptr_t)ptrs[3] - lo) >> 4);
    unsigned sg[3] = {0, 0, 0}, rb[3] = {0, 0, 0};
    for (int s = 0; s < nseg; s++) {
        const int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
        if (nblk <= 0 || a.seg[s].row_bytes > 0xffffu) return f;
        const int rpb = (a.seg[s].n_rows + nblk - 1) / nblk;     // (op_setup's deal of the rows)
        const int full = a.seg[s].n_rows / rpb, rem = a.seg[s].n_rows - full * rpb;
        if (rpb > 2047 || full > 1023 || rem > 2047) return f;
        sg[s] = (unsigned)rpb | ((unsigned)full << 11) | ((unsigned)rem << 21);
        rb[s] = (unsigned)a.seg[s].row_bytes;
    }
    if (a.seg_block0[0] != 0 || a.seg_block0[1] > 1023 || a.seg_block0[2] > 1023 || blocks != a.seg_block0[3]) return f;
    if (swiglu && (a.seg[1].row_bytes != a.seg[0].row_bytes || a.seg[1].n_rows != a.seg[0].n_rows)) return f;
    f.sg0 = sg[0]; f.sg1 = sg[1]; f.sg2 = sg[2]; f.rb01 = rb[0] | (rb[1] << 16); f.rb2 = rb[2];
    f.kf = (unsigned)(a.K >> 8) | ((unsigned)(nseg > 1 ? a.seg_block0[1] : 0) << 8) | ((unsigned)(nseg > 2 ? a.seg_block0[2] : 0) << 18) | ((unsigned)nseg << 28) |
           (swiglu ? 1u << 30 : 0u) | (1u << 31);
    return f;
}

hipError_t launch_mmvq_stream(MMVQArgs a, hipStream_t st) {
    if (!mmvq_stream_applicable(a)) return hipErrorInvalidValue;
    const int kb = (a.K + 2047) >> 11;
    mmvq_stream_plan(a, num_cu());
    const int blocks = a.seg_block0[3];
    const size_t lds = (size_t)st_layout(kb).total;
    // the role of the launch in a decode step, told from its shape (only the kernel's NAME depends on it): Q | K | V = two or three segments behind the fused
    // RMSNorm; gate | up = the SwiGLU pair; ffn_down = one segment that quantises its input and adds the residual; the head = one long segment behind the norm
    enum { ROLE_ANY, ROLE_QKV, ROLE_GATE_UP, ROLE_DOWN, ROLE_HEAD };
    int role = ROLE_ANY;
    if (a.n_sel <= 1 && !a.seg[0].expert_sel) {
        if (a.fuse_mode == 1 && a.epi == EPI_SWIGLU) role = ROLE_GATE_UP;
        else if (a.fuse_mode == 1 && a.epi == EPI_STORE && a.n_seg >= 2) role = ROLE_QKV;
        else if (a.fuse_mode == 1 && a.epi == EPI_STORE && a.n_seg == 1 && a.seg[0].n_rows >= 16384) role = ROLE_HEAD;
        else if (a.fuse_mode == 2 && a.epi == EPI_ADD && a.n_seg == 1) role = ROLE_DOWN;
    }
    const char *const role_name = role == ROLE_QKV ? "qkv" : role == ROLE_GATE_UP ? "gate_up" : role == ROLE_DOWN ? "ffn_down" : role == ROLE_HEAD ? "head" : "stream";
    (void)role_name;
    // the stream-bound roles start their stream from preloaded geometry where the packing fits (stream_body_fast)
    static const bool fast_off = getenv("MI355_STREAM_FAST_START") && getenv("MI355_STREAM_FAST_START")[0] == '0';
    const StFast fp = (role == ROLE_GATE_UP || role == ROLE_HEAD) && !fast_off && !g_stream_anyorder ? st_fast_pack(a, blocks) : StFast{};
    const bool fast_ok = (fp.kf >> 31) != 0;
    static const bool down_early_off = getenv("MI355_STREAM_DOWN_EARLY") && getenv("MI355_STREAM_DOWN_EARLY")[0] == '0';
    const bool down_early = !down_early_off && !g_stream_anyorder && a.nx && !a.out_host;
#define STREAM_K(KERNEL, KBV, FZ)                                                                                        \
    do {                                                                                                                 \
        /* (per launch: the attribute is per device, and a process may hold contexts on several) */                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<KBV, FZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                   \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                       \
        if (g_stream_anyorder) hipExtLaunchKernelGGL((KERNEL<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a); \
        else if (g_kernel_timer && g_kernel_timer->next(role_name, &ev0_, &ev1_)) hipExtLaunchKernelGGL((KERNEL<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, ev0_, ev1_, 0, a); \
        else hipLaunchKernelGGL((KERNEL<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, a);                              \
    } while (0)
    // (one macro per prologue form, so that only the kernels a form can take are instantiated)
#define STREAM0(KBV) STREAM_K(mmvq_stream_kernel, KBV, 0)
#define STREAM_FAST(KERNEL, KBV)                                                                                         \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<KBV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                   \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                       \
        if (g_kernel_timer && g_kernel_timer->next(role_name, &ev0_, &ev1_))                                             \
            hipExtLaunchKernelGGL((KERNEL<KBV>), dim3(blocks), dim3(ST_NT), lds, st, ev0_, ev1_, 0, fp.nx, fp.wbase, fp.ow0, fp.ow1, fp.ow2, fp.onw, fp.kf, fp.sg0, fp.sg1, fp.sg2, fp.rb01, fp.rb2, a); \
        else hipLaunchKernelGGL((KERNEL<KBV>), dim3(blocks), dim3(ST_NT), lds, st, fp.nx, fp.wbase, fp.ow0, fp.ow1, fp.ow2, fp.onw, fp.kf, fp.sg0, fp.sg1, fp.sg2, fp.rb01, fp.rb2, a); \
    } while (0)
#define STREAM1(KBV)                                                                                                     \
    do {                                                                                                                 \
        if (role == ROLE_QKV) STREAM_K(mmvq_stream_qkv, KBV, 1);                                                         \
        else if (role == ROLE_GATE_UP && fast_ok) STREAM_FAST(mmvq_stream_gate_up_fast, KBV);                            \
        else if (role == ROLE_GATE_UP) STREAM_K(mmvq_stream_gate_up, KBV, 1);                                            \
        else if (role == ROLE_HEAD && fast_ok) STREAM_FAST(mmvq_stream_head_fast, KBV);                                  \
        else if (role == ROLE_HEAD) STREAM_K(mmvq_stream_head, KBV, 1);                                                  \
        else STREAM_K(mmvq_stream_kernel, KBV, 1);                                                                       \
    } while (0)
#define STREAM_EARLY(KBV)                                                                                                \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mmvq_stream_ffn_down_early<KBV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                   \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                       \
        if (g_kernel_timer && g_kernel_timer->next(role_name, &ev0_, &ev1_))                                             \
            hipExtLaunchKernelGGL((mmvq_stream_ffn_down_early<KBV>), dim3(blocks), dim3(ST_NT), lds, st, ev0_, ev1_, 0, a.nx, (unsigned)(a.K >> 8), a); \
        else hipLaunchKernelGGL((mmvq_stream_ffn_down_early<KBV>), dim3(blocks), dim3(ST_NT), lds, st, a.nx, (unsigned)(a.K >> 8), a); \
    } while (0)
#define STREAM2(KBV)                                                                                                     \
    do {                                                                                                                 \
        if constexpr (KBV == 6 || KBV == 7) { if (role == ROLE_DOWN && down_early) { STREAM_EARLY(KBV); break; } }       \
        if (role == ROLE_DOWN) STREAM_K(mmvq_stream_ffn_down, KBV, 2); else STREAM_K(mmvq_stream_kernel, KBV, 2);        \
    } while (0)
#define STREAM_F(KBV) do { if (a.fuse_mode == 0) STREAM0(KBV); else if (a.fuse_mode == 1) STREAM1(KBV); else STREAM2(KBV); } while (0)
    // a mixture-of-experts step (expert index read on the device): the forms its two launches take - gate | up with the fused RMSNorm prologue over the hidden
    // size, down with the quantise-only prologue over the feed-forward width
#define STREAM_MOE(KBV, FZ)                                                                                              \
    do {                                                                                                                 \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mmvq_stream_kernel<KBV, FZ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                   \
        hipLaunchKernelGGL((mmvq_stream_kernel<KBV, FZ, true>), dim3(blocks), dim3(ST_NT), lds, st, a);                 \
    } while (0)
    if (a.n_sel > 1 || a.seg[0].expert_sel) {
        if (a.fuse_mode == 1 && kb == 1) STREAM_MOE(1, 1);
        else if (a.fuse_mode == 1 && kb == 2) STREAM_MOE(2, 1);
        else if (a.fuse_mode == 2 && kb == 2) STREAM_MOE(2, 2);
        else if (a.fuse_mode == 2 && kb == 4) STREAM_MOE(4, 2);
        else if (a.fuse_mode == 2 && kb == 7) STREAM_MOE(7, 2);
        else return hipErrorInvalidValue;
        goto STREAM_DONE;
    }
    switch (kb) {
        case 1: STREAM_F(1); break;
        case 2: STREAM_F(2); break;
        case 3: STREAM_F(3); break;
        case 4: STREAM_F(4); break;
        case 10: if (a.fuse_mode == 2) STREAM2(10); else if (a.fuse_mode == 0) STREAM0(10); else return hipErrorInvalidValue; break;
        case 6: if (a.fuse_mode == 2) STREAM2(6); else if (a.fuse_mode == 0) STREAM0(6); else return hipErrorInvalidValue; break;
        case 7: if (a.fuse_mode == 2) STREAM2(7); else if (a.fuse_mode == 0) STREAM0(7); else return hipErrorInvalidValue; break;
        default: return hipErrorInvalidValue;
    }
    STREAM_DONE:
#undef STREAM_MOE
#undef STREAM_F
#undef STREAM0
#undef STREAM1
#undef STREAM2
#undef STREAM_EARLY
#undef STREAM_FAST
#undef STREAM_K
    return hipGetLastError();
}

}  // namespace mi355
