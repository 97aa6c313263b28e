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
static bool g_stream_anyorder = false;
void mmvq_stream_set_anyorder_for_timing(bool on) { g_stream_anyorder = on; }

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
#define STREAM_K(KERNEL, KBV, FZ)                                                                                        \
    do {                                                                                                                 \
        /* (per launch: the attribute is per device, and a process may hold contexts on several) */                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL<KBV, FZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                   \
        if (g_stream_anyorder) hipExtLaunchKernelGGL((KERNEL<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, nullptr, nullptr, hipExtAnyOrderLaunch, a); \
        else hipLaunchKernelGGL((KERNEL<KBV, FZ>), dim3(blocks), dim3(ST_NT), lds, st, a);                              \
    } while (0)
    // (one macro per prologue form, so that only the kernels a form can take are instantiated)
#define STREAM0(KBV) STREAM_K(mmvq_stream_kernel, KBV, 0)
#define STREAM1(KBV)                                                                                                     \
    do {                                                                                                                 \
        if (role == ROLE_QKV) STREAM_K(mmvq_stream_qkv, KBV, 1);                                                         \
        else if (role == ROLE_GATE_UP) STREAM_K(mmvq_stream_gate_up, KBV, 1);                                            \
        else if (role == ROLE_HEAD) STREAM_K(mmvq_stream_head, KBV, 1);                                                  \
        else STREAM_K(mmvq_stream_kernel, KBV, 1);                                                                       \
    } while (0)
#define STREAM2(KBV) do { if (role == ROLE_DOWN) STREAM_K(mmvq_stream_ffn_down, KBV, 2); else STREAM_K(mmvq_stream_kernel, KBV, 2); } while (0)
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
#undef STREAM_K
    return hipGetLastError();
}

}  // namespace mi355
