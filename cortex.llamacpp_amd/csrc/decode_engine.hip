// decode_engine.hip — the mat-vecs of one decoder layer of a single-token step inside ONE persistent launch:
//     attn_output (+ residual)  ->  ffn_gate | ffn_up (RMSNorm prologue, SwiGLU)  ->  ffn_down (+ residual)  ->  next layer's Q | K | V (RMSNorm prologue)
// (the llama.cpp graph of SURVEY.md §A.3 between two attention calls; reference call site: llama_decode, src/llama_server_context.cc:1635).
// One workgroup per CU, the loader / consumer roles of mmvq_stream_dev.h: the two loader waves of a CU stream the workgroup's rows of ALL four
// tensors back to back into the 128 KiB LDS ring — they never wait for a hand-over, so while the consumers exchange a mat-vec's results the ring fills
// with the next mat-vec's weights (up to 32 MB on the chip: the whole attn_output / Q|K|V tensors, half of gate|up).  What a launch boundary cost per
// mat-vec (cold kernel arguments and instruction cache, ~1.2 us to the first byte, the decode tail, the boundary itself: ~4 us) is paid
// once per layer.
//
// Hand-overs (MI355X_MICROARCH.md "Persistent kernels ... price list", cdna_hip_programming.md Guideline 16 R2): a result leaves its CU as an 8-byte
// {tag, value} GRANULE written by one device-coherent (sc1) store; the data is its own flag — no counter, no fence, no second round trip.  The tag is
// (step serial, layer, edge), so a slot never has to be cleared.  On the reading side ONE wave per CU — the consumer wave that finished its rows last —
// sweeps the granules with sc1 loads (16 in flight per lane), re-reads a batch until every tag matches, stages the values in LDS and raises an LDS flag
// for the other seven; while it sweeps, the CU's loaders keep only ST_THIN_D slots in flight (a gather queued behind an unthrottled 96 KiB refill burst
// takes 2-3x as long).  Round 2's version of this launch (arrival counters + plain payload + every consumer loading its own
// activation from global memory behind the stream) lost to four launches (52 vs 37.6 us); the differences are exactly these three.
//   x  (E f32)            attn_output / ffn_down -> every CU: 8 B granules, gathered into LDS as f32; RMSNorm * w and the Q8_K quantisation run from LDS
//                          on the 8 consumers as in the per-launch prologue (consumer_prologue<.., XLDS>), the residual of the workgroup's own rows is kept in LDS
//   silu(g) * u (FF f32)  -> Q8_K for ffn_down in two hops: every CU publishes its f32 results as granules; the CU that owns a 256-block gathers that block's
//                          256 granules, quantises it with wave_quant_q8k (the per-launch code: bit-identical) and publishes 64 code granules + the scale;
//                          every CU gathers the K/4 code granules + K/256 scales and forms the 16-element block sums itself (integer sums of the codes)
// The arithmetic of every mat-vec is consumer_op of mmvq_stream_dev.h with the same lane roles and summation order as the per-launch kernels: the step
// is bit-identical to the per-launch path (tests/test_gpu_model.py::test_layer_engine_matches_per_launch_bitwise).
// Every wait is bounded; a wait that gives up raises the sticky error word (stream_set_error_word) and the workgroup's abort flag: the launch ends, the
// step is reported as failed by Context::decode, and the context falls back to one launch per mat-vec.
#include "mmvq_stream_dev.h"

namespace mi355 {

namespace {

#ifndef MI355_ENGINE_GATHER_SPINS
#define MI355_ENGINE_GATHER_SPINS (1 << 17)
#endif
constexpr int EN_GATHER_SPINS = MI355_ENGINE_GATHER_SPINS;
constexpr int EN_NB = 8;                       // granule loads in flight per lane of the gathering wave

// ---- LDS layout behind the ring: A = the gathered f32 vector (E floats) / the Q8_K codes of ffn_down's activation (FF bytes), AD = that activation's
// scales and block sums, B = the Q8_K planes of an E-long activation (qs | d | bs, the layout loader_planes / consumer_prologue use), RS = residual rows
template <int KBE, int KBF> struct EnLayout {
    static constexpr int A = ST_OFF_ACT;
    static constexpr int A_BYTES = KBE * 8192 > KBF * 2048 ? KBE * 8192 : KBF * 2048;
    static constexpr int AD_D = A + A_BYTES, AD_BS = AD_D + 1024, AD_END = AD_BS + ((KBF * 256 + 1023) & ~1023);
    static constexpr int B_QS = AD_END, B_D = B_QS + KBE * 2048, B_BS = B_D + 1024, B_END = B_BS + ((KBE * 256 + 1023) & ~1023);
    static constexpr int RS = B_END, RS_FLOATS = 128, DESC = RS + RS_FLOATS * 4;                 // DESC: the layer's descriptors, copied once
    static constexpr int TOTAL = DESC + (((int)sizeof(EngineLayer) + 15) & ~15);
    __device__ static StLayout lay_e() { StLayout l; l.qs = B_QS; l.d = B_D; l.bs = B_BS; l.total = TOTAL; return l; }
    __device__ static StLayout lay_f() { StLayout l; l.qs = A; l.d = AD_D; l.bs = AD_BS; l.total = TOTAL; return l; }
};

struct EngineArgsDev {
    const EngineLayer *layer;       // device memory; copied into LDS once per launch (by value in the kernel argument segment hipcc reads all 160 words
                                    // at the top of the kernel and keeps them in spilled SGPRs: 250 spills, a slower step)
    unsigned long long *gx, *gs, *gc, *gd, *gx2;
    const unsigned *epoch;
    int layer_index;
    unsigned long long *probe;      // nullable: per wave EN_PROBE_STAMPS wall-clock stamps
};
constexpr int EN_PROBE_STAMPS = 48;
#define EN_STAMP(i) do { if (ea.probe && lane == 0) ea.probe[((size_t)blockIdx.x * ST_NW + wave) * EN_PROBE_STAMPS + (i)] = wall_clock64(); } while (0)

__device__ __forceinline__ bool en_abort(const int *sy) { return ld_sync(sy + SY_ABORT) != 0; }
__device__ __forceinline__ void en_give_up(int *sy, int lane) {
    st_timeout(ST_ERR_GATHER);
    if (lane == 0) st_sync(sy + SY_ABORT, 1);
}

// A wave sweeps granules [0, n) of g: EN_NB loads per lane in flight, a batch is re-read until all its tags match.  f(idx, value) consumes a granule.
template <class F>
__device__ __forceinline__ bool gather_granules(const unsigned long long *g, int n, unsigned tag, int lane, int *sy, F f) {
    for (int base = 0; base < n; base += 64 * EN_NB) {
        unsigned long long x[EN_NB];
        for (int spins = 0;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < EN_NB; k++) {
                const int idx = base + k * 64 + lane;
                const bool in = idx < n;
                x[k] = __hip_atomic_load(g + (in ? idx : n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (!in || (unsigned)(x[k] >> 32) == tag);
            }
            if (__all(ok)) break;
            if (++spins >= EN_GATHER_SPINS || en_abort(sy)) { en_give_up(sy, lane); return false; }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int k = 0; k < EN_NB; k++) {
            const int idx = base + k * 64 + lane;
            if (idx < n) f(idx, (unsigned)x[k]);
        }
    }
    return true;
}

// the last consumer of the workgroup to finish mat-vec number `round` (1 ..) gets true: it opens the workgroup's part of the hand-over
__device__ __forceinline__ bool en_arrive(int *sy, int lane, int round) {
    asm volatile("" ::: "memory");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(sy + SY_DONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = uni(old);
    return old == ST_NC * round - 1;
}
// ... and lets the eight consumers gather: every one sweeps its eighth of the granules (one round trip for all of it instead of one per batch)
__device__ __forceinline__ void en_gather_go(int *sy, int lane, int round) {
    asm volatile("" ::: "memory");
    if (lane == 0) { st_sync(sy + SY_THIN, 1); st_sync(sy + SY_GATHER, round); }
}
__device__ __forceinline__ void en_gather_wait(int *sy, int round) {
    ST_SPIN_WHILE(ld_sync(sy + SY_GATHER) < round && ld_sync(sy + SY_ABORT) == 0, 1);
}
// the gathered values are in LDS once all eight have arrived here; the last one lets the loaders off the leash again
__device__ __forceinline__ void en_gather_done(int *sy, int lane, int round) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(sy + SY_GDONE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (uni(old) == ST_NC * round - 1 && lane == 0) st_sync(sy + SY_THIN, 0);
    ST_SPIN_WHILE(ld_sync(sy + SY_GDONE) < ST_NC * round && ld_sync(sy + SY_ABORT) == 0, 1);
}
// granules [lo, hi) of consumer c out of n: an eighth, in whole quads of 64 (the 16-element block sums are formed across four neighbouring lanes)
__device__ __forceinline__ void en_slice(int n, int c, int &lo, int &hi) {
    const int per = (((n + ST_NC - 1) / ST_NC) + 63) & ~63;
    lo = c * per < n ? c * per : n;
    hi = lo + per < n ? lo + per : n;
}

// RMSNorm * w + Q8_K of the gathered f32 vector by the 8 consumers: consumer_prologue<KB, 1, false> of mmvq_stream_dev.h cut in two, so that the norm
// weights are REQUESTED before the hand-over is waited for (a consumer's small global loads queue behind the CU's weight stream: requested after the
// gather they cost the prologue 1.5 us).  Same blocks per wave, same sums, same order: the same bits.
template <int KB> struct EnNorm { static constexpr int NJW = (KB * 8 + ST_NC - 1) / ST_NC; f32x4_t w[NJW]; };
template <int KB>
__device__ __forceinline__ void en_norm_issue(EnNorm<KB> &nw, const StOp &a, int c, int lane) {
    const int nbt = a.K >> 8;
#pragma unroll
    for (int i = 0; i < EnNorm<KB>::NJW; i++) {
        const int b = c + ST_NC * i;
        const int bc = b < nbt ? b : nbt - 1;
        nw.w[i] = *reinterpret_cast<const f32x4_t *>(a.nw + bc * 256 + lane * 4);
    }
}
template <int KB>
__device__ __forceinline__ void en_norm_finish(const EnNorm<KB> &nw, const StOp &a, const float *xf, uint8_t *smem, const StLayout &lay, int c, int lane, int round,
                                               unsigned long long *pr = nullptr) {   // pr: diagnosis, 4 wall-clock stamps
    constexpr int NJW = EnNorm<KB>::NJW;
    const int nbt = a.K >> 8;
    int8_t *qs = reinterpret_cast<int8_t *>(smem + lay.qs);
    float *d = reinterpret_cast<float *>(smem + lay.d);
    int16_t *bs = reinterpret_cast<int16_t *>(smem + lay.bs);
    double *red = reinterpret_cast<double *>(smem + ST_OFF_RED);
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    f32x4_t rxv[NJW];
#pragma unroll
    for (int i = 0; i < NJW; i++) {
        const int b = c + ST_NC * i;
        const int bc = b < nbt ? b : nbt - 1;
        rxv[i] = *reinterpret_cast<const f32x4_t *>(xf + bc * 256 + lane * 4);
    }
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
    if (pr && lane == 0) pr[0] = wall_clock64();
    consumers_rendezvous(sy + SY_PRO1, lane, round);
    if (pr && lane == 0) pr[1] = wall_clock64();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < ST_NC; w++) tot += red[w];
    // tot / K: a power-of-two K (every model's embedding width so far) divides exactly by a multiplication - the same bits as the division
    const double dk = (double)a.K;
    const float mean = (a.K & (a.K - 1)) == 0 ? (float)(tot * (1.0 / dk)) : (float)(tot / dk);
    const float scale = 1.0f / sqrtf(mean + a.neps);
    // the wave's blocks side by side in ONE straight line (a clamped block where the last one does not exist, only its stores are skipped): each
    // block's quantisation is a chain of wave-level steps (max -> first lane holding it -> its value -> 1 / scale -> codes -> sums) that waits on
    // itself; two of them interleave
    if (pr && lane == 0) pr[2] = wall_clock64();
    uint32_t packed[NJW]; int bsum[NJW]; float dq[NJW];
#pragma unroll
    for (int i = 0; i < NJW; i++) {
        f32x4_t x = rxv[i];
        const f32x4_t ww = nw.w[i];
        x.x = (x.x * scale) * ww.x; x.y = (x.y * scale) * ww.y; x.z = (x.z * scale) * ww.z; x.w = (x.w * scale) * ww.w;
        const float vv[4] = {x.x, x.y, x.z, x.w};
        wave_quant_q8k(vv, lane, packed[i], bsum[i], dq[i]);
    }
#pragma unroll
    for (int i = 0; i < NJW; i++) {
        const int b = c + ST_NC * i;
        if (b < nbt) {
            *reinterpret_cast<uint32_t *>(qs + b * 256 + lane * 4) = packed[i];
            if ((lane & 3) == 0) bs[b * 16 + (lane >> 2)] = (int16_t)bsum[i];
            if (lane == 0) d[b] = dq[i];
        }
    }
    if (pr && lane == 0) pr[3] = wall_clock64();
    consumers_rendezvous(sy + SY_PRO2, lane, round);
}

template <int KBE, int KBF>
__global__ __launch_bounds__(ST_NT) void decode_engine_kernel(const EngineArgsDev ea) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using LY = EnLayout<KBE, KBF>;
    const int wave = uni(tid_now() >> 6);
    const int lane = tid_now() & 63;
    // the layer's descriptors: one copy into LDS by the whole workgroup (a description read from global memory when its mat-vec starts would cost ~1 us
    // of dependent loads on the critical path of every hand-over)
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(ea.layer);
        unsigned *dst = reinterpret_cast<unsigned *>(smem + LY::DESC);
        for (int i = (int)threadIdx.x; i < (int)(sizeof(EngineLayer) / 4); i += ST_NT) dst[i] = src[i];
    }
    const unsigned tag0 = (unsigned)uni((int)ea.epoch[0]) * 1024u + (unsigned)ea.layer_index * 8u;
    { unsigned t = tag0; asm volatile("" :: "s"(t)); }
    sync_init(smem);
    const EngineLayer *L = reinterpret_cast<const EngineLayer *>(smem + LY::DESC);
    const int has_qkv = uni(L->has_qkv);
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    unsigned long long *pw = ea.probe ? ea.probe + ((size_t)blockIdx.x * ST_NW + wave) * EN_PROBE_STAMPS : nullptr;
    EN_STAMP(0);
    StOp a;
    unsigned g0 = 0;
    if (wave < ST_NL) {
        // ---------------- loaders: the four tensors' rows of this workgroup, back to back
        LoaderState st{0, 0, 0, (unsigned)ST_RING_SLOTS, 0};
        op_setup<true>(L->wo, a);
        if (wave == 0) st.pre = loader_planes(a, smem, LY::lay_e(), lane);
        ST_SPIN_WHILE(ld_sync(sy + SY_GO) < ST_NC, 0);
        EN_STAMP(1);
        loader_op<true>(st, a, smem, wave, g0, lane); g0 += (unsigned)a.ns_pad;
        EN_STAMP(2);
        if (pw && lane == 0) { pw[12] = st.w_depth; pw[13] = st.w_space; } st.w_depth = st.w_space = 0;
        op_setup<true>(L->gu, a);
        loader_op<true>(st, a, smem, wave, g0, lane); g0 += (unsigned)a.ns_pad;
        EN_STAMP(3);
        if (pw && lane == 0) { pw[14] = st.w_depth; pw[15] = st.w_space; } st.w_depth = st.w_space = 0;
        op_setup<true>(L->dn, a);
        loader_op<true>(st, a, smem, wave, g0, lane); g0 += (unsigned)a.ns_pad;
        EN_STAMP(4);
        if (pw && lane == 0) { pw[16] = st.w_depth; pw[17] = st.w_space; } st.w_depth = st.w_space = 0;
        if (has_qkv) {
            op_setup<true>(L->qkv, a);
            loader_op<true>(st, a, smem, wave, g0, lane);
        }
        EN_STAMP(5);
        if (pw && lane == 0) { pw[18] = st.w_depth; pw[19] = st.w_space; }
        loader_drain(st, smem, wave);
        EN_STAMP(6);
        return;
    }
    const int c = wave - ST_NL;
    int step_base = 0;                                         // the step tickets run over the launch: every mat-vec uses its steps + one per consumer (the "none left" answers)
    auto steps_of = [](const StOp &o) { return (o.n_rows_wg + ((o.pair && !o.swiglu) ? 1 : 0)) / ((o.pair && !o.swiglu) ? 2 : 1) + ST_NC; };
    float *xf = reinterpret_cast<float *>(smem + LY::A);
    float *rs = reinterpret_cast<float *>(smem + LY::RS);
    int lo, hi;

    // ---------------- 1. attn_output: planes by DMA, residual from global memory, x' as granules
    op_setup<true>(L->wo, a);
    {
        EngIO io; io.gran = ea.gx; io.tag = tag0 + 1u; io.probe = pw ? pw + 20 : nullptr; io.step_base = step_base;
        consumer_dispatch<KBE, 0, 2>(a, smem, c, g0, LY::lay_e(), io);
    }
    g0 += (unsigned)a.ns_pad; step_base += steps_of(a);
    EN_STAMP(1);
    StOp an, ad;
    op_setup<true>(L->gu, an);
    EnNorm<KBE> nrm;
    en_norm_issue<KBE>(nrm, an, c, lane);                       // ffn_norm's weights: requested before the hand-over is waited for
    EN_STAMP(32);
    if (en_arrive(sy, lane, 1)) en_gather_go(sy, lane, 1);
    op_setup<true>(L->dn, ad);
    en_gather_wait(sy, 1);
    EN_STAMP(2);
    en_slice(an.K, c, lo, hi);
    gather_granules(ea.gx + lo, hi - lo, tag0 + 1u, lane, sy, [&](int idx, unsigned v) { reinterpret_cast<unsigned *>(xf)[lo + idx] = v; });
    en_gather_done(sy, lane, 1);
    EN_STAMP(3);
    // the residual of ffn_down's rows of this workgroup: kept aside, A is reused for ffn_down's activation
    if (c == 0) for (int i = lane; i < ad.n_rows_wg && i < LY::RS_FLOATS; i += 64) rs[i] = xf[ad.b0 + i];

    // ---------------- 2. ffn_gate | ffn_up: RMSNorm * w + Q8_K from the gathered x', SwiGLU, results as f32 granules
    a = an;
    en_norm_finish<KBE>(nrm, a, xf, smem, LY::lay_e(), c, lane, 1, pw ? pw + 34 : nullptr);
    {
        EngIO io; io.gran = ea.gs; io.tag = tag0 + 2u; io.probe = pw ? pw + 23 : nullptr; io.step_base = step_base;
        consumer_dispatch<KBE, 3, 1>(a, smem, c, g0, LY::lay_e(), io);
    }
    g0 += (unsigned)a.ns_pad; step_base += steps_of(a);
    EN_STAMP(4);
    const int FF = ad.K, nb = FF >> 8;
    EN_STAMP(33);
    if (en_arrive(sy, lane, 2)) {
        if (lane == 0) st_sync(sy + SY_THIN, 1);
        EN_STAMP(5);
        // the 256-blocks this workgroup quantises: block b belongs to workgroup floor(b * nwg / nb)
        const int nwg = (int)gridDim.x, w = (int)blockIdx.x;
        const int b_lo = (w * nb + nwg - 1) / nwg, b_hi = ((w + 1) * nb + nwg - 1) / nwg;
        bool ok = true;
        for (int b = b_lo; b < b_hi && b < nb && ok; b++) {
            unsigned long long x[4];
            for (int spins = 0;;) {
                bool good = true;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    x[i] = __hip_atomic_load(ea.gs + (size_t)b * 256 + lane * 4 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    good = good && (unsigned)(x[i] >> 32) == tag0 + 2u;
                }
                if (__all(good)) break;
                if (++spins >= EN_GATHER_SPINS || en_abort(sy)) { en_give_up(sy, lane); ok = false; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) break;
            const float vv[4] = {__uint_as_float((unsigned)x[0]), __uint_as_float((unsigned)x[1]), __uint_as_float((unsigned)x[2]), __uint_as_float((unsigned)x[3])};
            uint32_t packed; int bsum; float dq;
            wave_quant_q8k(vv, lane, packed, bsum, dq);
            st_store_granule(ea.gc + (size_t)b * 64 + lane, tag0 + 3u, packed);
            if (lane == 0) st_store_granule(ea.gd + b, tag0 + 3u, __float_as_uint(dq));
        }
        EN_STAMP(6);
        en_gather_go(sy, lane, 2);
    }
    if (has_qkv) { op_setup<true>(L->qkv, an); en_norm_issue<KBE>(nrm, an, c, lane); }   // the next attn_norm's weights: requested a hand-over and a mat-vec early
    en_gather_wait(sy, 2);
    {
        unsigned *qs = reinterpret_cast<unsigned *>(smem + LY::A);
        int16_t *bs = reinterpret_cast<int16_t *>(smem + LY::AD_BS);
        float *dd = reinterpret_cast<float *>(smem + LY::AD_D);
        en_slice(FF >> 2, c, lo, hi);
        gather_granules(ea.gc + lo, hi - lo, tag0 + 3u, lane, sy, [&](int idx, unsigned v) {
            qs[lo + idx] = v;
            int s = dot4((int)v, 0x01010101, 0);              // the 16-element block sum: four granules = four neighbouring lanes
            s += dpp_i<DPP_QP_1032>(s);
            s += dpp_i<DPP_QP_2301>(s);
            if ((lane & 3) == 0) bs[(lo + idx) >> 2] = (int16_t)s;
        });
        if (c == ST_NC - 1) gather_granules(ea.gd, nb, tag0 + 3u, lane, sy, [&](int idx, unsigned v) { reinterpret_cast<unsigned *>(dd)[idx] = v; });
    }
    en_gather_done(sy, lane, 2);
    EN_STAMP(7);

    // ---------------- 3. ffn_down: the gathered Q8_K planes, residual from LDS, x'' as granules and as plain stores (the next launch's residual)
    a = ad;
    {
        EngIO io; io.gran = ea.gx2; io.tag = tag0 + 4u; io.plain = a.out; io.rs = rs; io.probe = pw ? pw + 26 : nullptr; io.step_base = step_base;
        consumer_dispatch<KBF, 3, 1>(a, smem, c, g0, LY::lay_f(), io);
    }
    g0 += (unsigned)a.ns_pad; step_base += steps_of(a);
    EN_STAMP(8);
    if (!has_qkv) return;
    if (en_arrive(sy, lane, 3)) en_gather_go(sy, lane, 3);
    en_gather_wait(sy, 3);
    EN_STAMP(9);
    en_slice(an.K, c, lo, hi);
    gather_granules(ea.gx2 + lo, hi - lo, tag0 + 4u, lane, sy, [&](int idx, unsigned v) { reinterpret_cast<unsigned *>(xf)[lo + idx] = v; });
    en_gather_done(sy, lane, 3);
    EN_STAMP(10);

    // ---------------- 4. the next layer's Q | K | V: RMSNorm * w + Q8_K from the gathered x'', plain stores (the attention launch reads them)
    a = an;
    en_norm_finish<KBE>(nrm, a, xf, smem, LY::lay_e(), c, lane, 2);
    {
        EngIO io; io.probe = pw ? pw + 29 : nullptr; io.step_base = step_base;
        consumer_dispatch<KBE, 3, 1>(a, smem, c, g0, LY::lay_e(), io);
    }
    EN_STAMP(11);
}

}  // namespace

void decode_engine_set_error_word(unsigned *w) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_err_word), &w, sizeof(w)); }

size_t decode_engine_granule_words(int E, int FF) {      // 8-byte words of the hand-over buffers: gx | gx2 | gs | gc | gd
    return (size_t)E * 2 + (size_t)FF + (size_t)(FF >> 2) + (size_t)(FF >> 8) + 64;
}

// the shapes the launch exists for: dense llama layer, K-quant tensors the stream kernel has forms for, E <= 4096 (the f32 vector and ffn_down's codes share
// 16 KiB of LDS behind the ring)
bool decode_engine_applicable(const EngineLayer &l, int E, int FF) {
    if (E % 2048 != 0 || E > 4096 || FF % 1024 != 0) return false;
    const int kbe = E >> 11, kbf = (FF + 2047) >> 11;
    if (!(kbe == 2 && kbf == 7)) return false;          // (FF 11008 / 5632 are no multiples of 1024: those ffn_down tensors have no stream form)
    auto kq = [](int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K; };
    for (const MMVQArgs *a : {&l.wo, &l.gu, &l.dn, &l.qkv}) {
        if (a == &l.qkv && !l.has_qkv) continue;
        if (!mmvq_stream_applicable(*a)) return false;
        const int n = a->epi == EPI_SWIGLU ? 2 : a->n_seg;
        for (int s = 0; s < n; s++) if (!kq(a->seg[s].type)) return false;
    }
    if (l.wo.fuse_mode != 0 || l.gu.fuse_mode != 1 || l.gu.epi != EPI_SWIGLU || l.dn.fuse_mode != 2 || l.dn.epi != EPI_ADD || l.wo.epi != EPI_ADD) return false;
    if (l.wo.K != E || l.gu.K != E || l.dn.K != FF || l.wo.seg[0].n_rows != E || l.dn.seg[0].n_rows != E || l.gu.seg[0].n_rows != FF) return false;
    if (l.has_qkv && (l.qkv.fuse_mode != 1 || l.qkv.K != E || l.qkv.epi != EPI_STORE)) return false;
    // rows per workgroup of the residual-carrying mat-vecs must fit the LDS residual rows
    if ((E + num_cu() - 1) / num_cu() > 128) return false;
    return true;
}
void decode_engine_plan(EngineLayer &l) {
    mmvq_stream_plan(l.wo, num_cu()); mmvq_stream_plan(l.gu, num_cu()); mmvq_stream_plan(l.dn, num_cu());
    if (l.has_qkv) mmvq_stream_plan(l.qkv, num_cu());
}

hipError_t launch_decode_engine(const EngineLayer *layer_dev, int E, int FF, unsigned long long *granules, const unsigned *epoch_dev, int layer_index,
                                unsigned long long *probe, hipStream_t st) {
    EngineArgsDev ea;
    ea.layer = layer_dev;
    ea.gx = granules; ea.gx2 = ea.gx + E; ea.gs = ea.gx2 + E; ea.gc = ea.gs + FF; ea.gd = ea.gc + (FF >> 2);
    ea.epoch = epoch_dev; ea.layer_index = layer_index; ea.probe = probe;
    const int kbe = E >> 11, kbf = (FF + 2047) >> 11;
    const int blocks = num_cu();
#define ENGINE(KE, KF)                                                                                                                   \
    do {                                                                                                                                 \
        const size_t lds = (size_t)EnLayout<KE, KF>::TOTAL;                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&decode_engine_kernel<KE, KF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                                   \
        hipLaunchKernelGGL((decode_engine_kernel<KE, KF>), dim3(blocks), dim3(ST_NT), lds, st, ea);                                      \
    } while (0)
    if (kbe == 2 && kbf == 7) ENGINE(2, 7);
    else return hipErrorInvalidValue;
#undef ENGINE
    return hipGetLastError();
}

}  // namespace mi355
