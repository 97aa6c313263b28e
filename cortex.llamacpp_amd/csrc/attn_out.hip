// attn_out.hip — single-token step: the decode attention AND the attn_output mat-vec in ONE launch (round 4).
//
// Reference call site: llama_decode inside UpdateSlots (src/llama_server_context.cc:1635); ops: ggml_rope_ext + the KV store + ggml_flash_attn_ext +
// ggml_mul_mat(attn_output) + the residual add of llm_build_llama.
//
// Why one launch.  The per-launch path ran [attention: 72 workgroups, 9.8 us] -> boundary -> [attn_output: 256 workgroups, 5.9 us for 9.4 MB] per layer.
// The second launch spends 2.7 us between "arguments loaded" and "activation ready" (cold instruction cache, the first memory round trip for 4.6 KB of
// ready-made codes) while its loaders fill rings nobody can decode yet, plus the ~1.5 us boundary - and the first launch leaves 184 CUs idle for 10 us.
// Here every workgroup (one per CU, 8 waves) owns a contiguous run of W_o rows AND, if its index is below the number of (kv head, chunk) items, one
// attention item:
//   1. item workgroups run the attention chunk exactly as flash_attn_decode_item does (rope of q and of the new K row, KV store, scores, chunk softmax,
//      P.V, partial record published write-through, ticket; the last arriver of a kv head merges the chunks and quantises the head group's 256-blocks);
//      the others start at once with 2;
//   2. the workgroup's W_o rows (36 KB at 4096 x 4096 Q4_K over 256 workgroups) are copied HBM -> LDS with the DMA form of the global load
//      (global_load_lds_dwordx4, mmvq_stream_dev.h) - item workgroups issue it behind their ticket (hipcc's counted waits for the item's own loads, and
//      the drain of its partial stores, must never sit behind 36 KB of DMA), the merging ones behind their flag - and are on chip before the merge ends;
//   3. the merging workgroups publish the Q8_K codes of their blocks write-through (sc1) and raise ONE flag per kv-head group = the step's serial number
//      (a device word the step's set-up launch increments, so a hipGraph replay needs no reset and no per-launch argument);
//   4. every workgroup: one wave polls the flags (relaxed, bounded), then all threads fetch the 4.6 KB of codes with sc1 loads into LDS, and the eight
//      waves decode one row pair each out of LDS with the weight stream's decoders (same lane roles, same f32 order: the mat-vec's output bits are those of
//      mmvq_stream_kernel given the same codes) and store row + residual.
// Nothing waits before its own items are done, so the launch cannot deadlock on itself; every spin is bounded and raises the sticky error word.
// The attention arithmetic is that of attn_decode_dev.h with 512 threads per item instead of 256 (16 cell groups instead of 8 in the P.V pass: another
// f32 summation order, within the parity tolerance) and, from 2049 cells on, 128-cell chunks (so that a 4096-cell context is 256 items = one
// per workgroup and the merge sums at most 32 partials in one request round).
//
// Round 6: where the layer's attn_q | attn_k | attn_v rows fit in LDS beside the W_o rows, the Q | K | V mat-vecs (with their RMSNorm -> Q8_K prologue) run in front
// of all this IN THE SAME LAUNCH (qkv_attn_out_kernel; the note in front of qf_setup below): one launch per layer for the whole attention block of a
// single-token step.  attn_out_kernel stays the form for every other shape, unchanged.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "attn_decode_dev.h"
#include "mmvq_stream_dev.h"

namespace mi355 {

namespace {

#ifndef MI355_AO_POLL_SLEEP
#define MI355_AO_POLL_SLEEP 1
#endif
#ifndef MI355_AO_SWEEP_SLEEP
#define MI355_AO_SWEEP_SLEEP 1
#endif
// Every word several workgroups poll or count on has its 128-byte line to itself (ATT_SYNC_STRIDE words apart, kernels.h): with the eight ticket counters
// in one line and the eight flags in another the step was 2.5 % slower (context filled: 4 %) - atomics and polls of different kv heads queued on one line
// (same-box A/B, tools/ab_libs.sh).
constexpr int AO_NT = 512, AO_NW = AO_NT / 64;
constexpr int AO_D = 128, AO_NB = AO_D / 32, AO_NCG = AO_NT / (AO_D / 4);      // 16 cell groups of 32 lanes in the P.V pass
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr int AO_GPB = 96;                     // granules per 256-block of the quantised attention output (attn_out_granule_words)
#ifndef MI355_AO_REC
#define MI355_AO_REC (AO_D + 4)
#endif
constexpr int AO_REC = MI355_AO_REC;            // floats of a chunk's partial record per head: O [D] | m | l | 0 | 0 (whole 16-byte pieces; padding the record to whole 128-byte lines measured no gain)

struct AOArgs {
    const uint8_t *W; float *out; const float *resid;
    int type, n_rows, K, epi;
    unsigned row_bytes;
    int rows_per_wg;
    unsigned slice_lds;             // LDS bytes reserved for the workgroup's rows (whole 4 KiB slots)
    unsigned *flags;                // this layer's flag words, one per merge ticket group: "this group's merge has its sums" (the codes follow as granules)
    unsigned long long *gran;       // the quantised attention output as tagged granules {tag, value}, AO_GPB per 256-block: 64 x four codes | 16 block sums | scale
    int layer;                      // tag = serial * 256 + layer + 1: distinct per step and per layer (the granule buffer is shared by the layers of a step)
    const unsigned *serial;         // device word: the step's serial number (never 0)
    int n_flags, n_items;           // ticket groups; attention items = G * splits
    unsigned long long *probe;      // diagnosis (MI355_AO_PROBE=1): per workgroup 16 wall-clock stamps
};

template <int R, int C> struct AOSmem {
    alignas(16) float qf[R * AO_D];
    alignas(16) int8_t qc[R * AO_D];
    float qd[R * AO_NB];
    float S[R * C];
    float ml[R * 2];
    int vis[C];
    uint32_t ksc[C * AO_NB / 2], vsc[C * AO_NB / 2];
    alignas(16) float accs[AO_NCG * R * AO_D];
    alignas(16) uint8_t newk[AO_D * 2];
    alignas(16) uint8_t newv[AO_D * 2];
    uint32_t newkd[2], newvd[2];
    int last_flag;
};
template <int R, int C> struct AOSmemQ {   // QF: AOSmem without the P.V / merge scratch `accs` (AO_NCG * R * D floats), which lives in the space of the Q | K | V rows once they have been decoded
    alignas(16) float qf[R * AO_D];
    alignas(16) int8_t qc[R * AO_D];
    float qd[R * AO_NB];
    float S[R * C];
    float ml[R * 2];
    int vis[C];
    uint32_t ksc[C * AO_NB / 2], vsc[C * AO_NB / 2];
    alignas(16) uint8_t newk[AO_D * 2];
    alignas(16) uint8_t newv[AO_D * 2];
    uint32_t newkd[2], newvd[2];
    int last_flag;
};

// Round 6, QF (KB > 0): Q | K | V of the layer inside this launch (see qf_* below and the note in front of ao_body).  The three tensors as the weight
// stream would plan them (mmvq_stream_plan: workgroups per tensor in proportion to its bytes, a workgroup's rows are of ONE tensor, contiguous).
struct QFArgs {
    const uint8_t *W0, *W1, *W2;
    unsigned rb0, rb1, rb2;
    int t0, t1, t2, n0, n1, n2;
    int blk1, blk2, blk3;            // first workgroup of tensor 1, of tensor 2, and one past the last workgroup with rows
    const float *nx, *nw; float neps; int K;     // the layer's input row, the attn_norm weights; K = n_embd (== H * D here)
    unsigned long long *gran;         // the token's q | k | v as tagged granules {tag, f32 bits}: H * D + 2 * G * D of them, shared by the layers of a step (the tag names the layer)
    int n_q, n_kv;                    // H * D, G * D
    unsigned qkv_lds;                 // LDS bytes reserved for the workgroup's Q | K | V rows (whole 4 KiB slots)
    unsigned accs_off;                // LDS byte offset of the attention's P.V / merge scratch (AO_NCG * R * D floats)
};
constexpr int SY_QF_GO = 40;          // QF sync word: waves whose activation requests are queued (the loaders start behind them)
constexpr unsigned AO_QF_HDR = 512;   // QF: the weight stream's sync words and reduction scratch in front of everything (mmvq_stream_dev.h ST_OFF_SYNC, ST_OFF_RED)

// LDS: [QF: 512 B of sync words] | [.., + slice_lds) the W_o rows | [QF: the Q | K | V rows] | qs [K] | d (1 KiB of room) | bs [K / 8, whole KiB] | AOSmem
struct AOLayout { unsigned wo, qkv, qs, d, bs, attn, total; };
__host__ __device__ inline AOLayout ao_layout(unsigned slice_lds, int K, size_t attn_bytes, unsigned hdr = 0, unsigned qkv_lds = 0) {
    AOLayout l;
    l.wo = hdr;
    l.qkv = hdr + slice_lds;
    l.qs = l.qkv + qkv_lds;
    l.d = l.qs + (unsigned)K;
    l.bs = l.d + 1024u;
    l.attn = l.bs + (((unsigned)K / 8u + 1023u) & ~1023u);
    l.total = l.attn + (unsigned)((attn_bytes + 15) & ~(size_t)15);
    return l;
}

// the workgroup's rows, HBM -> LDS: wave w copies the 4 KiB slots w, w + 8, ..; the tail re-reads the run's last 16 B (as loader_op does)
__device__ __forceinline__ void ao_issue_dma(const AOArgs &o, uint8_t *smem, int b0, int nrw, int wave, int lane) {
    if (nrw <= 0) return;
    const unsigned total = (unsigned)nrw * o.row_bytes, last = total - 16u;
    const uint8_t *src = o.W + (size_t)b0 * o.row_bytes;
    const unsigned lds0 = lds_addr(smem);
    const int n_slots = (int)((total + ST_SLOT - 1) / ST_SLOT);
    for (int s = wave; s < n_slots; s += AO_NW) {
        const unsigned off = (unsigned)s * ST_SLOT;
        if (off + ST_SLOT <= total) dma_slot(src + off + (size_t)lane * 16, lds0 + off);
        else {
#pragma unroll
            for (int p = 0; p < ST_SI; p++) {
                const unsigned x = off + (unsigned)lane * 16u + p * 1024u;
                dma16<true>(src + (x < total ? x : last), lds0 + off + p * 1024);
            }
        }
    }
}

// What an item requests from memory before it can compute anything (QF: q and the token's K / V row arrive as granules instead - ao_item_run).
template <int R, int TK, int TV, int C> struct AOItemLd {
    static constexpr int KROW = TK == T_F16 ? 2 * AO_D : AO_D, LPC = KROW / 16, KP = C * LPC / AO_NT, CPG = C / AO_NCG;
    int chunk; bool skip;
    int cpos; unsigned long long cseq;
    f32x2_t qv, csv;
    u32x4_t kreg[KP];
    u32x2_t vreg[CPG];
    uint32_t ks2, vs2;
    f32x4_t xn4, cs4;
};
template <int R, int TK, int TV, int C, bool QF>
__device__ __forceinline__ void ao_item_issue(const AttnArgs &a, const float *cs_table, int n_rot, const DecodeFuse &fz, const AOArgs &o, int g, int sp, AOItemLd<R, TK, TV, C> &ld) {
    constexpr int D = AO_D, NB = AO_NB, NT = AO_NT;
    constexpr int KROW = TK == T_F16 ? 2 * D : D;
    constexpr int LPC = KROW / 16;                           // lanes per cell in the score pass
    constexpr int KP = C * LPC / NT;                         // 16-byte K pieces per thread
    static_assert(C * LPC % NT == 0 && KP >= 1, "chunk size");
    constexpr int DQ = D / 4, NCG = AO_NCG, CPG = C / NCG;   // P.V pass: 32 lanes of 4 dims, 16 cell groups of CPG cells
    const int tid = tid_now(), lane = tid & 63;
    const int n_ctx = a.n_ctx;
    ld.chunk = sp; ld.skip = false;
    if (a.tok_chunks) {                                      // (C == 64 only: the lists count 64-cell chunks)
        if (sp >= a.tok_nchunks[0]) { ld.skip = true; return; }
        ld.chunk = a.tok_chunks[sp];
    }
    const int c_lo = ld.chunk * C;
    const size_t head_row0 = (size_t)g * n_ctx;

    // ---- every global load of the item
    ld.cpos = -1;
    ld.cseq = 0;
    if (tid < C && c_lo + tid < n_ctx) { ld.cpos = a.cell_pos[c_lo + tid]; ld.cseq = a.cell_seq[c_lo + tid]; }
    constexpr int HP = D / 2, NPAIR = R * HP;
    static_assert(NPAIR <= NT, "query pairs per thread");
    ld.qv = f32x2_t{0.0f, 0.0f}; ld.csv = f32x2_t{1.0f, 0.0f};
    if (tid < NPAIR) {
        const int r = tid / HP, i = tid % HP;
        if (!QF) ld.qv = *reinterpret_cast<const f32x2_t *>(a.q + ((size_t)g * R + r) * D + 2 * i);
        if (2 * i < n_rot) ld.csv = *reinterpret_cast<const f32x2_t *>(cs_table + 2 * i);
    }
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + NT * j;
        int cell = c_lo + p / LPC;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TK == T_F16) ld.kreg[j] = *reinterpret_cast<const u32x4_t *>(reinterpret_cast<const uint16_t *>(a.kv.k) + rowi * D + (p % LPC) * 8);
        else ld.kreg[j] = *reinterpret_cast<const u32x4_t *>(a.kv.k + rowi * KROW + (p % LPC) * 16);
    }
    const int dq = tid % DQ, cg = tid / DQ;
#pragma unroll
    for (int i = 0; i < CPG; i++) {
        int cell = c_lo + cg + NCG * i;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TV == T_F16) ld.vreg[i] = *reinterpret_cast<const u32x2_t *>(reinterpret_cast<const uint16_t *>(a.kv.v) + rowi * D + dq * 4);
        else { ld.vreg[i].x = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + dq * 4); ld.vreg[i].y = 0; }
    }
    ld.ks2 = 0; ld.vs2 = 0;
    if (tid < C * NB / 2) {
        int cell = c_lo + tid / (NB / 2);
        if (cell >= n_ctx) cell = n_ctx - 1;
        if (TK != T_F16) ld.ks2 = *reinterpret_cast<const uint32_t *>(a.kv.kd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
        if (TV != T_F16) ld.vs2 = *reinterpret_cast<const uint32_t *>(a.kv.vd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
    }

    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 8] = wall_clock64();
    // the token's un-rotated K / V row of this kv head: 1 KB, requested by wave 0 of EVERY item now (which chunk holds the token's cell is only known once
    // the scalars are back; waiting for them first cost that chunk - every kv head's slowest item - a second memory round trip)
    // (every wave, unconditionally: a load under a condition makes hipcc wait for ALL outstanding loads where the branches join)
    // (native vector types: HIP's float4 class, modified under a condition below, is placed in scratch by hipcc)
    {
        const bool isk = lane < 32;
        const int dd = (lane & 31) * 4;
        if (!QF) ld.xn4 = *reinterpret_cast<const f32x4_t *>((isk ? fz.knew : fz.vnew) + g * D + dd);
        else ld.xn4 = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        ld.cs4 = *reinterpret_cast<const f32x4_t *>(cs_table + (dd < n_rot ? dd : n_rot - 4));   // c0 s0 c1 s1 (clamped: used only where dd < n_rot)
    }
}

// ---- one attention item (kv head g, chunk slot sp) on the 512 threads of the workgroup; every exit is workgroup-uniform.  `dma` (the request for the
// workgroup's W_o rows; it does nothing after its first call) is called by every wave once the item has nothing outstanding and nobody waits for it.
// the step's scalars every item needs, read in one batch of scalar loads UNDER the item's vector loads
struct AOScalars { unsigned serial; int tpos, tseq, cellnew; };
__device__ __forceinline__ AOScalars ao_load_scalars(const unsigned *serial, const int32_t *tpos, const int32_t *tseq, const int32_t *cell) {
    unsigned s0; int s1, s2, s3;
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %5, 0x0\n\ts_load_dword %2, %6, 0x0\n\ts_load_dword %3, %7, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3) : "s"(serial), "s"(tpos), "s"(tseq), "s"(cell) : "memory");
    return AOScalars{s0, s1, s2, s3};
}
// ---- one attention item of the launch WITHOUT Q | K | V (attn_out_kernel), as round 4 wrote it: requests and their uses in one straight-line function.  (The
// split form below - ao_item_issue / ao_item_run, which the fused launch needs - compiled 16 more registers and a scratch frame into this kernel and cost
// Llama-2-7B's steps 3.4 % on the same box, tools/ab_libs.sh; the two forms hold the same arithmetic.)
// (kv head g, chunk slot sp) on the 512 threads of the workgroup; every exit is workgroup-uniform.  `dma` (the request for the
// workgroup's W_o rows; it does nothing after its first call) is called by every wave once the item has nothing outstanding and nobody waits for it.
// the step's scalars every item needs, read in one batch of scalar loads UNDER the item's vector loads
template <int R, int TK, int TV, int C, class Dma>
__device__ __forceinline__ void ao_attn_item(const AttnArgs &a, const float *cs_table, int n_rot, const DecodeFuse &fz, const AOArgs &o,
                                             int g, int sp, AOSmem<R, C> &sm, Dma dma) {
    constexpr int D = AO_D, NB = AO_NB, NT = AO_NT;
    constexpr int KROW = TK == T_F16 ? 2 * D : D;
    constexpr int LPC = KROW / 16;                           // lanes per cell in the score pass
    constexpr int KP = C * LPC / NT;                         // 16-byte K pieces per thread
    static_assert(C * LPC % NT == 0 && KP >= 1, "chunk size");
    constexpr int DQ = D / 4, NCG = AO_NCG, CPG = C / NCG;   // P.V pass: 32 lanes of 4 dims, 16 cell groups of CPG cells
    constexpr int CL = C / 64;                               // cells per lane in the softmax
    const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
    const int n_ctx = a.n_ctx;
    int chunk = sp;
    if (a.tok_chunks) {                                      // (C == 64 only: the lists count 64-cell chunks)
        if (sp >= a.tok_nchunks[0]) { dma(); return; }
        chunk = a.tok_chunks[sp];
    }
    const int c_lo = chunk * C;
    const size_t head_row0 = (size_t)g * n_ctx;

    // ---- every global load of the item
    int cpos = -1;
    unsigned long long cseq = 0;
    if (tid < C && c_lo + tid < n_ctx) { cpos = a.cell_pos[c_lo + tid]; cseq = a.cell_seq[c_lo + tid]; }
    constexpr int HP = D / 2, NPAIR = R * HP;
    static_assert(NPAIR <= NT, "query pairs per thread");
    f32x2_t qv = {0.0f, 0.0f}, csv = {1.0f, 0.0f};
    if (tid < NPAIR) {
        const int r = tid / HP, i = tid % HP;
        qv = *reinterpret_cast<const f32x2_t *>(a.q + ((size_t)g * R + r) * D + 2 * i);
        if (2 * i < n_rot) csv = *reinterpret_cast<const f32x2_t *>(cs_table + 2 * i);
    }
    u32x4_t kreg[KP];
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + NT * j;
        int cell = c_lo + p / LPC;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TK == T_F16) kreg[j] = *reinterpret_cast<const u32x4_t *>(reinterpret_cast<const uint16_t *>(a.kv.k) + rowi * D + (p % LPC) * 8);
        else kreg[j] = *reinterpret_cast<const u32x4_t *>(a.kv.k + rowi * KROW + (p % LPC) * 16);
    }
    const int dq = tid % DQ, cg = tid / DQ;
    u32x2_t vreg[CPG];
#pragma unroll
    for (int i = 0; i < CPG; i++) {
        int cell = c_lo + cg + NCG * i;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TV == T_F16) vreg[i] = *reinterpret_cast<const u32x2_t *>(reinterpret_cast<const uint16_t *>(a.kv.v) + rowi * D + dq * 4);
        else { vreg[i].x = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + dq * 4); vreg[i].y = 0; }
    }
    uint32_t ks2 = 0, vs2 = 0;
    if (tid < C * NB / 2) {
        int cell = c_lo + tid / (NB / 2);
        if (cell >= n_ctx) cell = n_ctx - 1;
        if (TK != T_F16) ks2 = *reinterpret_cast<const uint32_t *>(a.kv.kd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
        if (TV != T_F16) vs2 = *reinterpret_cast<const uint32_t *>(a.kv.vd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
    }

    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 8] = wall_clock64();
    // the token's un-rotated K / V row of this kv head: 1 KB, requested by wave 0 of EVERY item now (which chunk holds the token's cell is only known once
    // the scalars are back; waiting for them first cost that chunk - every kv head's slowest item - a second memory round trip)
    // (every wave, unconditionally: a load under a condition makes hipcc wait for ALL outstanding loads where the branches join)
    // (native vector types: HIP's float4 class, modified under a condition below, is placed in scratch by hipcc)
    f32x4_t xn4, cs4;
    {
        const bool isk = lane < 32;
        const int dd = (lane & 31) * 4;
        xn4 = *reinterpret_cast<const f32x4_t *>((isk ? fz.knew : fz.vnew) + g * D + dd);
        cs4 = *reinterpret_cast<const f32x4_t *>(cs_table + (dd < n_rot ? dd : n_rot - 4));   // c0 s0 c1 s1 (clamped: used only where dd < n_rot)
    }
    // ---- the step's scalars (one batch of scalar loads, under the vector loads above)
    const AOScalars sc = ao_load_scalars(o.serial, a.tok_pos, a.tok_seq, fz.tok_cell);
    const unsigned serial = sc.serial;
    const int tpos = sc.tpos, tseq = sc.tseq;
    // ---- this token's own K / V row (the chunk that holds its cell): rotate K, convert, write the cache row, keep the codes for the patch below
    int own_cl = -1;
    {
        const int cellnew = sc.cellnew;
        if (cellnew >= c_lo && cellnew < c_lo + C) own_cl = cellnew - c_lo;
        if (own_cl >= 0 && wave == 0) {                          // lanes 0 .. 31: K, lanes 32 .. 63: V; four elements each
            const bool isk = lane < 32;
            const int dd = (lane & 31) * 4;
            float xa[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
            if (isk && dd < n_rot) {
                const float x0 = xn4.x, x1 = xn4.y, x2 = xn4.z, x3 = xn4.w;
                xa[0] = x0 * cs4.x - x1 * cs4.y; xa[1] = x0 * cs4.y + x1 * cs4.x;
                xa[2] = x2 * cs4.z - x3 * cs4.w; xa[3] = x2 * cs4.w + x3 * cs4.z;
            }
            const size_t rowi = head_row0 + cellnew;
            const int TT = isk ? TK : TV;
            // (the cache planes through opaque copies: from `isk ? a.kv.k : a.kv.v` on the kernel arguments hipcc builds a two-entry pointer table in scratch)
            uint8_t *kvk = a.kv.k, *kvv = a.kv.v;
            asm volatile("" : "+s"(kvk), "+s"(kvv));
            uint32_t packed = 0; float dsc = 0.0f;
            if (TK != T_F16 || TV != T_F16) wave_quant_q80(xa, packed, dsc);   // whole wave takes part (8-lane groups)
            if (TT == T_F16) {
                u32x2_t ov; ov.x = (uint32_t)f2h(xa[0]) | ((uint32_t)f2h(xa[1]) << 16); ov.y = (uint32_t)f2h(xa[2]) | ((uint32_t)f2h(xa[3]) << 16);
                *reinterpret_cast<u32x2_t *>((isk ? sm.newk : sm.newv) + dd * 2) = ov;
                *reinterpret_cast<u32x2_t *>(reinterpret_cast<uint16_t *>(isk ? kvk : kvv) + rowi * D + dd) = ov;
            } else {
                *reinterpret_cast<uint32_t *>((isk ? sm.newk : sm.newv) + dd) = packed;
                *reinterpret_cast<uint32_t *>((isk ? kvk : kvv) + rowi * D + dd) = packed;
                if ((lane & 7) == 0) {
                    const uint16_t hd = f2h(dsc);
                    (isk ? a.kv.kd : a.kv.vd)[rowi * NB + (dd >> 5)] = hd;
                    reinterpret_cast<uint16_t *>(isk ? sm.newkd : sm.newvd)[dd >> 5] = hd;
                }
            }
        }
    }

    // ---- q: rotate, convert
    if (tid < C) sm.vis[tid] = (cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
    if (tid < C * NB / 2) { sm.ksc[tid] = ks2; sm.vsc[tid] = vs2; }
    if (tid < NPAIR) {
        const float x0 = qv.x, x1 = qv.y, c = csv.x, s = csv.y;
        float y0 = x0 * c - x1 * s, y1 = x0 * s + x1 * c;
        if (TK == T_F16) { y0 = h2f(f2h(y0)); y1 = h2f(f2h(y1)); }
        { const f32x2_t yy = {y0, y1}; *reinterpret_cast<f32x2_t *>(sm.qf + 2 * tid) = yy; }
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 9] = wall_clock64();
    if (own_cl >= 0) {   // workgroup-uniform: the row just produced instead of what the cache held before
        if (tid < NB / 2) {
            if (TK != T_F16) sm.ksc[own_cl * (NB / 2) + tid] = sm.newkd[tid];
            if (TV != T_F16) sm.vsc[own_cl * (NB / 2) + tid] = sm.newvd[tid];
        }
#pragma unroll
        for (int j = 0; j < KP; j++) {
            const int p = tid + NT * j;
            if (p / LPC == own_cl) kreg[j] = *reinterpret_cast<const u32x4_t *>(sm.newk + (p % LPC) * 16);
        }
#pragma unroll
        for (int i = 0; i < CPG; i++) {
            if (cg + NCG * i == own_cl) {
                if (TV == T_F16) vreg[i] = *reinterpret_cast<const u32x2_t *>(sm.newv + dq * 8);
                else { vreg[i].x = *reinterpret_cast<const uint32_t *>(sm.newv + dq * 4); vreg[i].y = 0; }
            }
        }
    }
    if (TK != T_F16) {   // q8_0 of the rotated q: 4 values per thread, 8-lane groups (whole waves: R * D / 4 is a multiple of 64)
        if (tid * 4 < R * D) {
            const f32x4_t v4 = *reinterpret_cast<const f32x4_t *>(sm.qf + tid * 4);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(sm.qc + tid * 4) = packed;
            if ((lane & 7) == 0) sm.qd[(tid * 4) >> 5] = h2f(f2h(d));
        }
        __syncthreads();
    }

    // ---- scores
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + NT * j;
        const int cl = p / LPC, piece = p % LPC;
        float sc[R];
        if (TK == T_F16) {
            const uint32_t kw[4] = {kreg[j].x, kreg[j].y, kreg[j].z, kreg[j].w};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float *qq = sm.qf + r * D + piece * 8;
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    s += h2f((uint16_t)(kw[i] & 0xffff)) * qq[2 * i];
                    s += h2f((uint16_t)(kw[i] >> 16)) * qq[2 * i + 1];
                }
                s += dpp_f<DPP_QP_1032>(s); s += dpp_f<DPP_QP_2301>(s); s += dpp_f<DPP_HALF_MIRROR>(s); s += dpp_f<DPP_MIRROR>(s);      // 16 lanes
                sc[r] = s;
            }
        } else {
            const uint32_t kpair = sm.ksc[cl * (NB / 2) + (piece >> 2)];
            const float dk = h2f((uint16_t)(((piece >> 1) & 1) ? (kpair >> 16) : (kpair & 0xffff)));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32x4_t qq = *reinterpret_cast<const u32x4_t *>(sm.qc + r * D + piece * 16);
                int s = 0;
                s = dot4(kreg[j].x, qq.x, s); s = dot4(kreg[j].y, qq.y, s); s = dot4(kreg[j].z, qq.z, s); s = dot4(kreg[j].w, qq.w, s);
                s += dpp_i<DPP_QP_1032>(s);                    // both halves of the 32-block (integer)
                float f = (piece & 1) ? 0.0f : (float)s * (dk * sm.qd[r * NB + (piece >> 1)]);
                f += dpp_f<DPP_QP_1032>(f); f += dpp_f<DPP_QP_2301>(f);                                 // 4 lanes
                f += dpp_f<DPP_HALF_MIRROR>(f);                                                          // 8 lanes
                sc[r] = f;
            }
        }
        if (piece == 0) {
            const bool v = sm.vis[cl] != 0;
#pragma unroll
            for (int r = 0; r < R; r++) sm.S[r * C + cl] = v ? sc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 10] = wall_clock64();

    // ---- softmax of the chunk: wave w -> heads w, w + 8, ..; lane = cell (and cell + 64)
    for (int r = wave; r < R; r += AO_NW) {
        float s[CL], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < CL; k++) { s[k] = sm.S[r * C + 64 * k + lane]; m = fmaxf(m, s[k]); }
        m = wave_max(m);
        float l = 0.0f;
#pragma unroll
        for (int k = 0; k < CL; k++) {
            const float p = (s[k] == -INFINITY) ? 0.0f : expf(s[k] - m);
            sm.S[r * C + 64 * k + lane] = p;
            l += p;
        }
        l = wave_sum(l);
        if (lane == 0) { sm.ml[2 * r] = m; sm.ml[2 * r + 1] = l; }
    }
    __syncthreads();

    // ---- P.V
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
#pragma unroll
    for (int i = 0; i < CPG; i++) {
        const int cl = cg + NCG * i;
        float v4[4];
        if (TV == T_F16) {
            v4[0] = h2f((uint16_t)(vreg[i].x & 0xffff)); v4[1] = h2f((uint16_t)(vreg[i].x >> 16));
            v4[2] = h2f((uint16_t)(vreg[i].y & 0xffff)); v4[3] = h2f((uint16_t)(vreg[i].y >> 16));
        } else {
            const uint32_t vpair = sm.vsc[cl * (NB / 2) + (dq >> 4)];
            const float dv = h2f((uint16_t)(((dq >> 3) & 1) ? (vpair >> 16) : (vpair & 0xffff)));
            const uint32_t w = vreg[i].x;
            v4[0] = (float)(int8_t)(w & 0xff) * dv; v4[1] = (float)(int8_t)((w >> 8) & 0xff) * dv;
            v4[2] = (float)(int8_t)((w >> 16) & 0xff) * dv; v4[3] = (float)(int8_t)(w >> 24) * dv;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float p = sm.S[r * C + cl];
            acc[r][0] += v4[0] * p; acc[r][1] += v4[1] * p; acc[r][2] += v4[2] * p; acc[r][3] += v4[3] * p;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        { const f32x4_t av = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]}; *reinterpret_cast<f32x4_t *>(sm.accs + ((size_t)cg * R + r) * D + dq * 4) = av; }
    __syncthreads();
    // ---- the chunk's partial record per head: AO_REC floats = O [D] | m | l | 0 | 0, written through to the coherence point in 16-byte pieces (a 4-byte
    // write-through store is one fabric write each: 520 of them per item sat in the drain in front of the ticket)
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 11] = wall_clock64();
    {
        const __amdgpu_buffer_rsrc_t prs = coh_rsrc(a.part);
        if (tid * 4 < R * D) {
            const int r = (tid * 4) / D, d = tid * 4 - r * D;
            f32x4_t sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NCG; j++) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t *>(sm.accs + ((size_t)j * R + r) * D + d);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const int off = (int)(((((size_t)g * R + r) * a.splits + sp) * AO_REC + d) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(coh_u32x4, sum), prs, off, 0, 16);
            if (d == 0) {
                const f32x4_t mlv = {sm.ml[2 * r], sm.ml[2 * r + 1], 0.0f, 0.0f};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(coh_u32x4, mlv), prs, off + D * 4, 0, 16);
            }
        }
    }

    // ---- ticket; the last workgroup of this kv head (group) merges: GP kv heads per ticket, RM = GP * R heads merged (attn_decode_dev.h)
    constexpr int GP = 256 / ((R * D) % 256 == 0 ? 256 : 128);
    constexpr int RM = R * GP;
    const int gq = g / GP;
    const int hb = gq * RM;
    const int stride_s = a.splits;
    const int splits = a.tok_nchunks ? a.tok_nchunks[0] : a.splits;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through partial stores have left
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 12] = wall_clock64();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(fz.counters + gq * ATT_SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm.last_flag = (old == (unsigned)(splits * GP) - 1u) ? 1 : 0;
        if (sm.last_flag) __hip_atomic_store(fz.counters + gq * ATT_SYNC_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    // the W_o rows may queue now: nothing of this item is outstanding any more and nobody waits for this workgroup (36 KB of DMA in front of the partial
    // stores would have sat in their drain, i.e. in every kv head's ticket); the merging workgroup requests them after it has raised its flag
    if (!sm.last_flag) { dma(); return; }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 2] = wall_clock64();
    // The merge: out[h][d] = sum over the chunks of w[h][chunk] * O[h][chunk][d], w = exp(m - M) / sum(exp(m - M) * l) (flash_attn_combine_kernel's
    // weights).  The RM * D outputs are NDG groups of four dims; the 512 threads are NPH phases x NDG groups, phase p takes the chunks p, p + NPH, ..:
    // at most 8 (16 from 33 chunks on) 16-byte loads per thread, all requested before the (m, l) pairs so that the weights are computed under them; the
    // phases' partial sums meet in LDS and are added in phase order (a fixed order: the same bits on every run).
    constexpr int NDG = RM * D / 4, NPH = NT / NDG < 4 ? NT / NDG : 4, UB = 8;     // (at most four phases: threads beyond them only help with the weights)
    static_assert(NDG % 64 == 0 && NDG * NPH <= NT, "merge phases");
    float *psum = sm.accs;                             // [NPH][RM * D]
    float *merged = sm.accs + NPH * RM * D;            // [RM * D]
    static_assert((NPH + 1) * RM * D + (GP > 1 ? RM * 64 : 0) <= AO_NCG * R * D, "merge scratch");
    static_assert(RM * 64 <= R * C || GP > 1, "weights scratch");
    const int dg = tid % NDG, ph = tid / NDG;          // (ph is the same for all lanes of a wave: NDG is a multiple of 64)
    const int mr = (dg * 4) / D, md = dg * 4 - mr * D;
    const __amdgpu_buffer_rsrc_t prs = coh_rsrc(a.part);
    const int rec0 = (int)(((((size_t)hb + mr) * stride_s) * AO_REC + md) * 4);
    coh_u32x4 x[UB];
    auto request = [&](int s0) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s2 = s0 + ph + NPH * u;
            if (ph < NPH && s2 < splits) x[u] = __builtin_amdgcn_raw_buffer_load_b128(prs, rec0 + s2 * (AO_REC * 4), 0, 16);     // (wave-uniform condition)
        }
    };
    request(0);
    float *wg2 = GP > 1 ? sm.accs + (NPH + 1) * RM * D : sm.S;     // [RM][64] chunk weights (pairs of kv heads: S is too small for two heads' weights)
    for (int r = wave; r < RM; r += AO_NW) {
        float m = -INFINITY, l = 0.0f;
        if (lane < splits) {
            const coh_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(prs, (int)(((((size_t)hb + r) * stride_s + lane) * AO_REC + D) * 4), 0, 16);
            m = __uint_as_float(v.x); l = __uint_as_float(v.y);
        }
        const float M = wave_max(m);
        const float w = (lane < splits && m != -INFINITY) ? expf(m - M) : 0.0f;
        const float den = wave_sum(w * l);
        const float inv = 1.0f / den;
        wg2[r * 64 + lane] = w * inv;
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 13] = wall_clock64();
    f32x4_t macc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s2 = s0 + ph + NPH * u;
            if (ph < NPH && s2 < splits) {
                const float w = wg2[mr * 64 + s2];
                const f32x4_t xv = __builtin_bit_cast(f32x4_t, x[u]);
                macc.x += w * xv.x; macc.y += w * xv.y; macc.z += w * xv.z; macc.w += w * xv.w;
            }
        }
        s0 += NPH * UB;
        if (s0 >= splits) break;
        request(s0);
    }
    if (ph < NPH) *reinterpret_cast<f32x4_t *>(psum + (size_t)ph * RM * D + dg * 4) = macc;      // (the item's P.V sums in accs were consumed before the ticket)
    __syncthreads();
    for (int e = tid; e < RM * D; e += NT) {
        float v = 0.0f;
#pragma unroll
        for (int p2 = 0; p2 < NPH; p2++) v += psum[(size_t)p2 * RM * D + e];
        merged[e] = v;
    }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 14] = wall_clock64();
    __syncthreads();                                   // (no global store is outstanding here: a barrier behind stores waits for them)
    // The flag goes up NOW, before the codes exist: it only says "start looking".  The codes travel as 8-byte granules {tag, value} - data and validity in one
    // store, nothing to fence or drain - and every consumer sweeps them until each carries this step's tag: its first sweep is in flight while the codes
    // are being written.
    const unsigned tag = serial * 256u + (unsigned)o.layer + 1u;
    if (tid == 0) __hip_atomic_store(o.flags + gq * ATT_SYNC_STRIDE, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int NBLK = (RM * D) >> 8;                // 256-blocks this ticket group owns in the H * D row
    const __amdgpu_buffer_rsrc_t grs = coh_rsrc(o.gran);
    for (int b = wave; b < NBLK; b += AO_NW) {
        const f32x4_t v4 = *reinterpret_cast<const f32x4_t *>(merged + b * 256 + lane * 4);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const int gb = ((hb * D) >> 8) + b;            // global block index
        uint32_t packed; int bs; float dq8;
        wave_quant_q8k(vv, lane, packed, bs, dq8);
        // two granules per 16-byte write-through store (each 8-byte half carries its own tag): the even lane takes its neighbour's word
        const uint32_t packed1 = (uint32_t)dpp_i<DPP_QP_1032>((int)packed);
        // a block's granules: [64 x four codes | 16 block sums | scale, 15 unused] = AO_GPB granules = six 128-byte lines that no other merger writes
        if ((lane & 1) == 0) {
            const coh_u32x4 g2 = {packed, tag, packed1, tag};
            __builtin_amdgcn_raw_buffer_store_b128(g2, grs, (gb * AO_GPB + lane) * 8, 0, 16);
        }
        const int bs1 = dpp_i<0x104>(bs);             // row_shl:4 - the block sum four lanes further on (the next 16-code group)
        if ((lane & 7) == 0) {
            const coh_u32x4 g2 = {(unsigned)bs & 0xffffu, tag, (unsigned)bs1 & 0xffffu, tag};
            __builtin_amdgcn_raw_buffer_store_b128(g2, grs, (gb * AO_GPB + 64 + (lane >> 2)) * 8, 0, 16);
        }
        if (lane == 0) st_store_granule(o.gran + gb * AO_GPB + 80, tag, __float_as_uint(dq8));
    }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 3] = wall_clock64();
    for (int e = tid; e < RM * D; e += NT) a.out[(size_t)hb * D + e] = merged[e];   // (the f32 rows: nobody in this launch reads them)
    dma();
}

// qf (QF only): the launch's own Q | K | V granules and the step's scalars, read at the top of the launch
template <int R, int TK, int TV, int C, bool QF, class Dma>
__device__ __forceinline__ void ao_item_run(const AttnArgs &a, int n_rot, const DecodeFuse &fz, const AOArgs &o, const QFArgs &f, const AOScalars *pre,
                                            int g, int sp, AOSmemQ<R, C> &sm, float *accs, AOItemLd<R, TK, TV, C> &ld, Dma dma) {
    constexpr int D = AO_D, NB = AO_NB, NT = AO_NT;
    constexpr int KROW = TK == T_F16 ? 2 * D : D;
    constexpr int LPC = KROW / 16;                           // lanes per cell in the score pass
    constexpr int KP = C * LPC / NT;                         // 16-byte K pieces per thread
    constexpr int DQ = D / 4, NCG = AO_NCG, CPG = C / NCG;   // P.V pass: 32 lanes of 4 dims, 16 cell groups of CPG cells
    constexpr int CL = C / 64;                               // cells per lane in the softmax
    constexpr int HP = D / 2, NPAIR = R * HP;
    const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
    const int n_ctx = a.n_ctx;
    if (ld.skip) { dma(); return; }
    const int c_lo = ld.chunk * C;
    const size_t head_row0 = (size_t)g * n_ctx;
    const int cpos = ld.cpos;
    const unsigned long long cseq = ld.cseq;
    const uint32_t ks2 = ld.ks2, vs2 = ld.vs2;
    const f32x2_t csv = ld.csv;
    const f32x4_t cs4 = ld.cs4;
    u32x4_t kreg[KP];
    u32x2_t vreg[CPG];
#pragma unroll
    for (int j = 0; j < KP; j++) kreg[j] = ld.kreg[j];
#pragma unroll
    for (int i = 0; i < CPG; i++) vreg[i] = ld.vreg[i];
    const int dq = tid % DQ, cg = tid / DQ;
    // ---- the step's scalars (one batch of scalar loads, under the vector loads above; QF: read at the top of the launch)
    const AOScalars sc = QF ? *pre : ao_load_scalars(o.serial, a.tok_pos, a.tok_seq, fz.tok_cell);
    const unsigned serial = sc.serial;
    const int tpos = sc.tpos, tseq = sc.tseq;
    f32x2_t qv = ld.qv;
    f32x4_t xn4 = ld.xn4;
    if constexpr (QF) {
        // q of this kv head's R query heads and (wave 0) the token's K / V row of this kv head, as the projections' workgroups published them: tagged
        // granules, swept until every one carries this step's and this layer's tag (data and validity in one 8-byte store: nothing to fence)
        const unsigned tag = serial * 256u + (unsigned)o.layer + 1u;
        const __amdgpu_buffer_rsrc_t qrs = coh_rsrc(f.gran);
        const bool isk = lane < 32;
        const int dd = (lane & 31) * 4;
        const int qoff = (int)((((size_t)g * R + (tid < NPAIR ? tid / HP : 0)) * D + 2 * (tid % HP)) * 8);
        const int koff = (int)(((size_t)f.n_q + (isk ? 0 : (size_t)f.n_kv) + (size_t)g * D + dd) * 8);
        coh_u32x4 qg = {0u, tag, 0u, tag}, k0 = {0u, tag, 0u, tag}, k1 = {0u, tag, 0u, tag};
        int spins = 0;
        for (;;) {
            if (tid < NPAIR) qg = __builtin_amdgcn_raw_buffer_load_b128(qrs, qoff, 0, 16);                 // (wave-uniform: NPAIR is a multiple of 64)
            if (wave == 0) { k0 = __builtin_amdgcn_raw_buffer_load_b128(qrs, koff, 0, 16); k1 = __builtin_amdgcn_raw_buffer_load_b128(qrs, koff + 16, 0, 16); }
            const bool ok = qg.y == tag && qg.w == tag && k0.y == tag && k0.w == tag && k1.y == tag && k1.w == tag;
            if (__all(ok)) break;
            if (++spins >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_GATHER); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        qv = f32x2_t{__uint_as_float(qg.x), __uint_as_float(qg.z)};
        xn4 = f32x4_t{__uint_as_float(k0.x), __uint_as_float(k0.z), __uint_as_float(k1.x), __uint_as_float(k1.z)};
        if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 15] = wall_clock64();
    }
    // ---- this token's own K / V row (the chunk that holds its cell): rotate K, convert, write the cache row, keep the codes for the patch below
    int own_cl = -1;
    {
        const int cellnew = sc.cellnew;
        if (cellnew >= c_lo && cellnew < c_lo + C) own_cl = cellnew - c_lo;
        if (own_cl >= 0 && wave == 0) {                          // lanes 0 .. 31: K, lanes 32 .. 63: V; four elements each
            const bool isk = lane < 32;
            const int dd = (lane & 31) * 4;
            float xa[4] = {xn4.x, xn4.y, xn4.z, xn4.w};
            if (isk && dd < n_rot) {
                const float x0 = xn4.x, x1 = xn4.y, x2 = xn4.z, x3 = xn4.w;
                xa[0] = x0 * cs4.x - x1 * cs4.y; xa[1] = x0 * cs4.y + x1 * cs4.x;
                xa[2] = x2 * cs4.z - x3 * cs4.w; xa[3] = x2 * cs4.w + x3 * cs4.z;
            }
            const size_t rowi = head_row0 + cellnew;
            const int TT = isk ? TK : TV;
            // (the cache planes through opaque copies: from `isk ? a.kv.k : a.kv.v` on the kernel arguments hipcc builds a two-entry pointer table in scratch)
            uint8_t *kvk = a.kv.k, *kvv = a.kv.v;
            asm volatile("" : "+s"(kvk), "+s"(kvv));
            uint32_t packed = 0; float dsc = 0.0f;
            if (TK != T_F16 || TV != T_F16) wave_quant_q80(xa, packed, dsc);   // whole wave takes part (8-lane groups)
            if (TT == T_F16) {
                u32x2_t ov; ov.x = (uint32_t)f2h(xa[0]) | ((uint32_t)f2h(xa[1]) << 16); ov.y = (uint32_t)f2h(xa[2]) | ((uint32_t)f2h(xa[3]) << 16);
                *reinterpret_cast<u32x2_t *>((isk ? sm.newk : sm.newv) + dd * 2) = ov;
                *reinterpret_cast<u32x2_t *>(reinterpret_cast<uint16_t *>(isk ? kvk : kvv) + rowi * D + dd) = ov;
            } else {
                *reinterpret_cast<uint32_t *>((isk ? sm.newk : sm.newv) + dd) = packed;
                *reinterpret_cast<uint32_t *>((isk ? kvk : kvv) + rowi * D + dd) = packed;
                if ((lane & 7) == 0) {
                    const uint16_t hd = f2h(dsc);
                    (isk ? a.kv.kd : a.kv.vd)[rowi * NB + (dd >> 5)] = hd;
                    reinterpret_cast<uint16_t *>(isk ? sm.newkd : sm.newvd)[dd >> 5] = hd;
                }
            }
        }
    }

    // ---- q: rotate, convert
    if (tid < C) sm.vis[tid] = (cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
    if (tid < C * NB / 2) { sm.ksc[tid] = ks2; sm.vsc[tid] = vs2; }
    if (tid < NPAIR) {
        const float x0 = qv.x, x1 = qv.y, c = csv.x, s = csv.y;
        float y0 = x0 * c - x1 * s, y1 = x0 * s + x1 * c;
        if (TK == T_F16) { y0 = h2f(f2h(y0)); y1 = h2f(f2h(y1)); }
        { const f32x2_t yy = {y0, y1}; *reinterpret_cast<f32x2_t *>(sm.qf + 2 * tid) = yy; }
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 9] = wall_clock64();
    if (own_cl >= 0) {   // workgroup-uniform: the row just produced instead of what the cache held before
        if (tid < NB / 2) {
            if (TK != T_F16) sm.ksc[own_cl * (NB / 2) + tid] = sm.newkd[tid];
            if (TV != T_F16) sm.vsc[own_cl * (NB / 2) + tid] = sm.newvd[tid];
        }
#pragma unroll
        for (int j = 0; j < KP; j++) {
            const int p = tid + NT * j;
            if (p / LPC == own_cl) kreg[j] = *reinterpret_cast<const u32x4_t *>(sm.newk + (p % LPC) * 16);
        }
#pragma unroll
        for (int i = 0; i < CPG; i++) {
            if (cg + NCG * i == own_cl) {
                if (TV == T_F16) vreg[i] = *reinterpret_cast<const u32x2_t *>(sm.newv + dq * 8);
                else { vreg[i].x = *reinterpret_cast<const uint32_t *>(sm.newv + dq * 4); vreg[i].y = 0; }
            }
        }
    }
    if (TK != T_F16) {   // q8_0 of the rotated q: 4 values per thread, 8-lane groups (whole waves: R * D / 4 is a multiple of 64)
        if (tid * 4 < R * D) {
            const f32x4_t v4 = *reinterpret_cast<const f32x4_t *>(sm.qf + tid * 4);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(sm.qc + tid * 4) = packed;
            if ((lane & 7) == 0) sm.qd[(tid * 4) >> 5] = h2f(f2h(d));
        }
        __syncthreads();
    }

    // ---- scores
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + NT * j;
        const int cl = p / LPC, piece = p % LPC;
        float sc[R];
        if (TK == T_F16) {
            const uint32_t kw[4] = {kreg[j].x, kreg[j].y, kreg[j].z, kreg[j].w};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float *qq = sm.qf + r * D + piece * 8;
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    s += h2f((uint16_t)(kw[i] & 0xffff)) * qq[2 * i];
                    s += h2f((uint16_t)(kw[i] >> 16)) * qq[2 * i + 1];
                }
                s += dpp_f<DPP_QP_1032>(s); s += dpp_f<DPP_QP_2301>(s); s += dpp_f<DPP_HALF_MIRROR>(s); s += dpp_f<DPP_MIRROR>(s);      // 16 lanes
                sc[r] = s;
            }
        } else {
            const uint32_t kpair = sm.ksc[cl * (NB / 2) + (piece >> 2)];
            const float dk = h2f((uint16_t)(((piece >> 1) & 1) ? (kpair >> 16) : (kpair & 0xffff)));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const u32x4_t qq = *reinterpret_cast<const u32x4_t *>(sm.qc + r * D + piece * 16);
                int s = 0;
                s = dot4(kreg[j].x, qq.x, s); s = dot4(kreg[j].y, qq.y, s); s = dot4(kreg[j].z, qq.z, s); s = dot4(kreg[j].w, qq.w, s);
                s += dpp_i<DPP_QP_1032>(s);                    // both halves of the 32-block (integer)
                float f = (piece & 1) ? 0.0f : (float)s * (dk * sm.qd[r * NB + (piece >> 1)]);
                f += dpp_f<DPP_QP_1032>(f); f += dpp_f<DPP_QP_2301>(f);                                 // 4 lanes
                f += dpp_f<DPP_HALF_MIRROR>(f);                                                          // 8 lanes
                sc[r] = f;
            }
        }
        if (piece == 0) {
            const bool v = sm.vis[cl] != 0;
#pragma unroll
            for (int r = 0; r < R; r++) sm.S[r * C + cl] = v ? sc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 10] = wall_clock64();

    // ---- softmax of the chunk: wave w -> heads w, w + 8, ..; lane = cell (and cell + 64)
    for (int r = wave; r < R; r += AO_NW) {
        float s[CL], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < CL; k++) { s[k] = sm.S[r * C + 64 * k + lane]; m = fmaxf(m, s[k]); }
        m = wave_max(m);
        float l = 0.0f;
#pragma unroll
        for (int k = 0; k < CL; k++) {
            const float p = (s[k] == -INFINITY) ? 0.0f : expf(s[k] - m);
            sm.S[r * C + 64 * k + lane] = p;
            l += p;
        }
        l = wave_sum(l);
        if (lane == 0) { sm.ml[2 * r] = m; sm.ml[2 * r + 1] = l; }
    }
    __syncthreads();

    // ---- P.V
    float acc[R][4];
#pragma unroll
    for (int r = 0; r < R; r++) { acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.0f; }
#pragma unroll
    for (int i = 0; i < CPG; i++) {
        const int cl = cg + NCG * i;
        float v4[4];
        if (TV == T_F16) {
            v4[0] = h2f((uint16_t)(vreg[i].x & 0xffff)); v4[1] = h2f((uint16_t)(vreg[i].x >> 16));
            v4[2] = h2f((uint16_t)(vreg[i].y & 0xffff)); v4[3] = h2f((uint16_t)(vreg[i].y >> 16));
        } else {
            const uint32_t vpair = sm.vsc[cl * (NB / 2) + (dq >> 4)];
            const float dv = h2f((uint16_t)(((dq >> 3) & 1) ? (vpair >> 16) : (vpair & 0xffff)));
            const uint32_t w = vreg[i].x;
            v4[0] = (float)(int8_t)(w & 0xff) * dv; v4[1] = (float)(int8_t)((w >> 8) & 0xff) * dv;
            v4[2] = (float)(int8_t)((w >> 16) & 0xff) * dv; v4[3] = (float)(int8_t)(w >> 24) * dv;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float p = sm.S[r * C + cl];
            acc[r][0] += v4[0] * p; acc[r][1] += v4[1] * p; acc[r][2] += v4[2] * p; acc[r][3] += v4[3] * p;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        { const f32x4_t av = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]}; *reinterpret_cast<f32x4_t *>(accs + ((size_t)cg * R + r) * D + dq * 4) = av; }
    __syncthreads();
    // ---- the chunk's partial record per head: AO_REC floats = O [D] | m | l | 0 | 0, written through to the coherence point in 16-byte pieces (a 4-byte
    // write-through store is one fabric write each: 520 of them per item sat in the drain in front of the ticket)
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 11] = wall_clock64();
    {
        const __amdgpu_buffer_rsrc_t prs = coh_rsrc(a.part);
        if (tid * 4 < R * D) {
            const int r = (tid * 4) / D, d = tid * 4 - r * D;
            f32x4_t sum = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < NCG; j++) {
                const f32x4_t v = *reinterpret_cast<const f32x4_t *>(accs + ((size_t)j * R + r) * D + d);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
            const int off = (int)(((((size_t)g * R + r) * a.splits + sp) * AO_REC + d) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(coh_u32x4, sum), prs, off, 0, 16);
            if (d == 0) {
                const f32x4_t mlv = {sm.ml[2 * r], sm.ml[2 * r + 1], 0.0f, 0.0f};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(coh_u32x4, mlv), prs, off + D * 4, 0, 16);
            }
        }
    }

    // ---- ticket; the last workgroup of this kv head (group) merges: GP kv heads per ticket, RM = GP * R heads merged (attn_decode_dev.h)
    constexpr int GP = 256 / ((R * D) % 256 == 0 ? 256 : 128);
    constexpr int RM = R * GP;
    const int gq = g / GP;
    const int hb = gq * RM;
    const int stride_s = a.splits;
    const int splits = a.tok_nchunks ? a.tok_nchunks[0] : a.splits;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through partial stores have left
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 12] = wall_clock64();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(fz.counters + gq * ATT_SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm.last_flag = (old == (unsigned)(splits * GP) - 1u) ? 1 : 0;
        if (sm.last_flag) __hip_atomic_store(fz.counters + gq * ATT_SYNC_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    // the W_o rows may queue now: nothing of this item is outstanding any more and nobody waits for this workgroup (36 KB of DMA in front of the partial
    // stores would have sat in their drain, i.e. in every kv head's ticket); the merging workgroup requests them after it has raised its flag
    if (!sm.last_flag) { dma(); return; }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 2] = wall_clock64();
    // The merge: out[h][d] = sum over the chunks of w[h][chunk] * O[h][chunk][d], w = exp(m - M) / sum(exp(m - M) * l) (flash_attn_combine_kernel's
    // weights).  The RM * D outputs are NDG groups of four dims; the 512 threads are NPH phases x NDG groups, phase p takes the chunks p, p + NPH, ..:
    // at most 8 (16 from 33 chunks on) 16-byte loads per thread, all requested before the (m, l) pairs so that the weights are computed under them; the
    // phases' partial sums meet in LDS and are added in phase order (a fixed order: the same bits on every run).
    constexpr int NDG = RM * D / 4, NPH = NT / NDG < 4 ? NT / NDG : 4, UB = 8;     // (at most four phases: threads beyond them only help with the weights)
    static_assert(NDG % 64 == 0 && NDG * NPH <= NT, "merge phases");
    float *psum = accs;                             // [NPH][RM * D]
    float *merged = accs + NPH * RM * D;            // [RM * D]
    static_assert((NPH + 1) * RM * D + (GP > 1 ? RM * 64 : 0) <= AO_NCG * R * D, "merge scratch");
    static_assert(RM * 64 <= R * C || GP > 1, "weights scratch");
    const int dg = tid % NDG, ph = tid / NDG;          // (ph is the same for all lanes of a wave: NDG is a multiple of 64)
    const int mr = (dg * 4) / D, md = dg * 4 - mr * D;
    const __amdgpu_buffer_rsrc_t prs = coh_rsrc(a.part);
    const int rec0 = (int)(((((size_t)hb + mr) * stride_s) * AO_REC + md) * 4);
    coh_u32x4 x[UB];
    auto request = [&](int s0) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s2 = s0 + ph + NPH * u;
            if (ph < NPH && s2 < splits) x[u] = __builtin_amdgcn_raw_buffer_load_b128(prs, rec0 + s2 * (AO_REC * 4), 0, 16);     // (wave-uniform condition)
        }
    };
    request(0);
    float *wg2 = GP > 1 ? accs + (NPH + 1) * RM * D : sm.S;     // [RM][64] chunk weights (pairs of kv heads: S is too small for two heads' weights)
    for (int r = wave; r < RM; r += AO_NW) {
        float m = -INFINITY, l = 0.0f;
        if (lane < splits) {
            const coh_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(prs, (int)(((((size_t)hb + r) * stride_s + lane) * AO_REC + D) * 4), 0, 16);
            m = __uint_as_float(v.x); l = __uint_as_float(v.y);
        }
        const float M = wave_max(m);
        const float w = (lane < splits && m != -INFINITY) ? expf(m - M) : 0.0f;
        const float den = wave_sum(w * l);
        const float inv = 1.0f / den;
        wg2[r * 64 + lane] = w * inv;
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 13] = wall_clock64();
    f32x4_t macc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int s0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s2 = s0 + ph + NPH * u;
            if (ph < NPH && s2 < splits) {
                const float w = wg2[mr * 64 + s2];
                const f32x4_t xv = __builtin_bit_cast(f32x4_t, x[u]);
                macc.x += w * xv.x; macc.y += w * xv.y; macc.z += w * xv.z; macc.w += w * xv.w;
            }
        }
        s0 += NPH * UB;
        if (s0 >= splits) break;
        request(s0);
    }
    if (ph < NPH) *reinterpret_cast<f32x4_t *>(psum + (size_t)ph * RM * D + dg * 4) = macc;      // (the item's P.V sums in accs were consumed before the ticket)
    __syncthreads();
    for (int e = tid; e < RM * D; e += NT) {
        float v = 0.0f;
#pragma unroll
        for (int p2 = 0; p2 < NPH; p2++) v += psum[(size_t)p2 * RM * D + e];
        merged[e] = v;
    }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 14] = wall_clock64();
    __syncthreads();                                   // (no global store is outstanding here: a barrier behind stores waits for them)
    // The flag goes up NOW, before the codes exist: it only says "start looking".  The codes travel as 8-byte granules {tag, value} - data and validity in one
    // store, nothing to fence or drain - and every consumer sweeps them until each carries this step's tag: its first sweep is in flight while the codes
    // are being written.
    const unsigned tag = serial * 256u + (unsigned)o.layer + 1u;
    if (tid == 0) __hip_atomic_store(o.flags + gq * ATT_SYNC_STRIDE, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int NBLK = (RM * D) >> 8;                // 256-blocks this ticket group owns in the H * D row
    const __amdgpu_buffer_rsrc_t grs = coh_rsrc(o.gran);
    for (int b = wave; b < NBLK; b += AO_NW) {
        const f32x4_t v4 = *reinterpret_cast<const f32x4_t *>(merged + b * 256 + lane * 4);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
        const int gb = ((hb * D) >> 8) + b;            // global block index
        uint32_t packed; int bs; float dq8;
        wave_quant_q8k(vv, lane, packed, bs, dq8);
        // two granules per 16-byte write-through store (each 8-byte half carries its own tag): the even lane takes its neighbour's word
        const uint32_t packed1 = (uint32_t)dpp_i<DPP_QP_1032>((int)packed);
        // a block's granules: [64 x four codes | 16 block sums | scale, 15 unused] = AO_GPB granules = six 128-byte lines that no other merger writes
        if ((lane & 1) == 0) {
            const coh_u32x4 g2 = {packed, tag, packed1, tag};
            __builtin_amdgcn_raw_buffer_store_b128(g2, grs, (gb * AO_GPB + lane) * 8, 0, 16);
        }
        const int bs1 = dpp_i<0x104>(bs);             // row_shl:4 - the block sum four lanes further on (the next 16-code group)
        if ((lane & 7) == 0) {
            const coh_u32x4 g2 = {(unsigned)bs & 0xffffu, tag, (unsigned)bs1 & 0xffffu, tag};
            __builtin_amdgcn_raw_buffer_store_b128(g2, grs, (gb * AO_GPB + 64 + (lane >> 2)) * 8, 0, 16);
        }
        if (lane == 0) st_store_granule(o.gran + gb * AO_GPB + 80, tag, __float_as_uint(dq8));
    }
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 3] = wall_clock64();
    for (int e = tid; e < RM * D; e += NT) a.out[(size_t)hb * D + e] = merged[e];   // (the f32 rows: nobody in this launch reads them)
    dma();
}

// ---- the mat-vec part: wave w decodes the row pairs w, w + 8, .. of the workgroup's rows out of LDS (consumer_op's arithmetic: passes in order, each
// row's lane partials summed by wave_sum, residual + value)
template <int TYPE>
__device__ __forceinline__ void ao_decode(const AOArgs &o, const uint8_t *slice, const ActL &AL, int b0, int nrw, int wave, int lane, float rs0, float rs1) {
    using Rw = Raw<TYPE>;
    const LaneRole L = make_role<TYPE>(lane);
    constexpr int SBP = role_sbp<TYPE>();
    const int nb = o.K >> 8, NP = (nb + SBP - 1) / SBP;           // passes of the wave over a row (lane role of the type: 8 or 16 super-blocks each)
    const unsigned rb = o.row_bytes;
    const int n_steps = (nrw + 1) >> 1;
    for (int s = wave; s < n_steps; s += AO_NW) {
        const bool two = 2 * s + 1 < nrw;
        const unsigned off0 = 2u * (unsigned)s * rb, off1 = two ? off0 + rb : off0;
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 2
        for (int p = 0; p < NP; p++) {
            const ActSlice sl = read_slice_t<TYPE>(AL, p * SBP + L.sbl, nb, L);
            int sb = p * SBP + L.sbl;
            if (sb >= nb) sb = nb - 1;
            Rw w0, w1;
            ring_load<false>(w0, slice, 0u, off0, nb, sb, L);
            ring_load<false>(w1, slice, 0u, off1, nb, sb, L);
            acc0 += w0.dot(sl, L);
            acc1 += w1.dot(sl, L);
        }
        const float v0 = wave_sum(acc0), v1 = wave_sum(acc1);
        if (lane == 0) {
            const int row0 = b0 + 2 * s;
            const bool first = s == wave;
            if (o.epi == EPI_ADD) {
                o.out[row0] = (first ? rs0 : o.resid[row0]) + v0;
                if (two) o.out[row0 + 1] = (first ? rs1 : o.resid[row0 + 1]) + v1;
            } else {
                o.out[row0] = v0;
                if (two) o.out[row0 + 1] = v1;
            }
        }
    }
}

// ================================================================= QF: the layer's Q | K | V inside the same launch (round 6)
// The per-launch path ran [Q | K | V: weight stream, 6.8 us for 14.7 MB] -> boundary -> [this kernel: 13 us] per layer, and the second launch's first 2 us are
// spent waiting for its own K / V cache rows, which do not depend on the first launch at all.  Both launches' weights fit in LDS at once (Q | K | V: ~58 KB per
// workgroup, W_o: 37 KB), so: ten waves per workgroup; waves 8 and 9 are LOADERS (mmvq_stream_dev.h's division of labour: only they ever sit in the memory queue) -
// they copy the workgroup's Q | K | V rows and then its W_o rows HBM -> LDS by DMA, publish what has landed and END (a barrier only counts live waves); waves 0 - 7:
//   1. request the layer input and the norm weights in the kernel's first instructions (early_issue, from preloaded arguments), then the step's scalars; behind those
//      requests the loaders are let go ([SY_QF_GO]: a small request queued behind the stream waits for all of it);
//   2. RMSNorm * w -> Q8_K exactly as the weight stream's prologue does (consumer_prologue: same code, same bits), decode the workgroup's rows out of LDS with the
//      stream's decoders as the slots land, and publish every result as ONE 8-byte granule {tag, f32 bits} written through to the coherence point;
//   3. item workgroups request their chunk's K / V rows, scales and cell metadata (behind the projections: in front of them those requests - megabytes with the context
//      filled - hold the prologue and the stream back), sweep the granules of their kv head (q: 2 KB, the K / V row: 1 KB) until each carries the tag and carry on as
//      before; W_o is on chip long before the merged codes arrive.
// One launch, one boundary and one ramp less per layer.  Outputs are bit-identical to the two launches.
struct QFRun { const uint8_t *W; unsigned rb, total; int type, b0, nrw, grow0, ns; };     // grow0: granule index of row b0; ns: 4 KiB slots of the run
__device__ __forceinline__ QFRun qf_setup(QFArgs f) {
    // (through opaque copies: from adjacent kernel arguments selected by one index hipcc builds a table in scratch memory)
    asm volatile("" : "+s"(f.rb0), "+s"(f.rb1), "+s"(f.rb2), "+s"(f.t0), "+s"(f.t1), "+s"(f.t2), "+s"(f.n0), "+s"(f.n1), "+s"(f.n2));
    const int b = (int)blockIdx.x;
    const int s = b >= f.blk2 ? 2 : b >= f.blk1 ? 1 : 0;
    const int lo = s == 0 ? 0 : s == 1 ? f.blk1 : f.blk2, hi = s == 0 ? f.blk1 : s == 1 ? f.blk2 : f.blk3;
    QFRun q;
    q.W = s == 0 ? f.W0 : s == 1 ? f.W1 : f.W2;
    q.rb = s == 0 ? f.rb0 : s == 1 ? f.rb1 : f.rb2;
    q.type = s == 0 ? f.t0 : s == 1 ? f.t1 : f.t2;
    const int n_rows = s == 0 ? f.n0 : s == 1 ? f.n1 : f.n2;
    const int nblk = hi - lo, bl = b - lo;
    int b0 = 0, b1 = 0;
    if (nblk > 0 && b < f.blk3) {                                 // (op_setup's deal of a tensor's rows over its workgroups)
        const int rpb = (n_rows + nblk - 1) / nblk;
        b0 = bl * rpb; b1 = b0 + rpb;
        if (b0 > n_rows) b0 = n_rows;
        if (b1 > n_rows) b1 = n_rows;
    }
    q.b0 = b0; q.nrw = b1 - b0;
    q.grow0 = (s == 0 ? 0 : s == 1 ? f.n_q : f.n_q + f.n_kv) + b0;
    q.total = (unsigned)q.nrw * q.rb;
    q.ns = (int)((q.total + ST_SLOT - 1) / ST_SLOT);
    return q;
}
struct QFWoRun { int b0, nrw; const uint8_t *W; unsigned rb; };     // the workgroup's run of W_o rows
// a loader wave (lq = 0, 1): global slot j of the workgroup = slot j of its Q | K | V run, then the slots of its W_o run; loader lq copies the slots j = lq mod 2
// and publishes how many of ITS slots have landed ([SY_LANDED + lq]; they land in order), never more than 15 slots = 60 DMA instructions in flight: what the wave's
// vector-memory counter can count.
__device__ __forceinline__ void qf_loader(const QFRun &q, const QFWoRun &wo, uint8_t *smem, unsigned lds_q, unsigned lds_o, int lq, int lane) {
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    ST_SPIN_WHILE(ld_sync(sy + SY_QF_GO) < AO_NW, 0);              // the consumers' activation requests are queued: a small request behind the stream waits for all of it
    const unsigned total_o = wo.nrw > 0 ? (unsigned)wo.nrw * wo.rb : 0u;
    const int nso = (int)((total_o + ST_SLOT - 1) / ST_SLOT);
    const uint8_t *src_q = q.W + (size_t)q.b0 * q.rb, *src_o = wo.W + (size_t)wo.b0 * wo.rb;
    int issued = 0, published = 0;
    auto publish = [&]() {
        const int landed = (ST_SI * issued - vm_outstanding()) / ST_SI;
        if (landed > published) { published = landed; st_sync(sy + SY_LANDED + lq, landed); }
    };
    for (int j = lq; j < q.ns + nso; j += 2) {
        // at most 15 slots = 60 DMA instructions of this wave in flight: what its vector-memory counter can count (a run of more slots waits for landings here)
        int polls = 0;
        while (issued - published >= 15) {
            if (++polls >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_LOADER); return; }
            publish(); __builtin_amdgcn_s_sleep(1);
        }
        const bool isq = j < q.ns;
        const uint8_t *src = isq ? src_q : src_o;
        const unsigned total = isq ? q.total : total_o, last = total - 16u;
        const unsigned off = (unsigned)(isq ? j : j - q.ns) * ST_SLOT;
        const unsigned dst = (isq ? lds_q : lds_o) + off;
        if (off + ST_SLOT <= total) dma_slot(src + off + (size_t)lane * 16, dst);
        else {
#pragma unroll
            for (int p = 0; p < ST_SI; p++) {
                const unsigned x = off + (unsigned)lane * 16u + p * 1024u;       // past the end: every such lane re-reads the run's last 16 B
                dma16<true>(src + (x < total ? x : last), dst + p * 1024);
            }
        }
        issued++;
        publish();
    }
    int polls = 0;
    while (published < issued) {
        if (++polls >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_DRAIN); break; }
        publish(); __builtin_amdgcn_s_sleep(1);
    }
}
// waves 0 - 7: row pairs w, w + 8, .. of the workgroup's Q | K | V rows out of LDS (consumer_op's arithmetic, as ao_decode below), each result one granule.
// (Measured, same box: a contiguous run of three rows per wave decoded side by side - equal work for every wave - is 0.6 us SLOWER: the phase ends one decode
// step behind the landing of the run's last slot, and a pair is the shortest step.)
template <int TYPE, int KB>
__device__ __forceinline__ void qf_decode(const QFRun &q, const QFArgs &f, const uint8_t *slice, const ActL &AL, int *sy, unsigned tag, int wave, int lane) {
    using Rw = Raw<TYPE>;
    const LaneRole L = make_role<TYPE>(lane);
    constexpr int SBP = role_sbp<TYPE>(), NP = role_passes<TYPE, KB>();   // (K <= 4096: one pass of the wide lane roles, two of the others; a partial last pass decodes zero-scale slices)
    static_assert(NP <= 2, "activation slices in registers");
    const int nb = f.K >> 8;
    const unsigned rb = q.rb;
    const int n_steps = (q.nrw + 1) >> 1;
    const ActSlice S0 = read_slice_t<TYPE>(AL, L.sbl, nb, L), S1 = NP > 1 ? read_slice_t<TYPE>(AL, SBP + L.sbl, nb, L) : S0;
    for (int s = wave; s < n_steps; s += AO_NW) {
        const bool two = 2 * s + 1 < q.nrw;
        const unsigned off0 = 2u * (unsigned)s * rb, off1 = two ? off0 + rb : off0;
        const unsigned n = (off0 + (two ? 2u : 1u) * rb + ST_SLOT - 1) / ST_SLOT;       // slots [0, n) of the run hold the step's rows
        const int need0 = (int)((n + 1u) >> 1), need1 = (int)(n >> 1);
        ST_SPIN_WHILE(ld_sync(sy + SY_LANDED) < need0 || ld_sync(sy + SY_LANDED + 1) < need1, 1);
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const ActSlice sl = p == 0 ? S0 : S1;
            int sb = p * SBP + L.sbl;
            if (sb >= nb) sb = nb - 1;
            Rw w0, w1;
            ring_load<false>(w0, slice, 0u, off0, nb, sb, L);
            ring_load<false>(w1, slice, 0u, off1, nb, sb, L);
            acc0 += w0.dot(sl, L);
            acc1 += w1.dot(sl, L);
        }
        const float v0 = wave_sum(acc0), v1 = wave_sum(acc1);
        if (lane == 0) {
            const int gr = q.grow0 + 2 * s;
            st_store_granule(f.gran + gr, tag, __float_as_uint(v0));
            if (two) st_store_granule(f.gran + gr + 1, tag, __float_as_uint(v1));
        }
    }
}

// ea (QF): the layer input and the norm weights, requested in the kernel's first instructions from PRELOADED arguments (qkv_attn_out_kernel)
template <int R, int TK, int TV, int C, int KB>
__device__ __forceinline__ void ao_body(const AttnArgs &a, const float *cs_table, int n_rot, const DecodeFuse &fz, const AOArgs &o, const QFArgs &f,
                                        const EarlyAct<(KB > 0 ? KB : 1)> *eap = nullptr) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr bool QF = KB > 0;
    const int tid = tid_now(), lane = tid & 63, wave = uni(tid >> 6);
    if constexpr (QF) sync_init(smem);                 // (all ten waves)
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 0] = wall_clock64();
    const AOLayout lay = ao_layout(o.slice_lds, o.K, sizeof(AOSmemQ<R, C>), QF ? AO_QF_HDR : 0u, QF ? f.qkv_lds : 0u);
    AOSmemQ<R, C> &sm = *reinterpret_cast<AOSmemQ<R, C> *>(smem + lay.attn);
    // the P.V / merge scratch: in the space of the workgroup's Q | K | V rows where it fits there (they are dead once decoded; the first write is behind the item's
    // first workgroup barrier, which every wave reaches after its last row), behind everything else otherwise (f.accs_off: the launcher's choice)
    float *accs = reinterpret_cast<float *>(smem + f.accs_off);
    const int b0r = (int)blockIdx.x * o.rows_per_wg;
    const int b0 = b0r < o.n_rows ? b0r : o.n_rows;
    const int nrw = b0 + o.rows_per_wg <= o.n_rows ? o.rows_per_wg : o.n_rows - b0;
    QFRun qr{};
    if constexpr (QF) {
        qr = qf_setup(f);
        if (wave >= AO_NW) {
            const QFWoRun wr{b0, nrw, o.W, o.row_bytes};
            qf_loader(qr, wr, smem, lds_addr(smem + lay.qkv), lds_addr(smem + lay.wo), wave - AO_NW, lane);
            return;
        }
    }
    // residual of this wave's first row pair: requested now, used at the very end
    float rs0 = 0.0f, rs1 = 0.0f;
    if (o.epi == EPI_ADD && 2 * wave < nrw) {
        rs0 = o.resid[b0 + 2 * wave];
        if (2 * wave + 1 < nrw) rs1 = o.resid[b0 + 2 * wave + 1];
    }
    bool dma_done = QF;                                // (QF: the loaders bring the W_o rows)
    auto dma = [&]() { if (!dma_done) { ao_issue_dma(o, smem + lay.wo, b0, nrw, wave, lane); dma_done = true; } };
    const int G = a.G;
    int *sy = reinterpret_cast<int *>(smem + ST_OFF_SYNC);
    AOScalars scal{};
    bool had_item = false;
    int it0 = (int)blockIdx.x;
    if constexpr (QF) {
        static_assert(ST_NC == AO_NW, "the weight stream's prologue is cut for eight consumer waves");
#ifndef MI355_QF_LATE_GO
        // the loaders go NOW: only the activation's requests must be ahead of the stream
        asm volatile("" ::: "memory");
        if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_QF_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        // (an item's cache rows are requested BEHIND the projections, as part of the item: requested here, to wait in registers for q, they sit in the memory queue in
        // front of the projections' stream and hold the prologue and the rows back - activation ready 2.8 -> 4.6 us with the context filled (256 items x 34 KB);
        // same box: 622 -> 645 tok/s filled, 683 -> 687 at pos 512)
        scal = ao_load_scalars(o.serial, a.tok_pos, a.tok_seq, fz.tok_cell);
        StOp pa{};
        pa.K = f.K; pa.neps = f.neps; pa.nx = f.nx; pa.nw = f.nw;
        const StLayout stl{(int)lay.qs, (int)lay.d, (int)lay.bs, 0};
#ifdef MI355_QF_LATE_GO
        if (lane == 0) (void)__hip_atomic_fetch_add(sy + SY_QF_GO, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
        // (the activation in the format THIS workgroup's tensor contracts with: Q8_K planes, or Q8_0 codes + block scales for a Q8_0 tensor; workgroup-uniform)
        if (act_is_q80(qr.type)) consumer_prologue<KB, 1, true, false, true>(pa, smem, stl, wave, lane, EngIO(), eap);
        else consumer_prologue<KB, 1, false, false, true>(pa, smem, stl, wave, lane, EngIO(), eap);
        if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 7] = wall_clock64();
        if (qr.nrw > 0) {
            const unsigned tag = scal.serial * 256u + (unsigned)o.layer + 1u;
            const ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
            switch (qr.type) {
                case T_Q4_K: qf_decode<T_Q4_K, KB>(qr, f, smem + lay.qkv, AL, sy, tag, wave, lane); break;
                case T_Q5_K: qf_decode<T_Q5_K, KB>(qr, f, smem + lay.qkv, AL, sy, tag, wave, lane); break;
                case T_Q6_K: qf_decode<T_Q6_K, KB>(qr, f, smem + lay.qkv, AL, sy, tag, wave, lane); break;
                case T_Q8_0: qf_decode<T_Q8_0, KB>(qr, f, smem + lay.qkv, AL, sy, tag, wave, lane); break;
                default: break;
            }
        }
    }
    for (int it = it0; it < o.n_items; it += (int)gridDim.x) {
        if (had_item) __syncthreads();                 // (the item's LDS is reused)
        AOItemLd<R, TK, TV, C> ld;
        ao_item_issue<R, TK, TV, C, QF>(a, cs_table, n_rot, fz, o, it % G, it / G, ld);
        ao_item_run<R, TK, TV, C, QF>(a, n_rot, fz, o, f, QF ? &scal : nullptr, it % G, it / G, sm, accs, ld, dma);
        had_item = true;
    }
    dma();                                             // workgroups without an item: at once
    const unsigned serial = QF ? scal.serial : ao_load_scalars(o.serial, a.tok_pos, a.tok_seq, fz.tok_cell).serial;     // (a second batch for workgroups that had an item: scalar-cache hits)
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 1] = wall_clock64();

    // ---- the merged, quantised attention output: wait until every ticket group's merge has its sums (one wave polls, relaxed, bounded), then every wave
    // sweeps its share of the granules until each carries this step's tag and puts the values where the decoders expect the Q8_K planes
    const unsigned tag = serial * 256u + (unsigned)o.layer + 1u;
    // (ONE polling wave per workgroup: four of them, a quarter of a round trip apart, noticed the flags earlier and still cost 1.3 % of the step - 1024
    // pollers on eight words slow the mergers' own traffic down; same-box A/B, tools/ab_libs.sh)
    if (wave == 0) {
        int spins = 0;
        for (;;) {
            unsigned v = tag;
            if (lane < o.n_flags) v = __hip_atomic_load(o.flags + lane * ATT_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v == tag)) break;
            if (++spins >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_GATHER); break; }
            __builtin_amdgcn_s_sleep(MI355_AO_POLL_SLEEP);
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 4] = wall_clock64();
    {
        constexpr int PPB = 41;                        // granule pairs a block holds (32 of codes, 8 of block sums, 1 = scale + an unused word)
        const int nb = o.K >> 8, np = nb * PPB;
        constexpr int PPT = (32 * PPB + AO_NT - 1) / AO_NT;      // pairs per thread at the longest K the launch takes (8192)
        const __amdgpu_buffer_rsrc_t grs = coh_rsrc(o.gran);
        coh_u32x4 val[PPT];
        int blk[PPT], w[PPT];
#pragma unroll
        for (int j = 0; j < PPT; j++) { const int i = tid + AO_NT * j; blk[j] = i / PPB; w[j] = i - blk[j] * PPB; }
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                if (tid + AO_NT * j < np) {
                    val[j] = __builtin_amdgcn_raw_buffer_load_b128(grs, (blk[j] * AO_GPB + 2 * w[j]) * 8, 0, 16);
                    ok = ok && val[j].y == tag && (val[j].w == tag || w[j] == PPB - 1);
                }
            }
            if (__all(ok)) break;
            if (++spins >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_GATHER); break; }
            __builtin_amdgcn_s_sleep(MI355_AO_SWEEP_SLEEP);
        }
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            if (tid + AO_NT * j >= np) continue;
            if (w[j] < 32) { unsigned *q = reinterpret_cast<unsigned *>(smem + lay.qs) + blk[j] * 64 + 2 * w[j]; q[0] = val[j].x; q[1] = val[j].z; }
            else if (w[j] < 40) reinterpret_cast<unsigned *>(smem + lay.bs)[blk[j] * 8 + (w[j] - 32)] = (val[j].x & 0xffffu) | (val[j].z << 16);
            else reinterpret_cast<unsigned *>(smem + lay.d)[blk[j]] = val[j].x;
        }
    }
    if constexpr (QF) {                                // every slot of the workgroup (the W_o rows are the last ones) has landed
        const int nt = qr.ns + (int)(((nrw > 0 ? (unsigned)nrw * o.row_bytes : 0u) + ST_SLOT - 1) / ST_SLOT);
        ST_SPIN_WHILE(ld_sync(sy + SY_LANDED) < ((nt + 1) >> 1) || ld_sync(sy + SY_LANDED + 1) < (nt >> 1), 1);
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the W_o rows (DMA) has landed
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 5] = wall_clock64();
    if (nrw > 0) {
        const ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
        switch (o.type) {
            case T_Q4_K: ao_decode<T_Q4_K>(o, smem + lay.wo, AL, b0, nrw, wave, lane, rs0, rs1); break;
            case T_Q5_K: ao_decode<T_Q5_K>(o, smem + lay.wo, AL, b0, nrw, wave, lane, rs0, rs1); break;
            case T_Q6_K: ao_decode<T_Q6_K>(o, smem + lay.wo, AL, b0, nrw, wave, lane, rs0, rs1); break;
            default: break;
        }
    }
    if (o.probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) o.probe[(size_t)blockIdx.x * 16 + 6] = wall_clock64();
    }
}

// The kernel argument segment is cold at every launch and hipcc reads a field where it is first used: seven dependent scalar-load round trips sat in front
// of the first K / V request (1.4 us from entry to "loads issued", MI355_AO_PROBE).  One independent read per 64-byte line of the segment, all in flight
// together, makes the later field reads scalar-cache hits.
template <size_t BYTES> __device__ __forceinline__ void ao_touch_kernargs() {
    const __attribute__((address_space(4))) unsigned *ka = (const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < (int)((BYTES + 63) / 64); i++) acc ^= ka[i * 16];
    asm volatile("" :: "s"(acc));
}
template <int R, int TK, int TV, int C>
__global__ __launch_bounds__(AO_NT) void attn_out_kernel(const AttnArgs a, const float *cs_table, int n_rot, const DecodeFuse fz, const AOArgs o) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // The kernel argument segment is cold at every launch and hipcc reads a field where it is first used: seven dependent scalar-load round trips sat in front
    // of the first K / V request (1.4 us from entry to "loads issued", MI355_AO_PROBE).  One independent read per 64-byte line of the segment, all in flight
    // together, makes the later field reads scalar-cache hits.
    {
        struct KArgs { AttnArgs a; const float *cs; int n_rot; DecodeFuse fz; AOArgs o; };
        const __attribute__((address_space(4))) unsigned *ka = (const __attribute__((address_space(4))) unsigned *)__builtin_amdgcn_kernarg_segment_ptr();
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < (int)((sizeof(KArgs) + 63) / 64); i++) acc ^= ka[i * 16];
        asm volatile("" :: "s"(acc));
    }
    const int tid = tid_now(), lane = tid & 63, wave = uni(tid >> 6);
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 0] = wall_clock64();
    const AOLayout lay = ao_layout(o.slice_lds, o.K, sizeof(AOSmem<R, C>));
    AOSmem<R, C> &sm = *reinterpret_cast<AOSmem<R, C> *>(smem + lay.attn);
    const int b0r = (int)blockIdx.x * o.rows_per_wg;
    const int b0 = b0r < o.n_rows ? b0r : o.n_rows;
    const int nrw = b0 + o.rows_per_wg <= o.n_rows ? o.rows_per_wg : o.n_rows - b0;
    // residual of this wave's first row pair: requested now, used at the very end
    float rs0 = 0.0f, rs1 = 0.0f;
    if (o.epi == EPI_ADD && 2 * wave < nrw) {
        rs0 = o.resid[b0 + 2 * wave];
        if (2 * wave + 1 < nrw) rs1 = o.resid[b0 + 2 * wave + 1];
    }
    bool dma_done = false;
    auto dma = [&]() { if (!dma_done) { ao_issue_dma(o, smem, b0, nrw, wave, lane); dma_done = true; } };
    const int G = a.G;
    bool had_item = false;
    for (int it = (int)blockIdx.x; it < o.n_items; it += (int)gridDim.x) {
        if (had_item) __syncthreads();                 // (the item's LDS is reused)
        ao_attn_item<R, TK, TV, C>(a, cs_table, n_rot, fz, o, it % G, it / G, sm, dma);
        had_item = true;
    }
    dma();                                             // workgroups without an item: at once
    const unsigned serial = ao_load_scalars(o.serial, a.tok_pos, a.tok_seq, fz.tok_cell).serial;     // (a second batch for workgroups that had an item: scalar-cache hits)
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 1] = wall_clock64();

    // ---- the merged, quantised attention output: wait until every ticket group's merge has its sums (one wave polls, relaxed, bounded), then every wave
    // sweeps its share of the granules until each carries this step's tag and puts the values where the decoders expect the Q8_K planes
    const unsigned tag = serial * 256u + (unsigned)o.layer + 1u;
    // (ONE polling wave per workgroup: four of them, a quarter of a round trip apart, noticed the flags earlier and still cost 1.3 % of the step - 1024
    // pollers on eight words slow the mergers' own traffic down; same-box A/B, tools/ab_libs.sh)
    if (wave == 0) {
        int spins = 0;
        for (;;) {
            unsigned v = tag;
            if (lane < o.n_flags) v = __hip_atomic_load(o.flags + lane * ATT_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v == tag)) break;
            if (++spins >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_GATHER); break; }
            __builtin_amdgcn_s_sleep(MI355_AO_POLL_SLEEP);
        }
        asm volatile("" ::: "memory");
    }
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 4] = wall_clock64();
    {
        constexpr int PPB = 41;                        // granule pairs a block holds (32 of codes, 8 of block sums, 1 = scale + an unused word)
        const int nb = o.K >> 8, np = nb * PPB;
        constexpr int PPT = (32 * PPB + AO_NT - 1) / AO_NT;      // pairs per thread at the longest K the launch takes (8192)
        const __amdgpu_buffer_rsrc_t grs = coh_rsrc(o.gran);
        coh_u32x4 val[PPT];
        int blk[PPT], w[PPT];
#pragma unroll
        for (int j = 0; j < PPT; j++) { const int i = tid + AO_NT * j; blk[j] = i / PPB; w[j] = i - blk[j] * PPB; }
        int spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < PPT; j++) {
                if (tid + AO_NT * j < np) {
                    val[j] = __builtin_amdgcn_raw_buffer_load_b128(grs, (blk[j] * AO_GPB + 2 * w[j]) * 8, 0, 16);
                    ok = ok && val[j].y == tag && (val[j].w == tag || w[j] == PPB - 1);
                }
            }
            if (__all(ok)) break;
            if (++spins >= ST_SPIN_LIMIT) { st_timeout(ST_ERR_GATHER); break; }
            __builtin_amdgcn_s_sleep(MI355_AO_SWEEP_SLEEP);
        }
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            if (tid + AO_NT * j >= np) continue;
            if (w[j] < 32) { unsigned *q = reinterpret_cast<unsigned *>(smem + lay.qs) + blk[j] * 64 + 2 * w[j]; q[0] = val[j].x; q[1] = val[j].z; }
            else if (w[j] < 40) reinterpret_cast<unsigned *>(smem + lay.bs)[blk[j] * 8 + (w[j] - 32)] = (val[j].x & 0xffffu) | (val[j].z << 16);
            else reinterpret_cast<unsigned *>(smem + lay.d)[blk[j]] = val[j].x;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the W_o rows (DMA) has landed
    __syncthreads();
    if (o.probe && tid == 0) o.probe[(size_t)blockIdx.x * 16 + 5] = wall_clock64();
    if (nrw > 0) {
        const ActL AL{reinterpret_cast<const int8_t *>(smem + lay.qs), reinterpret_cast<const float *>(smem + lay.d), reinterpret_cast<const int16_t *>(smem + lay.bs)};
        switch (o.type) {
            case T_Q4_K: ao_decode<T_Q4_K>(o, smem, AL, b0, nrw, wave, lane, rs0, rs1); break;
            case T_Q5_K: ao_decode<T_Q5_K>(o, smem, AL, b0, nrw, wave, lane, rs0, rs1); break;
            case T_Q6_K: ao_decode<T_Q6_K>(o, smem, AL, b0, nrw, wave, lane, rs0, rs1); break;
            default: break;
        }
    }
    if (o.probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) o.probe[(size_t)blockIdx.x * 16 + 6] = wall_clock64();
    }
}

constexpr int AO_QF_NT = AO_NT + 128;                   // QF: two loader waves behind the eight
// The first three arguments repeat f.nx, f.nw, f.K: the first dwords of a kernel's argument segment arrive in SGPRs with the wave (build.py: -amdgpu-kernarg-preload-count
// for this file), so the activation's requests - the head of the launch's dependent chain - are queued in the kernel's first instructions, before the cold argument
// segment has answered its first read (0.45 us).  (The loaders started from preloaded, packed geometry as well - tools/r6_qf_fast_start.patch - lose 1.4 %: the launch is
// bound by its prologue, and an earlier stream delays the consumers' small requests.)
template <int R, int TK, int TV, int C, int KB>
__global__ __launch_bounds__(AO_QF_NT) void qkv_attn_out_kernel(const float *p_nx, const float *p_nw, int p_K, const AttnArgs a, const float *cs_table, int n_rot, const DecodeFuse fz,
                                                                const AOArgs o, const QFArgs f) {
    EarlyAct<KB> ea;
    {
        const int tid = tid_now(), lane = tid & 63, wave = uni(tid >> 6);
        early_issue<KB, 1>(p_nx, p_nw, p_K, wave < AO_NW ? wave : 0, lane, ea);      // (the loaders' copies are never looked at: a request under a condition makes hipcc wait where the branches join)
    }
    struct KArgs { const float *nx, *nw; int K; AttnArgs a; const float *cs; int n_rot; DecodeFuse fz; AOArgs o; QFArgs f; };
    ao_touch_kernargs<sizeof(KArgs)>();
    ao_body<R, TK, TV, C, KB>(a, cs_table, n_rot, fz, o, f, &ea);
}

int g_attn_out_fused = -1;                  // -1: environment / default (on)
unsigned long long *g_ao_probe = nullptr;
int g_ao_probe_wgs = 0, g_ao_probe_items = 0;

}  // namespace

void set_attn_out_fused(int on) { g_attn_out_fused = on < 0 ? -1 : on ? 1 : 0; }
bool attn_out_fused_enabled() {
    static const bool env_off = getenv("MI355_ATTN_OUT_FUSED") && getenv("MI355_ATTN_OUT_FUSED")[0] == '0';
    return g_attn_out_fused < 0 ? !env_off : g_attn_out_fused > 0;
}
size_t attn_out_granule_words(int K) { return (size_t)(K / 256) * AO_GPB + 64; }
void attn_out_set_error_word(unsigned *w) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_st_err_word), &w, sizeof(w)); }

// chunk size the launch will use for a scan of n_kv_max cells: 64 cells while that gives every CU at most one item, 128 beyond
int attn_out_fused_chunk(const AttnArgs &a) {
    const int s64 = (a.n_kv_max + 63) / 64;
    if (a.tok_chunks) return 64;                      // (the per-token chunk lists count 64-cell chunks)
#ifdef MI355_AO_FORCE_C
    if (MI355_AO_FORCE_C == 128 || MI355_AO_FORCE_C == 64) return (MI355_AO_FORCE_C == 64 && s64 > 64) ? 128 : MI355_AO_FORCE_C;   // (tools: A/B of the chunk size)
#endif
    return s64 * a.G <= num_cu() && s64 <= 64 ? 64 : 128;
}
int attn_out_fused_splits(const AttnArgs &a) {
    const int C = attn_out_fused_chunk(a);
    return a.n_kv_max > 0 ? (a.n_kv_max + C - 1) / C : 1;
}

// a.splits as attn_out_fused_splits(a) says (with chunk lists: what launch_flash_attn_decode_fused takes)
bool attn_out_fused_applicable(const AttnArgs &a, const RopeArgs &ra, const MMVQSeg &wo, int K, int epi) {
    if (!attn_out_fused_enabled()) return false;
    if (a.T != 1 || a.D != AO_D || ra.neox || (ra.n_rot % 4) != 0 || ra.n_rot > a.D) return false;
    const int R = a.G > 0 ? a.H / a.G : 0;
    if (R * a.G != a.H || !(R == 1 || R == 2 || R == 4 || R == 8)) return false;
    const bool q8 = a.type_k == T_Q8_0 && a.type_v == T_Q8_0, f16 = a.type_k == T_F16 && a.type_v == T_F16;
    if (!q8 && !f16) return false;
    const int gp = (R * a.D) % 256 == 0 ? 1 : 2;
    if (a.G % gp != 0 || a.G / gp > 64) return false;
    if (!a.out_q || !a.out_q8k || a.out_q80) return false;
    if (wo.type != T_Q4_K && wo.type != T_Q5_K && wo.type != T_Q6_K) return false;
    if (K != a.H * a.D || (K % 256) != 0 || K > 8192 || wo.expert_sel) return false;
    if ((wo.row_bytes % 16) != 0 || (reinterpret_cast<uintptr_t>(wo.W) & 15) != 0) return false;
    if (epi != EPI_ADD && epi != EPI_STORE) return false;
    const int C = attn_out_fused_chunk(a);
    if (a.tok_chunks && C != 64) return false;
    // (with per-token chunk lists the lists say which chunks matter: the scan length does not bound them)
    if (a.splits < 1 || a.splits > 64 || (!a.tok_chunks && (size_t)a.splits * C < (size_t)a.n_kv_max)) return false;
    const int nwg = std::min(num_cu(), (wo.n_rows + 1) / 2);
    if (nwg < 1) return false;
    const int rpw = (wo.n_rows + nwg - 1) / nwg;
    if ((size_t)rpw * wo.row_bytes > 96 * 1024) return false;                  // the workgroup's rows stay in LDS whole
    return true;
}

void attn_out_probe_report() {
    if (!g_ao_probe || g_ao_probe_wgs <= 0) return;
    std::vector<unsigned long long> t((size_t)g_ao_probe_wgs * 16);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(t.data(), g_ao_probe, t.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < g_ao_probe_wgs; w++) if (t[(size_t)w * 16]) t0 = std::min(t0, t[(size_t)w * 16]);
    auto stat = [&](int k, const char *name) {
        double lo = 1e30, hi = 0, sum = 0; int n = 0;
        for (int w = 0; w < g_ao_probe_wgs; w++) {
            const unsigned long long v = t[(size_t)w * 16 + k];
            if (!v) continue;
            const double us = (double)(v - t0) * 0.01;
            lo = std::min(lo, us); hi = std::max(hi, us); sum += us; n++;
        }
        if (n) fprintf(stderr, "  %-28s n=%3d  min %.2f  mean %.2f  max %.2f us\n", name, n, lo, sum / n, hi);
    };
    fprintf(stderr, "attn_out probe: %d workgroups, %d attention items (us since the first workgroup entered)\n", g_ao_probe_wgs, g_ao_probe_items);
    stat(0, "entered"); stat(7, "qkv: activation ready"); stat(15, "item: q / k / v granules seen"); stat(8, "item: loads issued"); stat(9, "item: q rotated (loads back)"); stat(10, "item: scores done"); stat(11, "item: P.V done");
    stat(12, "item: partial stored+drained"); stat(1, "items done / dma issued"); stat(2, "merge: ticket won"); stat(13, "merge: weights done");
    stat(14, "merge: sums done"); stat(3, "merge: codes stored");
    stat(4, "all flags seen"); stat(5, "codes + rows in LDS"); stat(6, "outputs stored");
}

// the launch of either form: qf == nullptr: the attention + attn_output kernel; else the layer's Q | K | V in front of it in the same launch
struct QFPlan { QFArgs f; int blocks; int kb; size_t lds_qkv, lds_total; int slots; };
static hipError_t launch_ao(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew, const int32_t *tok_cell,
                            unsigned *counters, unsigned *flags, unsigned long long *gran, int layer, const unsigned *serial, const MMVQSeg &wo, int K, int epi,
                            const QFPlan *qf, hipStream_t st) {
    if (!counters || !flags || !gran || !serial || !tok_cell || layer < 0 || layer > 254) return hipErrorInvalidValue;
    if (!qf && (!knew || !vnew)) return hipErrorInvalidValue;
    const int R = a.H / a.G;
    const int C = attn_out_fused_chunk(a);
    AOArgs o{};
    o.W = wo.W; o.out = wo.out; o.resid = wo.resid; o.type = wo.type; o.n_rows = wo.n_rows; o.K = K; o.epi = epi;
    o.row_bytes = (unsigned)wo.row_bytes;
    const int nwg0 = std::min(num_cu(), (wo.n_rows + 1) / 2);
    o.rows_per_wg = (wo.n_rows + nwg0 - 1) / nwg0;
    const int nwg = (wo.n_rows + o.rows_per_wg - 1) / o.rows_per_wg;
    o.slice_lds = (unsigned)((((size_t)o.rows_per_wg * wo.row_bytes) + ST_SLOT - 1) / ST_SLOT * ST_SLOT);
    o.flags = flags; o.gran = gran; o.layer = layer; o.serial = serial;
    const int gp = (R * a.D) % 256 == 0 ? 1 : 2;
    o.n_flags = a.G / gp;
    o.n_items = a.G * a.splits;
    static const bool probe_on = getenv("MI355_AO_PROBE") && getenv("MI355_AO_PROBE")[0] == '1';
    if (probe_on) {
        if (!g_ao_probe && hipMalloc((void **)&g_ao_probe, 1024 * 16 * 8) != hipSuccess) g_ao_probe = nullptr;
        if (g_ao_probe) { (void)hipMemsetAsync(g_ao_probe, 0, 1024 * 16 * 8, st); g_ao_probe_wgs = std::min(nwg, 1024); g_ao_probe_items = o.n_items; }
        o.probe = g_ao_probe;
    }
    DecodeFuse fz{};
    fz.knew = knew; fz.vnew = vnew; fz.tok_cell = tok_cell; fz.counters = counters; fz.neox = 0;
    fz.q = *a.out_q; fz.want_q8k = 1; fz.want_q80 = 0;
    if (qf && qf->blocks != nwg) return hipErrorInvalidValue;
#define AO_LAUNCH(RR, TK, TV, CC)                                                                                                              \
    do {                                                                                                                                       \
        const size_t lds = ao_layout(o.slice_lds, K, sizeof(AOSmem<RR, CC>)).total;                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_out_kernel<RR, TK, TV, CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                                         \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                                             \
        if (kernel_timer() && kernel_timer()->next("attn_out", &ev0_, &ev1_))                                                                  \
            hipExtLaunchKernelGGL((attn_out_kernel<RR, TK, TV, CC>), dim3(nwg), dim3(AO_NT), lds, st, ev0_, ev1_, 0, a, cs_table, ra.n_rot, fz, o); \
        else hipLaunchKernelGGL((attn_out_kernel<RR, TK, TV, CC>), dim3(nwg), dim3(AO_NT), lds, st, a, cs_table, ra.n_rot, fz, o);             \
    } while (0)
#define QF_EARLY_ARGS qf->f.nx, qf->f.nw, qf->f.K     /* (repeated as the kernel's first, preloaded arguments) */
#define QF_LAUNCH(RR, TK, TV, CC, KBV)                                                                                                         \
    do {                                                                                                                                       \
        const size_t lds = qf->lds_total;                                                                                                      \
        if (lds > 160 * 1024 || o.slice_lds + AO_QF_HDR > qf->f.accs_off) return hipErrorInvalidValue;                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&qkv_attn_out_kernel<RR, TK, TV, CC, KBV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                                         \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                                             \
        if (kernel_timer() && kernel_timer()->next("qkv_attn_out", &ev0_, &ev1_))                                                              \
            hipExtLaunchKernelGGL((qkv_attn_out_kernel<RR, TK, TV, CC, KBV>), dim3(nwg), dim3(AO_QF_NT), lds, st, ev0_, ev1_, 0, QF_EARLY_ARGS, a, cs_table, ra.n_rot, fz, o, qf->f); \
        else hipLaunchKernelGGL((qkv_attn_out_kernel<RR, TK, TV, CC, KBV>), dim3(nwg), dim3(AO_QF_NT), lds, st, QF_EARLY_ARGS, a, cs_table, ra.n_rot, fz, o, qf->f); \
    } while (0)
#define AO_K(RR, TK, TV, CC) do { if (!qf) AO_LAUNCH(RR, TK, TV, CC); else if (qf->kb == 1) QF_LAUNCH(RR, TK, TV, CC, 1); else QF_LAUNCH(RR, TK, TV, CC, 2); } while (0)
#define AO_C(RR, TK, TV) do { if (C == 64) AO_K(RR, TK, TV, 64); else AO_K(RR, TK, TV, 128); } while (0)
#define AO_T(RR) do { if (a.type_k == T_F16) AO_C(RR, T_F16, T_F16); else AO_C(RR, T_Q8_0, T_Q8_0); } while (0)
#ifdef MI355_AO_DEV      // (development builds: one form only, seconds to compile)
    if (R == 4 && a.type_k == T_Q8_0 && C == 64) AO_K(4, T_Q8_0, T_Q8_0, 64); else return hipErrorInvalidValue;
#else
    switch (R) {
        case 1: AO_T(1); break;
        case 2: AO_T(2); break;
        case 4: AO_T(4); break;
        case 8: AO_T(8); break;
        default: return hipErrorInvalidValue;
    }
#endif
#undef AO_T
#undef AO_C
#undef AO_K
#undef QF_LAUNCH
#undef QF_EARLY_ARGS
#undef AO_LAUNCH
    return hipGetLastError();
}

hipError_t launch_attn_out_fused(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew, const int32_t *tok_cell,
                                 unsigned *counters, unsigned *flags, unsigned long long *gran, int layer, const unsigned *serial, const MMVQSeg &wo, int K, int epi,
                                 hipStream_t st) {
    return launch_ao(a, cs_table, ra, knew, vnew, tok_cell, counters, flags, gran, layer, serial, wo, K, epi, nullptr, st);
}

// ---- QF, host side
static int g_qkv_attn_fused = -1;           // -1: environment / default (on)
void set_qkv_attn_fused(int on) { g_qkv_attn_fused = on < 0 ? -1 : on ? 1 : 0; }
bool qkv_attn_fused_enabled() {
    static const bool env_off = getenv("MI355_QKV_ATTN_FUSED") && getenv("MI355_QKV_ATTN_FUSED")[0] == '0';
    return g_qkv_attn_fused < 0 ? !env_off : g_qkv_attn_fused > 0;
}
size_t qkv_attn_granule_words(int n_q, int n_kv) { return (size_t)n_q + 2 * (size_t)n_kv + 64; }

// the plan of the three projections over the launch's workgroups (mmvq_stream_plan's: per tensor in proportion to its bytes), or blocks == 0 where the shape has no form
static QFPlan qf_plan(const AttnArgs &a, const MMVQSeg &wo, int K, const QKVFuse &q) {
    QFPlan p{};
    if (!qkv_attn_fused_enabled() || !q.nx || !q.nw || !q.gran) return p;
    if (q.K != K || q.K != a.H * a.D || (q.K % 256) != 0 || q.K > 4096) return p;                  // (one activation region in LDS serves both mat-vecs)
    if (q.seg[0].n_rows != a.H * a.D || q.seg[1].n_rows != a.G * a.D || q.seg[2].n_rows != a.G * a.D) return p;
    if ((reinterpret_cast<uintptr_t>(q.nx) & 15) != 0 || (reinterpret_cast<uintptr_t>(q.nw) & 15) != 0) return p;
    for (int s = 0; s < 3; s++) {
        const MMVQSeg &g = q.seg[s];
        if (g.type != T_Q4_K && g.type != T_Q5_K && g.type != T_Q6_K && g.type != T_Q8_0) return p;    // (Q8_0: attn_k / attn_v of 8-expert files)
        if (g.expert_sel || (g.row_bytes % 16) != 0 || g.row_bytes > 0xffffffffull || (reinterpret_cast<uintptr_t>(g.W) & 15) != 0) return p;
    }
    const int nwg0 = std::min(num_cu(), (wo.n_rows + 1) / 2);
    if (nwg0 < 3) return p;
    const int rpw = (wo.n_rows + nwg0 - 1) / nwg0;
    const int nwg = (wo.n_rows + rpw - 1) / rpw;
    MMVQArgs m{};
    m.n_seg = 3; m.K = q.K; m.T = 1; m.epi = EPI_STORE;
    for (int s = 0; s < 3; s++) m.seg[s] = q.seg[s];
    mmvq_stream_plan(m, nwg);
    if (m.seg_block0[3] > nwg || m.seg_block0[1] <= 0 || m.seg_block0[2] <= m.seg_block0[1] || m.seg_block0[3] <= m.seg_block0[2]) return p;
    size_t run = 0;
    for (int s = 0; s < 3; s++) {
        const int nblk = m.seg_block0[s + 1] - m.seg_block0[s];
        const int rpb = (q.seg[s].n_rows + nblk - 1) / nblk;
        run = std::max(run, (size_t)rpb * q.seg[s].row_bytes);
    }
    p.lds_qkv = (run + ST_SLOT - 1) / ST_SLOT * ST_SLOT;
    const size_t wo_lds = ((size_t)rpw * wo.row_bytes + ST_SLOT - 1) / ST_SLOT * ST_SLOT;
    p.slots = (int)((p.lds_qkv + wo_lds) / ST_SLOT);
    if (p.slots > 60) return p;
    QFArgs &f = p.f;
    f.W0 = q.seg[0].W; f.W1 = q.seg[1].W; f.W2 = q.seg[2].W;
    f.rb0 = (unsigned)q.seg[0].row_bytes; f.rb1 = (unsigned)q.seg[1].row_bytes; f.rb2 = (unsigned)q.seg[2].row_bytes;
    f.t0 = q.seg[0].type; f.t1 = q.seg[1].type; f.t2 = q.seg[2].type;
    f.n0 = q.seg[0].n_rows; f.n1 = q.seg[1].n_rows; f.n2 = q.seg[2].n_rows;
    f.blk1 = m.seg_block0[1]; f.blk2 = m.seg_block0[2]; f.blk3 = m.seg_block0[3];
    f.nx = q.nx; f.nw = q.nw; f.neps = q.neps; f.K = q.K;
    f.gran = q.gran; f.n_q = a.H * a.D; f.n_kv = a.G * a.D;
    f.qkv_lds = (unsigned)p.lds_qkv;
    p.kb = (q.K + 2047) >> 11;
    // the kernel's LDS at the largest attention scratch of the forms (R, chunk) it may take
    const int R = a.H / a.G, C = attn_out_fused_chunk(a);
    size_t attn_bytes = 0;
#define QF_SZ(RR) attn_bytes = C == 64 ? sizeof(AOSmemQ<RR, 64>) : sizeof(AOSmemQ<RR, 128>)
    switch (R) { case 1: QF_SZ(1); break; case 2: QF_SZ(2); break; case 4: QF_SZ(4); break; case 8: QF_SZ(8); break; default: return p; }
#undef QF_SZ
    const AOLayout ql = ao_layout((unsigned)wo_lds, K, attn_bytes, AO_QF_HDR, f.qkv_lds);
    const size_t accs_bytes = (size_t)AO_NCG * R * AO_D * 4;
    f.accs_off = accs_bytes <= f.qkv_lds ? ql.qkv : ql.total;
    p.lds_total = accs_bytes <= f.qkv_lds ? ql.total : ql.total + accs_bytes;
    if (p.lds_total > 160 * 1024) return p;
    p.blocks = nwg;
    return p;
}
// host logic only (no device call): LDS bytes of the fused launch for a layer of this geometry at a context of n_kv cells, 0 where the step takes the two launches
size_t qkv_attn_out_plan_lds(int type_q, int type_k, int type_v, int type_o, int n_embd, int n_head, int n_head_kv, int head_dim, int type_kv, int n_kv, int *slots_out) {
    static float dummy[4];                                      // (addresses are only checked for alignment here)
    if (slots_out) *slots_out = 0;
    if (n_head < 1 || n_head_kv < 1 || head_dim < 1 || n_embd < 256) return 0;
    AttnArgs a{};
    a.type_k = type_kv; a.type_v = type_kv; a.T = 1; a.H = n_head; a.G = n_head_kv; a.D = head_dim; a.n_ctx = n_kv; a.n_kv_max = n_kv;
    ActQuant aq{};
    a.out_q = &aq; a.out_q8k = true; a.out_q80 = false;
    a.splits = attn_out_fused_splits(a);
    RopeArgs ra{};
    ra.n_rot = head_dim; ra.neox = 0;
    const int K = n_head * head_dim;
    const uint8_t *w = reinterpret_cast<const uint8_t *>(dummy);
    auto seg = [&](int type, int rows, int k) { MMVQSeg g{}; g.W = w; g.type = type; g.n_rows = rows; g.ld_out = rows; g.row_bytes = dev_row_bytes(type, k); return g; };
    const MMVQSeg wo = seg(type_o, n_embd, K);
    QKVFuse q{};
    q.seg[0] = seg(type_q, n_head * head_dim, n_embd); q.seg[1] = seg(type_k, n_head_kv * head_dim, n_embd); q.seg[2] = seg(type_v, n_head_kv * head_dim, n_embd);
    q.nx = dummy; q.nw = dummy; q.neps = 1e-5f; q.K = n_embd; q.gran = reinterpret_cast<unsigned long long *>(dummy);
    if (!attn_out_fused_applicable(a, ra, wo, K, EPI_ADD)) return 0;
    const QFPlan p = qf_plan(a, wo, K, q);
    if (p.blocks <= 0) return 0;
    if (slots_out) *slots_out = p.slots;
    return p.lds_total;
}
bool qkv_attn_out_applicable(const AttnArgs &a, const RopeArgs &ra, const MMVQSeg &wo, int K, int epi, const QKVFuse &q) {
    return attn_out_fused_applicable(a, ra, wo, K, epi) && qf_plan(a, wo, K, q).blocks > 0;
}
hipError_t launch_qkv_attn_out(const AttnArgs &a, const float *cs_table, RopeArgs ra, const int32_t *tok_cell, unsigned *counters, unsigned *flags, unsigned long long *gran,
                               int layer, const unsigned *serial, const MMVQSeg &wo, int K, int epi, const QKVFuse &q, hipStream_t st) {
    const QFPlan p = qf_plan(a, wo, K, q);
    if (p.blocks <= 0) return hipErrorInvalidValue;
    return launch_ao(a, cs_table, ra, nullptr, nullptr, tok_cell, counters, flags, gran, layer, serial, wo, K, epi, &p, st);
}

}  // namespace mi355
