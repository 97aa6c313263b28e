// attn_prefill.hip — prompt-processing attention on the matrix cores (ggml_flash_attn_ext for T >= 32 query rows;
// SURVEY.md §8a row a11), head_dim 128, q8_0 K and V cache — and, at the end of this comment, the f16 cache.
//
// Same arithmetic as the CPU path it stands in for (flash_attn_ext with a q8_0 K cache: q is quantised to Q8_0 per
// 32-block and K.q is vec_dot_q8_0_q8_0 = sum_b float(int dot) * (d_q * d_k); V rows are dequantised and accumulated in
// f32 with the online-softmax weights), laid out for MFMA:
//   * workgroup = (kv head g, tile of 32 query tokens), 4..8 waves = the R query heads of the group; a wave owns
//     32 queries x one head.  K / V chunks of 32 cells are staged in LDS once per workgroup and shared by the R heads.
//   * S^T = K . Q^T with v_mfma_i32_32x32x32_i8: A = K codes of one 32-block (rows = keys, from LDS), B = the wave's Q
//     codes of that block (columns = queries, resident in registers for the whole kernel); the int32 tile is scaled
//     by d_q (per lane = per query) x d_k (per register = per key) and summed over the 4 blocks in f32, as the CPU does.
//     With queries in the N dimension a lane holds ONE query (and 16 of the 32 keys): the softmax state (m, l) and the
//     rescale of O are per-lane scalars, the row max / sum need only the partner lane (lane ^ 32).
//   * O^T += V'^T . P^T with v_mfma_f32_32x32x16_f16: V' = code * d_v is exact in 18 bits, so it is split EXACTLY into two
//     f16 planes (hi + lo) when the chunk is staged; P is split into f16 hi + lo per wave (22 bits); three products
//     (Ph.Vh, Pl.Vh, Ph.Vl) accumulate in f32 — f32-grade accuracy on the f16 matrix pipe.  The MFMA K index is bound
//     to the key order the score tile leaves in registers, and the V^T planes are written to LDS in that order.
//   * visibility is the KV-cell test of the cache (pos >= 0, pos <= query pos, sequence bit), so ragged batches, several
//     sequences and fragmented caches take the same path; a prologue pass over the cell table marks the chunks that hold
//     a cell visible to some query of the tile, the main loop walks only those, with the next one's K / V in flight
//     (registers) while the current one is on the matrix cores.
// F16 = true (f16 K and V cache, the reference's default cache type): the CPU path converts q to f16 and takes
// vec_dot_f16 against the K row, so S^T = K . Q^T is eight v_mfma_f32_32x32x16_f16 per chunk (K rows as halfs in LDS, Q as
// halfs in registers, f32 accumulation, no block scales); V rows are already f16, so the hi plane IS V and the lo plane
// (and its product) disappear: O^T += V^T . (Ph + Pl)^T.  Everything else — visibility, online softmax, staging order — is shared.
#include <algorithm>
#include <cstdlib>
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int D = 128, NB = 4, CK = 32, QT = 32;      // head dim, 32-blocks per row, keys per chunk, queries per tile
constexpr int K_STRIDE = 144;                         // 128 codes + 16 B pad: ds_read_b128 conflict-free
constexpr int KF_STRIDE = 272;                        // f16 cache: 128 halfs + 16 B pad
constexpr int MAX_CHUNKS = 8192;                      // 262144 cells

// LDS layout of one chunk
struct Chunk {
    int8_t *k;          // [CK][K_STRIDE]
    float *dk;          // [NB][CK]      f32 block scales of K
    _Float16 *vh, *vl;  // [4 db][2 j][2 kg][32 m][8 slots]   V'^T planes in MFMA A-operand order
    int *cpos;          // [CK]
    unsigned long long *cseq;   // [CK]
};

// max / sum of a value with its copy in lane ^ 32 (the two lanes of a query): v_permlane32_swap hands each half of the wave the other
// half's registers in one VALU instruction - no LDS round trip (ds_bpermute) on the softmax's dependent chain.  Both lanes get the same bits
// as x op x[lane ^ 32] (the operations commute).
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// key index (0..31) of MFMA K-slot (j, kg, i): the order the score tile leaves its 16 keys per lane in registers
__device__ __forceinline__ int slot_key(int j, int kg, int i) { return 16 * j + 8 * (i >> 2) + 4 * kg + (i & 3); }

#ifdef MI355_FA_PROBE
// tools/build_fa_probe.sh: shader-clock split of one workgroup (printed by the last tile's first split): prologue | per chunk: scores,
// softmax, P.V, barrier, store, barrier
#define FA_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); fa_acc[i] += t_ - fa_t; fa_t = t_; } while (0)
#else
#define FA_T(i) do { } while (0)
#endif
// KH = key halves inside the workgroup (R <= 4: two): waves [0, R) walk the even chunks of the workgroup's run, waves [R, 2R) the odd ones,
// each half staging its own chunk buffer with its own 64 R threads; at the end the second half hands its (O, m, l) over through LDS and
// the first one merges them as two softmax partials.  A causal tile's chunk walk - the longest workgroups of a prompt - takes half
// the iterations, and with the long tiles dispatched first the short ones fill the slots they leave: no records, no second launch.
// QS = query sub-tiles of 32 per workgroup (R = 1: four, R = 2: two, else one - four waves per half in every case): the sub-tiles share
// every staged chunk, which a single-head workgroup of multi-head attention otherwise stages for one wave's use.
// Q4 (with F16 = false): a q4_0 K / V cache (cache_type "q4_0", src/llama_engine.cc:272-285).  Only the staging differs: a 16-byte piece of a K row is one
// 32-block of nibbles, unpacked to the int8 codes nibble - 8 on its way into LDS (elements 0..15 from the low nibbles, 16..31 from the high ones), V codes
// likewise; scores (vec_dot_q4_0_q8_0: the same integer dot with those codes), scales and the f16 hi / lo split of V' = code * d_v are the q8_0 path's.
template <int R, bool F16, int KH, int QS, int DH = 128, bool Q4 = false>
__global__ __launch_bounds__(64 * R * KH * QS, 2) void flash_attn_prefill_kernel(const AttnArgs a) {
    // head width: 128 or 64 (TinyLlama, Llama-3.2-1B, nomic-embed): everything below is written in 32-dim blocks (NB of them) and 16-dim matrix-core steps
    constexpr int D = DH, NB = DH / 32, K_STRIDE = DH + 16, KF_STRIDE = 2 * DH + 16, NKS = DH / 16, VPL = NB * 2 * 2 * 32 * 8, NVU = 2 * DH;
#ifdef MI355_FA_PROBE
    unsigned long long fa_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fa_t = __builtin_readcyclecounter();
    int fa_chunks = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KS = F16 ? KF_STRIDE : K_STRIDE;    // bytes per staged K row
    constexpr int PPK = F16 ? D / 8 : Q4 ? D / 32 : D / 16;      // 16-byte pieces per K row
    constexpr int KVROW = Q4 ? D / 2 : D;                         // bytes of a quantised K / V row
    constexpr int CHUNK_LDS = CK * KS + NB * CK * 4 + 2 * VPL * 2 + CK * 4 + CK * 8;   // one staged chunk (launcher: one per half)
    static_assert(CHUNK_LDS % 16 == 0, "chunk buffers stay 16-byte aligned");
    constexpr int NT = 64 * R * QS;                   // threads of one half: the staging roles below are per half
    const int tid_all = threadIdx.x, lane = tid_all & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int half = wave_all / (R * QS), wave = wave_all - half * (R * QS);   // wave: index inside the half = (query sub-tile, head of the kv group)
    const int sub = wave / R, hw = wave - sub * R;
    const int tid = tid_all - half * NT;
    auto chunk_at = [&](int par) {                    // buffer par of this half
        Chunk S;
        uint8_t *base = smem + (half * 2 + par) * CHUNK_LDS;
        S.k = reinterpret_cast<int8_t *>(base);
        S.dk = reinterpret_cast<float *>(base + CK * KS);
        S.vh = reinterpret_cast<_Float16 *>(base + CK * KS + NB * CK * 4);
        S.vl = S.vh + VPL;
        S.cpos = reinterpret_cast<int *>(S.vl + VPL);
        S.cseq = reinterpret_cast<unsigned long long *>(S.cpos + CK);
        return S;
    };
    __shared__ unsigned s_vis[MAX_CHUNKS / 32];                // bit c: some cell of chunk c is visible to some query of the tile
    __shared__ int s_tile_maxpos, s_tile_minpos;
    __shared__ unsigned long long s_tile_seqs;
    __shared__ int s_chunk_open_all[KH][2];                     // [half][parity]: every cell of the staged chunk is visible to every query of the tile
    int *s_chunk_open = s_chunk_open_all[half];
    __shared__ int s_nvis;                                      // marked chunks of this tile
    __shared__ int s_sub_maxpos[QS];                            // highest query position of each sub-tile
    __shared__ int s_chunk_minpos_all[KH][2];                   // [half][parity]: lowest position held by a staged chunk that is not open (a sub-tile wholly before it skips it)
    int *s_chunk_minpos = s_chunk_minpos_all[half];

    const int g = blockIdx.x, tile = (int)gridDim.y - 1 - (int)blockIdx.y;     // (the last tiles of a causal prompt walk the most chunks: first out)
    const int zsp = blockIdx.z, nsp = gridDim.z;               // key split: this workgroup walks its share of the tile's marked chunks
    const int n = lane & 31, kg = lane >> 5;
    const int H = a.H, n_ctx = a.n_ctx;
    const int h = g * R + hw;                                  // this wave's query head
    int qt = (tile * QS + sub) * QT + n;                       // this lane's query token
    const bool q_ok = qt < a.T;
    if (!q_ok) qt = a.T - 1;
    const int tpos = a.tok_pos[qt];
    const int tseq = a.tok_seq[qt];
    const int n_kv = *a.n_kv_dev;
    const int n_chunks = (n_kv + CK - 1) / CK;
    const size_t head_row0 = (size_t)g * n_ctx;

    // ---- tile-wide visibility bounds
    if (tid_all == 0) { s_tile_maxpos = -1; s_tile_minpos = 0x7fffffff; s_tile_seqs = 0ull; s_nvis = 0; }
    if (tid_all < QS) s_sub_maxpos[tid_all] = -1;
    for (int i = tid_all; i < MAX_CHUNKS / 32; i += NT * KH) s_vis[i] = 0u;
    __syncthreads();
    if (half == 0 && hw == 0 && kg == 0) {
        atomicMax(&s_tile_maxpos, tpos);
        atomicMax(&s_sub_maxpos[sub], tpos);
        atomicMin(&s_tile_minpos, tpos);
        atomicOr(&s_tile_seqs, 1ull << tseq);
    }

    // ---- Q of this lane's query: quantise to Q8_0 per 32-block (lane holds dims 32 b + 16 kg .. + 15 of block b);
    //      f16 cache: round to f16, lane holds dims 16 j + 8 kg .. + 7 of K-step j
    i32x4 qc[NB];
    float dq[NB];
    f16x8 qh[F16 ? NKS : 1];
    if constexpr (F16) {
        const float *qrow = a.q + ((size_t)qt * H + h) * D;
#pragma unroll
        for (int j = 0; j < NKS; j++) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(qrow + 16 * j + 8 * kg), x1 = *reinterpret_cast<const f32x4 *>(qrow + 16 * j + 8 * kg + 4);
            qh[j][0] = (_Float16)x0.x; qh[j][1] = (_Float16)x0.y; qh[j][2] = (_Float16)x0.z; qh[j][3] = (_Float16)x0.w;
            qh[j][4] = (_Float16)x1.x; qh[j][5] = (_Float16)x1.y; qh[j][6] = (_Float16)x1.z; qh[j][7] = (_Float16)x1.w;
        }
#pragma unroll
        for (int b = 0; b < NB; b++) { qc[b] = i32x4{0, 0, 0, 0}; dq[b] = 0.0f; }
    } else {
        const float *qrow = a.q + ((size_t)qt * H + h) * D;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            f32x4 x[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = *reinterpret_cast<const f32x4 *>(qrow + 32 * b + 16 * kg + 4 * i);
            float am = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) am = fmaxf(am, fmaxf(fmaxf(fabsf(x[i].x), fabsf(x[i].y)), fmaxf(fabsf(x[i].z), fabsf(x[i].w))));
            am = xhalf_max(am);                                // the other half of the block lives in lane ^ 32
            const float d = am / 127.0f;
            const float id = d != 0.0f ? 1.0f / d : 0.0f;
            int w[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int c0 = (int)roundf(x[i].x * id), c1 = (int)roundf(x[i].y * id), c2 = (int)roundf(x[i].z * id), c3 = (int)roundf(x[i].w * id);
                w[i] = (c0 & 0xff) | ((c1 & 0xff) << 8) | ((c2 & 0xff) << 16) | ((c3 & 0xff) << 24);
            }
            qc[b].x = w[0]; qc[b].y = w[1]; qc[b].z = w[2]; qc[b].w = w[3];
            dq[b] = h2f(f2h(d));
        }
    }

    // ---- staging roles: piece p = tid (+ NT) -> key p / PPK, 16-byte column p % PPK of the 128- (q8_0) or 256-byte (f16) K / V rows
    constexpr int NP = (CK * PPK + NT - 1) / NT;
    u32x4 kq[NP], vq[NP];
    constexpr int NKD = (CK * NB + NT - 1) / NT;
    float kdn[NKD];
    // q8_0 V: unit u = tid (+ NT) -> four consecutive keys 4 (u >> 5) .. + 3 x four dims 4 (u & 31) .. + 3: the four keys are four adjacent
    // K-slots of the transposed planes, so a dim's hi (lo) halves of the unit go out as ONE 8-byte LDS store (the key-per-thread
    // mapping wrote 32 two-byte stores per thread: the staging was 3.0 of a chunk's 8.3 kilocycles)
    constexpr int NV = F16 ? 1 : (NVU + NT - 1) / NT;
    uint32_t v4[NV][4];
    float vd4[NV][4];
    int cpn = -1;
    unsigned long long csn = 0ull;
    auto load_data = [&](int c) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + NT * i;
            int key = p / PPK;
            if (key >= CK) key = CK - 1;
            int cell = c * CK + key;
            if (cell >= n_ctx) cell = n_ctx - 1;
            const size_t rowi = head_row0 + cell;
            if constexpr (F16) {
                kq[i] = *reinterpret_cast<const u32x4 *>(a.kv.k + rowi * D * 2 + (p % PPK) * 16);
                vq[i] = *reinterpret_cast<const u32x4 *>(a.kv.v + rowi * D * 2 + (p % PPK) * 16);
            } else {
                kq[i] = *reinterpret_cast<const u32x4 *>(a.kv.k + rowi * KVROW + (p % PPK) * 16);
            }
        }
        if constexpr (!F16) {
#pragma unroll
            for (int i = 0; i < NV; i++) {
                int u = tid + NT * i;
                if (u >= NVU) u = NVU - 1;
                const int kgp = u / (D / 4), dg = u % (D / 4);
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    int cell = c * CK + 4 * kgp + kk;
                    if (cell >= n_ctx) cell = n_ctx - 1;
                    const size_t rowi = head_row0 + cell;
                    if constexpr (Q4) {     // dims 4 dg .. + 3 of block dg >> 3: four nibbles, the low or the high halves of four consecutive bytes; kept as 0 .. 15
                        const int e0 = (4 * dg) & 31;
                        const uint32_t w = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * KVROW + (dg >> 3) * 16 + (e0 & 15));
                        v4[i][kk] = (e0 < 16 ? w : (w >> 4)) & 0x0f0f0f0fu;
                    } else
                    v4[i][kk] = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + 4 * dg);
                    vd4[i][kk] = h2f(a.kv.vd[rowi * NB + (dg >> 3)]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (F16 ? 0 : NKD); i++) {
            int q = tid + NT * i;
            if (q >= CK * NB) q = CK * NB - 1;
            int cell = c * CK + (q & 31);
            if (cell >= n_ctx) cell = n_ctx - 1;
            kdn[i] = h2f(a.kv.kd[(head_row0 + cell) * NB + (q >> 5)]);
        }
        cpn = -1; csn = 0ull;
        const int mcell = c * CK + tid;
        if (tid < CK && mcell < n_kv) { cpn = a.cell_pos[mcell]; csn = a.cell_seq[mcell]; }
    };
    auto store_chunk = [&](int par) {                            // the registers load_data filled -> buffer par of this half
        const Chunk S = chunk_at(par);
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int p = tid + NT * i;
            if (p >= CK * PPK) continue;
            const int key = p / PPK, col = p % PPK;            // q8_0: dims 16 col .. + 15; f16: dims 8 col .. + 7
            if constexpr (Q4) {                                // piece = 32-block col: codes (nibble - 8) as int8, low nibbles first
                u32x4 lo, hi;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t x = kq[i][w];
                    lo[w] = (((x & 0x0f0f0f0fu) | 0x80808080u) - 0x08080808u) ^ 0x80808080u;          // per byte: n - 8 in two's complement, no borrow across bytes
                    hi[w] = ((((x >> 4) & 0x0f0f0f0fu) | 0x80808080u) - 0x08080808u) ^ 0x80808080u;
                }
                *reinterpret_cast<u32x4 *>(S.k + key * KS + col * 32) = lo;
                *reinterpret_cast<u32x4 *>(S.k + key * KS + col * 32 + 16) = hi;
            } else
            *reinterpret_cast<u32x4 *>(S.k + key * KS + col * 16) = kq[i];
            if constexpr (F16) {                               // V is f16 already: the hi plane is V itself, there is no lo plane; written
                const int j = key >> 4, r8 = key & 15;          // transposed in MFMA K-slot order: key = 16 j + 8 (i >> 2) + 4 kgs + (i & 3)
                const int kgs = (r8 >> 2) & 1, slot = ((r8 >> 3) << 2) | (r8 & 3);
                const uint32_t wv[4] = {vq[i].x, vq[i].y, vq[i].z, vq[i].w};
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int dim = col * 8 + e;
                    const uint16_t hb = (uint16_t)((wv[e >> 1] >> (16 * (e & 1))) & 0xffff);
                    const int o = ((((dim >> 5) * 2 + j) * 2 + kgs) * 32 + ((dim & 31) ^ (dim >> 5))) * 8 + slot;
                    reinterpret_cast<uint16_t *>(S.vh)[o] = hb;
                }
            }
        }
        if constexpr (!F16) {
            // V' = code * d_v, split exactly into f16 hi + lo, transposed into MFMA K-slot order: keys 4 kgp .. + 3 are slots slot0 .. + 3 of
            // (j = key >> 4, kgs = (key >> 2) & 1)
#pragma unroll
            for (int i = 0; i < NV; i++) {
                const int u = tid + NT * i;
                if (u >= NVU) continue;
                const int kgp = u / (D / 4), dg = u % (D / 4);
                const int j = kgp >> 2, kgs = kgp & 1, slot0 = ((kgp >> 1) & 1) * 4;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int dim = 4 * dg + e;
                    _Float16 hi[4], lo[4];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) {
                        const float v = (Q4 ? (float)((int)((v4[i][kk] >> (8 * e)) & 0xff) - 8) : (float)(int8_t)((v4[i][kk] >> (8 * e)) & 0xff)) * vd4[i][kk];
                        hi[kk] = (_Float16)v;
                        lo[kk] = (_Float16)(v - (float)hi[kk]);
                    }
                    const int o = ((((dim >> 5) * 2 + j) * 2 + kgs) * 32 + ((dim & 31) ^ (dim >> 5))) * 8 + slot0;   // (column ^ dim block: the 32 lanes of a store spread over the banks)
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<f16x4 *>(S.vh + o) = f16x4{hi[0], hi[1], hi[2], hi[3]};
                    *reinterpret_cast<f16x4 *>(S.vl + o) = f16x4{lo[0], lo[1], lo[2], lo[3]};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (F16 ? 0 : NKD); i++) {
            const int q = tid + NT * i;
            if (q < CK * NB) S.dk[(q >> 5) * CK + (q & 31)] = kdn[i];
        }
        if (tid < CK) { S.cpos[tid] = cpn; S.cseq[tid] = csn; }
        // a chunk below the diagonal of a causal prompt is visible to the whole tile: no per-score cell test then (16 x two LDS reads, a
        // 64-bit shift and three compares per lane and chunk).  Lanes 0..31 of wave 0 hold the chunk's cells.
        if (wave == 0) {
            const bool open = kg == 1 || (cpn >= 0 && cpn <= s_tile_minpos && (csn & s_tile_seqs) == s_tile_seqs);
            const bool all_open = __all(open);
            int mp = -1;
            if (QS > 1 && !all_open) {                        // (a tile of several sub-tiles: its diagonal spans several chunks)
                mp = (kg == 0 && cpn >= 0) ? cpn : 0x7fffffff;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) mp = min(mp, __shfl_xor(mp, o, 64));
            }
            if (lane == 0) { s_chunk_open[par] = all_open ? 1 : 0; if (QS > 1) s_chunk_minpos[par] = mp; }
        }
    };
    auto next_visible = [&](int from) -> int {                   // first marked chunk >= from, or n_chunks
        int c = from;
        while (c < n_chunks) {
            const unsigned w = s_vis[c >> 5] >> (c & 31);
            if (w) return c + __builtin_ctz(w);
            c = (c | 31) + 1;
        }
        return n_chunks;
    };

    // ---- online-softmax state of this lane's query and the O^T accumulators (d = 32 db + row of the tile)
    float m_run = -INFINITY, l_run = 0.0f;
    const float scale_l2 = a.scale * 1.4426950408889634f;       // softmax scale * log2(e)
    f32x16 O[NB];
#pragma unroll
    for (int db = 0; db < NB; db++)
#pragma unroll
        for (int r = 0; r < 16; r++) O[db][r] = 0.0f;

    __syncthreads();                                            // tile bounds complete
    // ---- which chunks matter: one pass over the cell table (a wave's 64 cells = two chunks per ballot)
    for (int cell0 = 0; cell0 < n_chunks * CK; cell0 += NT * KH) {
        const int cell = cell0 + tid_all;
        int mine = 0;
        if (cell < n_kv) {
            const int cp = a.cell_pos[cell];
            const unsigned long long cs = a.cell_seq[cell];
            mine = (cp >= 0 && cp <= s_tile_maxpos && (cs & s_tile_seqs) != 0ull) ? 1 : 0;
        }
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) {
            const int c0 = (cell0 + wave_all * 64) / CK;
            if (c0 < MAX_CHUNKS) {
                unsigned bits = ((unsigned)(bal & 0xffffffffull) != 0u ? 1u : 0u) | ((unsigned)(bal >> 32) != 0u ? 2u : 0u);
                if (bits) atomicOr(&s_vis[c0 >> 5], bits << (c0 & 31));
            }
        }
    }
    __syncthreads();
    // ---- key split (causal prompts leave tile i with i + 1 chunks: without it the last tile's workgroup runs 16x as long as the
    // first one's and half the chip idles): the marked chunks of the tile are dealt out in equal contiguous runs to the nsp
    // workgroups of the tile; each leaves an unnormalised (O, m, l) record per query and head, flash_attn_combine_kernel merges them
    int c = next_visible(0), left;                               // left: marked chunks of this workgroup's run
    {
        if (wave_all == 0) {
            int cnt = 0;
            for (int i = lane; i < (n_chunks + 31) / 32; i += 64) cnt += __builtin_popcount(s_vis[i]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) s_nvis = cnt;
        }
        __syncthreads();
        const int nvis = s_nvis, per = (nvis + nsp - 1) / nsp;
        const int first = zsp * per;
        left = nvis - first < per ? nvis - first : per;
        if (left < 0) left = 0;
        for (int i = 0; i < first + half && c < n_chunks; i++) c = next_visible(c + 1);     // (the second half starts one chunk in)
    }
    const int mine = (left + KH - 1 - half) / KH;               // chunks this half walks: every KH-th of the run
    const int n_it = (left + KH - 1) / KH;                      // iterations of the workgroup (the barriers are everybody's)
    // two buffers per half: chunk c is on the matrix cores from one while chunk c + 1 (requested an iteration ago) is converted into the
    // other and chunk c + 2's requests go out: ONE barrier per chunk (eight waves meet at it; the second one cost 0.5 kilocycles a chunk)
    auto next_mine = [&](int cur) -> int { int x = next_visible(cur + 1); if (KH == 2) x = next_visible(x + 1); return x; };
    int cn = n_chunks, par = 0;
    const int sub_maxpos = s_sub_maxpos[sub];
    const bool sub_live = (tile * QS + sub) * QT < a.T;         // (a ragged last tile may leave a sub-tile without queries)
    if (mine > 0) { load_data(c); store_chunk(0); }
    if (mine > 1) { cn = next_mine(c); load_data(cn); }
    __syncthreads();
    FA_T(0);
    for (int it = 0; it < n_it; it++) {
#ifdef MI355_FA_PROBE
        fa_chunks++;
#endif
        int cnn = n_chunks;
        if (it + 1 < mine) {
            store_chunk(par ^ 1);
            if (it + 2 < mine) { cnn = next_mine(cn); load_data(cnn); }     // in flight while this chunk and the next conversion run
        }
        FA_T(5);
        if (it < mine && (QS == 1 || (sub_live && (s_chunk_open[par] || s_chunk_minpos[par] <= sub_maxpos)))) {   // (wave-uniform)
            const Chunk S = chunk_at(par);
            // ---- scores S^T[key][query] of this chunk
            float sc[16];
#pragma unroll
            for (int r = 0; r < 16; r++) sc[r] = 0.0f;
            if constexpr (F16) {
                f32x16 sa;
#pragma unroll
                for (int r = 0; r < 16; r++) sa[r] = 0.0f;
#pragma unroll
                for (int j = 0; j < NKS; j++) {
                    const f16x8 ak = *reinterpret_cast<const f16x8 *>(S.k + n * KS + (16 * j + 8 * kg) * 2);
                    sa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ak, qh[j], sa, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) sc[r] = sa[r];
            }
#pragma unroll
            for (int b = 0; b < (F16 ? 0 : NB); b++) {
                const i32x4 ak = *reinterpret_cast<const i32x4 *>(S.k + n * K_STRIDE + 32 * b + 16 * kg);
                i32x16 z;
#pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0;
                const i32x16 I = __builtin_amdgcn_mfma_i32_32x32x32_i8(ak, qc[b], z, 0, 0, 0);
#pragma unroll
                for (int rq = 0; rq < 4; rq++) {
                    const f32x4 dk4 = *reinterpret_cast<const f32x4 *>(S.dk + b * CK + 8 * rq + 4 * kg);   // keys 8 rq + 4 kg + (0..3)
#pragma unroll
                    for (int ri = 0; ri < 4; ri++) sc[rq * 4 + ri] += (float)I[rq * 4 + ri] * (dq[b] * dk4[ri]);
                }
            }
            // ---- mask + online softmax (this lane: one query, 16 keys; partner lane ^ 32: the other 16)
            FA_T(1);
            // (scores and the running maximum are kept in log2 units: exp(s - m) is one v_exp_f32 of (s - m) * log2(e), which is
            // folded into the softmax scale; exp2(-inf) = 0 does the masking; the record a split leaves converts m back)
            float mloc = -INFINITY;
            if (s_chunk_open[par]) {                                          // (uniform over the half)
#pragma unroll
                for (int r = 0; r < 16; r++) { sc[r] *= scale_l2; mloc = fmaxf(mloc, sc[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const int cp = S.cpos[key];
                    const unsigned long long cs = S.cseq[key];
                    const bool vis = q_ok && cp >= 0 && cp <= tpos && ((cs >> tseq) & 1ull);
                    sc[r] = vis ? sc[r] * scale_l2 : -INFINITY;
                    mloc = fmaxf(mloc, sc[r]);
                }
            }
            mloc = xhalf_max(mloc);
            const float m_new = fmaxf(m_run, mloc);
            const float m_ref = m_new == -INFINITY ? 0.0f : m_new;           // (nothing visible yet: exp2(-inf - 0) = 0, no NaN)
            float p[16], lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                p[r] = __builtin_amdgcn_exp2f(sc[r] - m_ref);
                lsum += p[r];
            }
            lsum = xhalf_sum(lsum);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_ref);        // m_run = -inf: 0
            l_run = l_run * alpha + lsum;
            m_run = m_new;
            if (!__all(alpha == 1.0f)) {                                      // (the maximum moves in the first chunks, rarely later)
#pragma unroll
                for (int db = 0; db < NB; db++)
#pragma unroll
                    for (int r = 0; r < 16; r++) O[db][r] *= alpha;
            }
            // ---- P^T operands: f16 hi / lo, two MFMAs' worth (regs 0..7 and 8..15)
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const float v = p[8 * j + i];
                    const _Float16 hi = (_Float16)v;
                    ph[j][i] = hi;
                    pl[j][i] = (_Float16)(v - (float)hi);
                }
            FA_T(2);
            // ---- O^T[d][query] += V'^T . P^T
#pragma unroll
            for (int db = 0; db < NB; db++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const int o = (((db * 2 + j) * 2 + kg) * 32 + (n ^ db)) * 8;
                    const f16x8 avh = *reinterpret_cast<const f16x8 *>(S.vh + o);
                    const f16x8 avl = *reinterpret_cast<const f16x8 *>(S.vl + o);
                    O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, ph[j], O[db], 0, 0, 0);
                    O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avh, pl[j], O[db], 0, 0, 0);
                    if constexpr (!F16) O[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(avl, ph[j], O[db], 0, 0, 0);
                }
        }
        FA_T(3);
        __syncthreads();                                        // chunk c consumed, chunk cn staged
        FA_T(4);
        c = cn; cn = cnn; par ^= 1;
    }
    if constexpr (KH == 2) {
        // ---- the second half's partial (O, m, l) -> LDS (the chunk buffers are done with) -> merged into the first half's: two softmax
        // partials over disjoint keys, O = O_a 2^(m_a - m) + O_b 2^(m_b - m)
        constexpr int MR = 16 * NB;                                                     // O registers of a lane
        float *mrg = reinterpret_cast<float *>(smem) + (size_t)wave * (MR + 2) * 64;    // [MR + 2][64 lanes] per head: the O registers, m, l
        if (half == 1) {
#pragma unroll
            for (int db = 0; db < NB; db++)
#pragma unroll
                for (int r = 0; r < 16; r++) mrg[(db * 16 + r) * 64 + lane] = O[db][r];
            mrg[MR * 64 + lane] = m_run; mrg[(MR + 1) * 64 + lane] = l_run;
        }
        __syncthreads();
        if (half == 1) return;
        const float m_b = mrg[MR * 64 + lane], l_b = mrg[(MR + 1) * 64 + lane];
        const float m_new = fmaxf(m_run, m_b);
        const float m_ref = m_new == -INFINITY ? 0.0f : m_new;
        const float wa = __builtin_amdgcn_exp2f(m_run - m_ref), wb = __builtin_amdgcn_exp2f(m_b - m_ref);
#pragma unroll
        for (int db = 0; db < NB; db++)
#pragma unroll
            for (int r = 0; r < 16; r++) O[db][r] = O[db][r] * wa + mrg[(db * 16 + r) * 64 + lane] * wb;
        l_run = l_run * wa + l_b * wb;
        m_run = m_new;
    }
#ifdef MI355_FA_PROBE
    if (tid == 0 && g == 0 && tile == (int)gridDim.y - 1 && zsp == 0)
        printf("fa probe: %d chunks; cycles: prologue %llu | store + requests %llu scores %llu softmax %llu PV %llu barrier %llu\n", fa_chunks,
               fa_acc[0], fa_acc[5], fa_acc[1], fa_acc[2], fa_acc[3], fa_acc[4]);
#endif

    if (nsp > 1) {                                              // partial record [D O][m][l] of (query, head, split)
        if (q_ok) {
            float *dst = a.part + (((size_t)qt * H + h) * nsp + zsp) * (D + 2);
#pragma unroll
            for (int db = 0; db < NB; db++)
#pragma unroll
                for (int rq = 0; rq < 4; rq++) {
                    float *o = dst + 32 * db + 8 * rq + 4 * kg;  // records are 8-byte aligned only ((D + 2) floats)
                    o[0] = O[db][rq * 4 + 0]; o[1] = O[db][rq * 4 + 1]; o[2] = O[db][rq * 4 + 2]; o[3] = O[db][rq * 4 + 3];
                }
            if (kg == 0) { dst[D] = m_run * 0.6931471805599453f; dst[D + 1] = l_run; }   // (the merge works in natural-log units)
        }
        return;
    }
    // ---- out[t][h][d] = O / l ; lane (query n, kg) holds d = 32 db + 8 rq + 4 kg + (0..3)
    if (q_ok) {
        const float inv = 1.0f / l_run;
        float *orow = a.out + ((size_t)qt * H + h) * D;
#pragma unroll
        for (int db = 0; db < NB; db++)
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
                f32x4 v;
                v.x = O[db][rq * 4 + 0] * inv; v.y = O[db][rq * 4 + 1] * inv; v.z = O[db][rq * 4 + 2] * inv; v.w = O[db][rq * 4 + 3] * inv;
                *reinterpret_cast<f32x4 *>(orow + 32 * db + 8 * rq + 4 * kg) = v;
            }
    }
}

}  // namespace

bool flash_attn_prefill_applicable(const AttnArgs &a) {
    const int R = a.G > 0 ? a.H / a.G : 0;
    // head_dim 64 (TinyLlama, Llama-3.2-1B, nomic-embed): the power-of-two head ratios
    const bool pow2 = R == 1 || R == 2 || R == 4 || R == 8;
    if (a.D == 64 && !pow2) return false;
    return (a.D == 128 || a.D == 64) && a.T >= 32 && ((a.type_k == T_Q8_0 && a.type_v == T_Q8_0) || (a.type_k == T_F16 && a.type_v == T_F16) || (a.type_k == T_Q4_0 && a.type_v == T_Q4_0 && a.D == 128 && pow2)) && R >= 1 && R <= 8 &&
           a.n_kv_max <= MAX_CHUNKS * CK;
}

// Key splits of a prompt batch.  Two things pull: the causal tiles are 1 .. n_chunks long, so a few splits balance the workgroups
// (512 tokens: 128 (kv head, tile) pairs on 512 workgroup slots, the last 16x as long as the first - 4 splits took 3.1 -> 2.0 ms);
// but every split leaves a (D + 2)-float record per query and head that the merge reads back (2048 tokens x 8 splits: 272 MB, a
// 110 us merge per layer beside a 327 us kernel).  So: as many splits as fill ~512 workgroups (measured, 8B model: 512 tokens 4 splits
// 1.76 ms of attention per prompt against 1.80 with 2 and 2.07 with 6; 2048 tokens NO split 7.1 ms against 8.8 with 2 - the 512
// (kv head, tile) pairs fill the chip by themselves and the merge costs more than the imbalance), never more than a quarter of the
// chunks, at most 8, and a workspace of at most 512 MiB.
static int fa_key_halves(int R) {                               // key halves per workgroup: eight waves at most
    static const int env = getenv("MI355_FA_KH") ? atoi(getenv("MI355_FA_KH")) : 0;
    return (R <= 4 && env != 1) ? 2 : 1;
}
static int fa_sub_tiles(int R) {                                // query sub-tiles per workgroup: four waves per half
    static const int env = getenv("MI355_FA_QS") ? atoi(getenv("MI355_FA_QS")) : 0;
    return env == 1 ? 1 : R == 1 ? 4 : R == 2 ? 2 : 1;
}

int flash_attn_prefill_splits(int T, int H, int G, int D, int n_kv_max) {
    static const int env = getenv("MI355_ATTN_PREFILL_SPLITS") ? atoi(getenv("MI355_ATTN_PREFILL_SPLITS")) : 0;
    const int R = G > 0 ? H / G : 1, kh = fa_key_halves(R), qs = fa_sub_tiles(R);
    const int n_chunks = (n_kv_max + CK - 1) / CK, tiles = (T + QT * qs - 1) / (QT * qs);
    const int pairs = (G > 0 ? G : 1) * tiles;
    const int slots = 2048 / (R * kh * qs > 0 ? R * kh * qs : 1);  // workgroups the chip holds at two waves per SIMD
    int s = (slots + pairs - 1) / pairs;
    if (s > n_chunks / (4 * kh)) s = n_chunks / (4 * kh);        // (~4 chunks per half and workgroup at least)
    if (s > 8) s = 8;
    if (env > 0) s = env;
    while (s > 1 && (size_t)T * H * s * (D + 2) * sizeof(float) > ((size_t)512 << 20)) s >>= 1;
    return s < 1 ? 1 : s;
}

hipError_t launch_flash_attn_prefill(const AttnArgs &a, hipStream_t st) {
    const int R = a.H / a.G, kh = fa_key_halves(R), qs = fa_sub_tiles(R);
    const int nsp = (a.pf_splits > 1 && a.part) ? a.pf_splits : 1;
    const dim3 grid((unsigned)a.G, (unsigned)((a.T + QT * qs - 1) / (QT * qs)), (unsigned)nsp);
    const bool f16 = a.type_k == T_F16;
    const int nb = a.D / 32;
    const size_t chunk = (size_t)CK * (f16 ? 2 * a.D + 16 : a.D + 16) + nb * CK * 4 + 2 * (size_t)(nb * 2 * 2 * 32 * 8) * 2 + CK * 4 + CK * 8;
    const size_t lds = kh == 2 ? std::max(4 * chunk, (size_t)R * qs * (16 * nb + 2) * 64 * 4) : 2 * chunk;   // two chunk buffers per half, later the hand-over of the second half
#define FAP_D(RR, KK, QQ, DD) do { \
        if (f16) { if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&flash_attn_prefill_kernel<RR, true, KK, QQ, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                   hipLaunchKernelGGL((flash_attn_prefill_kernel<RR, true, KK, QQ, DD>), grid, dim3(64 * RR * KK * QQ), lds, st, a); } \
        else { if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&flash_attn_prefill_kernel<RR, false, KK, QQ, DD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
               hipLaunchKernelGGL((flash_attn_prefill_kernel<RR, false, KK, QQ, DD>), grid, dim3(64 * RR * KK * QQ), lds, st, a); } } while (0)
#define FAP(RR, KK, QQ) FAP_D(RR, KK, QQ, 128)
#define FAP_Q4(RR, KK, QQ) do { if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&flash_attn_prefill_kernel<RR, false, KK, QQ, 128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                                hipLaunchKernelGGL((flash_attn_prefill_kernel<RR, false, KK, QQ, 128, true>), grid, dim3(64 * RR * KK * QQ), lds, st, a); } while (0)
    if (a.type_k == T_Q4_0) {                                   // head_dim 128, the default wave arrangement of the power-of-two head ratios
        if (R == 1 && kh == 2 && qs == 4) FAP_Q4(1, 2, 4);
        else if (R == 2 && kh == 2 && qs == 2) FAP_Q4(2, 2, 2);
        else if (R == 4 && kh == 2 && qs == 1) FAP_Q4(4, 2, 1);
        else if (R == 8 && kh == 1 && qs == 1) FAP_Q4(8, 1, 1);
        else return hipErrorInvalidValue;
    } else
    if (a.D == 64) {                                            // the default wave arrangement of each head ratio only
        if (R == 1 && kh == 2 && qs == 4) FAP_D(1, 2, 4, 64);
        else if (R == 2 && kh == 2 && qs == 2) FAP_D(2, 2, 2, 64);
        else if (R == 4 && kh == 2 && qs == 1) FAP_D(4, 2, 1, 64);
        else if (R == 8 && kh == 1 && qs == 1) FAP_D(8, 1, 1, 64);
        else return hipErrorInvalidValue;
    } else
    switch (R) {
        case 1: if (kh == 2 && qs == 4) FAP(1, 2, 4); else if (kh == 2) FAP(1, 2, 1); else if (qs == 4) FAP(1, 1, 4); else FAP(1, 1, 1); break;
        case 2: if (kh == 2 && qs == 2) FAP(2, 2, 2); else if (kh == 2) FAP(2, 2, 1); else if (qs == 2) FAP(2, 1, 2); else FAP(2, 1, 1); break;
        case 4: if (kh == 2) FAP(4, 2, 1); else FAP(4, 1, 1); break;
        case 8: FAP(8, 1, 1); break;
        // 3, 5, 6, 7 query heads per kv head (Llama-3.2-3B, Qwen2-1.5B / 7B, Yi-34B): one wave per head as for 4 and 8 (the workgroup is 3 .. 7 waves wide per half)
        case 3: if (kh == 2) FAP(3, 2, 1); else FAP(3, 1, 1); break;
        case 5: FAP(5, 1, 1); break;
        case 6: FAP(6, 1, 1); break;
        case 7: FAP(7, 1, 1); break;
        default: return hipErrorInvalidValue;
    }
#undef FAP
#undef FAP_Q4
#undef FAP_D
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (nsp > 1) return launch_flash_attn_combine(a, nsp, st);      // merges the splits and quantises the rows when asked
    if (a.out_q) e = launch_quantize(a.out, a.H * a.D, a.T, *a.out_q, a.out_q8k, a.out_q80, st, a.out_bh, a.out_bl);
    return e;
}

}  // namespace mi355
