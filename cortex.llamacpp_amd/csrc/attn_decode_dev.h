// attn_decode_dev.h — device side of the decode attention (one 64-cell chunk of one kv head per 256-thread workgroup).
// Shared by attn.hip (one launch per layer) and decode_mega.hip (a phase of the whole-step kernel).
#pragma once
#include "kernels.h"
#include "quant_dev.h"

namespace mi355 {

// quantize_row_q4_0 of a 32-block held by 8 consecutive lanes, 4 consecutive values each: the FIRST element of largest magnitude sets d = max / -8, codes
// min(15, (int8)(x / d + 8.5)); nib4 = this lane's four codes, one per byte (0 .. 15).  The block's 16 bytes are q[j] | q[j + 16] << 4: lanes 0-3 of the group
// hold the low nibbles, lanes 4-7 the high ones (combine with the lane 4 further on).
__device__ __forceinline__ void wave_quant_q40(const float (&vv)[4], int lane, uint32_t &nib4, float &d) {
    // key: |x| bits above, first-index preference below (larger key = larger magnitude, then smaller index)
    unsigned long long key = 0ull;
    float vbest = 0.0f;
#pragma unroll
    for (int i = 3; i >= 0; i--) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(fabsf(vv[i])) << 32) | (unsigned long long)(0xffffffffu - (unsigned)((lane & 7) * 4 + i));
        if (k > key) { key = k; vbest = vv[i]; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned long long ok = __shfl_xor(key, o, 64);
        const float ov = __shfl_xor(vbest, o, 64);
        if (ok > key) { key = ok; vbest = ov; }
    }
    d = vbest / -8.0f;
    const float id = d != 0.0f ? 1.0f / d : 0.0f;
    nib4 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int q = (int)(int8_t)(vv[i] * id + 8.5f);
        q = q > 15 ? 15 : q;
        nib4 |= (uint32_t)(q & 0xff) << (8 * i);
    }
}

// four nibbles (one per byte, 0 .. 15) -> the int8 codes nibble - 8, per byte without borrows across bytes
__device__ __forceinline__ uint32_t nib_to_i8x4(uint32_t n) { return ((n | 0x80808080u) - 0x08080808u) ^ 0x80808080u; }

// Decode attention: workgroup = (kv head g, chunk of 64 cells, token t).  The query heads of the group are read
// un-rotated, rotated with the cos/sin table and converted to the K cache's dot type here (so no separate pass over q),
// and K / V / scales / cell table of the chunk are all requested before the first wait.
//
// FUSED (single-token step, R*D a multiple of 256) folds the two neighbouring launches into this one:
//  * KV store: the workgroup whose chunk holds the token's cell rotates K, converts K/V to the cache type, writes the
//    cache row and patches its own registers / LDS scales with the same codes (no other workgroup reads that cell).
//    A batched step whose tokens all belong to different sequences takes this part too (T > 1, no tickets): another
//    token's workgroup may read the row while it is written, but that cell is not visible to it - its score is replaced
//    by -inf and its V (old or new bytes, finite either way) is weighted by zero;
//  * merge: every workgroup publishes its chunk partial, then takes a ticket on a per-kv-head counter; the LAST arriver
//    merges the splits of its R heads exactly as flash_attn_combine_kernel does, writes the f32 rows and the quantised
//    activation of the attn_output mat-vec, and re-arms the counter.  Nobody waits: no spin, no co-residency needed.
struct DecodeFuse {
    const float *knew, *vnew;      // [G*D] un-rotated K and V of the token
    const int32_t *tok_cell;       // [1]
    unsigned *counters;            // [G] words ATT_SYNC_STRIDE apart (a 128-byte line each), zero between launches
    ActQuant q;
    int want_q8k, want_q80;
    int neox;                      // rotary pairing (i, i + D / 2) instead of (2i, 2i + 1): qwen2-type files; the whole head rotates (n_rot == D)
    // diagnosis (MI355_ATTN_PROBE=1): 100 MHz wall-clock stamps.  [2 * wg], [2 * wg + 1] = start / partials-stored of every
    // workgroup (wg = sp * G + g); [4096 + 8 * g + k] = the merging workgroup of kv head g: ticket taken, weights done,
    // partials summed, outputs stored
    unsigned long long *probe;
};

// body of one (kv head g, chunk slot sp, token t) work item; 256 threads, every exit is workgroup-uniform
// COH (decode_mega.hip): q / K / V of the token come from other workgroups of the same launch and the results go to
// other workgroups of the same launch: those accesses are made device-coherent (dev_common.h) and the ticket is a relaxed
// atomic, instead of an agent-scope release / acquire whose cache-wide write-back and invalidate cost microseconds.
// D = 128 (default) or 64: head_dim; every lane role below is derived from it
template <int R, int TK, int TV, bool FUSED, bool COH = false, int D = 128>
__device__ __forceinline__ void flash_attn_decode_item(const AttnArgs &a, const float *cs_table, int n_rot, const DecodeFuse &fz, int g, int sp, int t) {
    constexpr int C = 64, NB = D / 32;
    // (a q4_0 cache - TK = TV = T_Q4_0, head_dim 128 - keeps 16 nibble bytes per 32-block: a piece is a whole block, four lanes per cell)
    constexpr int KROW = TK == T_F16 ? 2 * D : TK == T_Q4_0 ? D / 2 : D, VROW = TV == T_Q4_0 ? D / 2 : D;   // bytes of a K row; bytes (halfs for f16) of a V row
    constexpr int LPC = KROW / 16;                           // lanes per cell in the score pass (16-byte pieces of a K row)
    constexpr int KP = C * LPC / 256;                       // 16-byte K pieces per thread
    constexpr int DQ = D / 4, NCG = 256 / DQ, CPG = C / NCG;   // P.V pass: DQ lanes of 4 dims, NCG cell groups of CPG cells
    __shared__ __attribute__((aligned(16))) float qf[R * D];         // rotated q (f16-rounded for an f16 cache)
    __shared__ __attribute__((aligned(16))) int8_t qc[R * D];        // q8_0 codes of q
    __shared__ float qd[R * NB];
    __shared__ float S[R * C];
    __shared__ float ml[R * 2];
    __shared__ int vis[C];
    __shared__ uint32_t ksc[C * NB / 2], vsc[C * NB / 2];           // f16 block scales of the chunk
    __shared__ __attribute__((aligned(16))) float accs[NCG * R * D];
    __shared__ __attribute__((aligned(16))) uint8_t newk[D * 2], newv[D * 2];   // the token's own cache row (codes or f16)
    __shared__ uint32_t newkd[2], newvd[2];                                      // its f16 block scales, two per word
    __shared__ int last_flag;
    const int tid = tid_now(), lane = tid & 63, wave = tid >> 6;
    const int n_ctx = a.n_ctx, H = a.H;
    if (FUSED && fz.probe && tid == 0) fz.probe[2 * (sp * a.G + g)] = wall_clock64();
    int chunk = sp;
    if (a.tok_chunks) {                                        // walk this token's chunk list only (sequences own cache regions)
        if (sp >= a.tok_nchunks[t]) return;
        chunk = a.tok_chunks[(size_t)t * a.chunk_stride + sp];
    }
    const int c_lo = chunk * C;
    const size_t head_row0 = (size_t)g * n_ctx;

    // ---- issue every global load of the workgroup
    int cpos = -1;
    unsigned long long cseq = 0;
    if (tid < C && c_lo + tid < n_ctx) { cpos = a.cell_pos[c_lo + tid]; cseq = a.cell_seq[c_lo + tid]; }
    const int32_t tpos = a.tok_pos[t];
    const int tseq = a.tok_seq[t];
    if (a.T > 1) {   // (FUSED with T > 1 is the store-fused form of a batched step: no tickets, the merge is its own launch)
        // batched steps (one token per sequence): most chunks hold other sequences' cells only.  Decide that before
        // touching K / V: such a chunk publishes (m, l) = (-inf, 0) and leaves; the merge skips it.
        const int mine = (tid < C && cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
        if (!__syncthreads_or(mine)) {
            if (tid < R) {
                float *dst = a.part + (((size_t)t * H + (size_t)g * R + tid) * a.splits + sp) * (D + 2);
                dst[D] = -INFINITY; dst[D + 1] = 0.0f;
            }
            return;
        }
    }
    // q pairs: R * D / 2 pairs, pair pp -> (head r = pp / (D / 2), i = pp % (D / 2)); NORM pairing (2i, 2i+1)
    constexpr int HP = D / 2, NPAIR = R * HP, PPT = (NPAIR + 255) / 256;
    float2 qv[PPT], csv[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int pp = tid + 256 * j;
        if (pp < NPAIR) {
            const int r = pp / HP, i = pp % HP;
            if (fz.neox) {                                      // the pair's halves lie D / 2 apart
                const int o = (int)((((size_t)t * H + (size_t)g * R + r) * D + i) * 4);
                qv[j] = make_float2(__uint_as_float(cld4<COH>(a.q, o)), __uint_as_float(cld4<COH>(a.q, o + HP * 4)));
            } else {
                const coh_u32x2 qq = cld8<COH>(a.q, (int)((((size_t)t * H + (size_t)g * R + r) * D + 2 * i) * 4));
                qv[j] = make_float2(__uint_as_float(qq.x), __uint_as_float(qq.y));
            }
            csv[j] = (2 * i < n_rot) ? *reinterpret_cast<const float2 *>(cs_table + (size_t)t * n_rot + 2 * i) : make_float2(1.0f, 0.0f);
        }
    }
    uint4 kreg[KP];
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + 256 * j;
        int cell = c_lo + p / LPC;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TK == T_F16) kreg[j] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(a.kv.k) + rowi * D + (p % LPC) * 8);
        else kreg[j] = *reinterpret_cast<const uint4 *>(a.kv.k + rowi * KROW + (p % LPC) * 16);
    }
    const int dq = tid % DQ, cg = tid / DQ;
    uint2 vreg[CPG];
#pragma unroll
    for (int i = 0; i < CPG; i++) {
        int cell = c_lo + cg + NCG * i;
        if (cell >= n_ctx) cell = n_ctx - 1;
        const size_t rowi = head_row0 + cell;
        if (TV == T_F16) vreg[i] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(a.kv.v) + rowi * D + dq * 4);
        else if (TV == T_Q4_0) {            // dims 4 dq .. + 3 of block dq >> 3: the low or the high nibbles of four consecutive bytes, kept one per byte
            const int e0 = (dq * 4) & 31;
            const uint32_t w = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * VROW + (dq >> 3) * 16 + (e0 & 15));
            vreg[i].x = (e0 < 16 ? w : (w >> 4)) & 0x0f0f0f0fu; vreg[i].y = 0;
        }
        else { vreg[i].x = *reinterpret_cast<const uint32_t *>(a.kv.v + rowi * D + dq * 4); vreg[i].y = 0; }
    }
    uint32_t ks2 = 0, vs2 = 0;
    if (tid < C * NB / 2) {                                            // NB / 2 words (two f16 scales each) per cell
        int cell = c_lo + tid / (NB / 2);
        if (cell >= n_ctx) cell = n_ctx - 1;
        if (TK != T_F16) ks2 = *reinterpret_cast<const uint32_t *>(a.kv.kd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
        if (TV != T_F16) vs2 = *reinterpret_cast<const uint32_t *>(a.kv.vd + (head_row0 + cell) * NB + (tid % (NB / 2)) * 2);
    }

    // ---- FUSED: this token's K/V row (wave 0: lanes 0..31 rotate + convert K, lanes 32..63 convert V; 4 elements each)
    int own_cl = -1;                                   // chunk-local index of the token's cell, if this chunk holds it
    if (FUSED) {
        const int cellnew = fz.tok_cell[t];                      // (t = 0 in a single-token step)
        if (cellnew >= c_lo && cellnew < c_lo + C) own_cl = cellnew - c_lo;
        if (own_cl >= 0 && wave == 0 && (lane & 31) < DQ) {      // lanes 0 .. DQ - 1: K, lanes 32 .. 32 + DQ - 1: V (whole 8-lane groups either way)
            const bool isk = lane < 32;
            const int dd = (lane & 31) * 4;
            float4 x4;
            {
                const coh_u32x4 xx = cld16<COH>(isk ? fz.knew : fz.vnew, ((t * a.G + g) * D + dd) * 4);
                x4 = make_float4(__uint_as_float(xx.x), __uint_as_float(xx.y), __uint_as_float(xx.z), __uint_as_float(xx.w));
            }
            if (fz.neox) {
                // NEOX pairing: element e < D / 2 rotates with e + D / 2, which sits DQ / 2 lanes further on (both K lanes and V lanes run the exchange)
                const float p0 = __shfl_xor(x4.x, DQ / 2), p1 = __shfl_xor(x4.y, DQ / 2), p2 = __shfl_xor(x4.z, DQ / 2), p3 = __shfl_xor(x4.w, DQ / 2);
                if (isk) {
                    const bool lo = dd < HP;
                    const float *cp = cs_table + (size_t)t * n_rot + 2 * (lo ? dd : dd - HP);              // (c, s) of pairs dd' .. dd' + 3
                    const float4 ca = *reinterpret_cast<const float4 *>(cp), cb = *reinterpret_cast<const float4 *>(cp + 4);
                    if (lo) { x4.x = x4.x * ca.x - p0 * ca.y; x4.y = x4.y * ca.z - p1 * ca.w; x4.z = x4.z * cb.x - p2 * cb.y; x4.w = x4.w * cb.z - p3 * cb.w; }
                    else { x4.x = p0 * ca.y + x4.x * ca.x; x4.y = p1 * ca.w + x4.y * ca.z; x4.z = p2 * cb.y + x4.z * cb.x; x4.w = p3 * cb.w + x4.w * cb.z; }
                }
            } else if (isk && dd < n_rot) {
                const float4 cs = *reinterpret_cast<const float4 *>(cs_table + (size_t)t * n_rot + dd);   // c0 s0 c1 s1
                const float x0 = x4.x, x1 = x4.y, x2 = x4.z, x3 = x4.w;
                x4.x = x0 * cs.x - x1 * cs.y; x4.y = x0 * cs.y + x1 * cs.x;
                x4.z = x2 * cs.z - x3 * cs.w; x4.w = x2 * cs.w + x3 * cs.z;
            }
            const float xa[4] = {x4.x, x4.y, x4.z, x4.w};
            const size_t rowi = head_row0 + cellnew;
            const int TT = isk ? TK : TV;              // (TK, TV are compile-time; the select folds per branch below)
            uint32_t packed = 0; float dsc = 0.0f;
            if (TK == T_Q4_0) wave_quant_q40(xa, lane, packed, dsc);            // (TK = TV = T_Q4_0 only)
            else if (TK != T_F16 || TV != T_F16) wave_quant_q80(xa, packed, dsc);   // whole wave takes part (8-lane groups)
            if (TT == T_Q4_0) {
                const uint32_t hi = __shfl_xor(packed, 4, 64);                  // the lane four further on holds elements j + 16 of the block
                if ((lane & 4) == 0) {
                    const uint32_t bytes = packed | (hi << 4);
                    const int off = (dd >> 5) * 16 + (dd & 15);
                    *reinterpret_cast<uint32_t *>((isk ? newk : newv) + off) = bytes;
                    *reinterpret_cast<uint32_t *>((isk ? a.kv.k : a.kv.v) + rowi * (D / 2) + off) = bytes;
                }
                if ((lane & 7) == 0) {
                    const uint16_t hd = f2h(dsc);
                    (isk ? a.kv.kd : a.kv.vd)[rowi * NB + (dd >> 5)] = hd;
                    reinterpret_cast<uint16_t *>(isk ? newkd : newvd)[dd >> 5] = hd;
                }
            } else if (TT == T_F16) {
                uint2 o; o.x = (uint32_t)f2h(xa[0]) | ((uint32_t)f2h(xa[1]) << 16); o.y = (uint32_t)f2h(xa[2]) | ((uint32_t)f2h(xa[3]) << 16);
                *reinterpret_cast<uint2 *>((isk ? newk : newv) + dd * 2) = o;
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(isk ? a.kv.k : a.kv.v) + rowi * D + dd) = o;
            } else {
                *reinterpret_cast<uint32_t *>((isk ? newk : newv) + dd) = packed;
                *reinterpret_cast<uint32_t *>((isk ? a.kv.k : a.kv.v) + rowi * D + dd) = packed;
                if ((lane & 7) == 0) {
                    const uint16_t hd = f2h(dsc);
                    (isk ? a.kv.kd : a.kv.vd)[rowi * NB + (dd >> 5)] = hd;
                    reinterpret_cast<uint16_t *>(isk ? newkd : newvd)[dd >> 5] = hd;
                }
            }
        }
    }

    // ---- q: rotate, convert
    if (tid < C) vis[tid] = (cpos >= 0 && cpos <= tpos && ((cseq >> tseq) & 1ull)) ? 1 : 0;
    if (tid < C * NB / 2) { ksc[tid] = ks2; vsc[tid] = vs2; }
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        const int pp = tid + 256 * j;
        if (pp < NPAIR) {
            const float x0 = qv[j].x, x1 = qv[j].y, c = csv[j].x, s = csv[j].y;
            float y0 = x0 * c - x1 * s, y1 = x0 * s + x1 * c;
            if (TK == T_F16) { y0 = h2f(f2h(y0)); y1 = h2f(f2h(y1)); }
            if (fz.neox) { const int r = pp / HP, i = pp % HP; qf[r * D + i] = y0; qf[r * D + i + HP] = y1; }
            else *reinterpret_cast<float2 *>(qf + 2 * pp) = make_float2(y0, y1);
        }
    }
    __syncthreads();
    if (FUSED && own_cl >= 0) {   // workgroup-uniform: use the row just produced instead of what the cache held before
        if (tid < NB / 2) {
            if (TK != T_F16) ksc[own_cl * (NB / 2) + tid] = newkd[tid];
            if (TV != T_F16) vsc[own_cl * (NB / 2) + tid] = newvd[tid];
        }
#pragma unroll
        for (int j = 0; j < KP; j++) {
            const int p = tid + 256 * j;
            if (p / LPC == own_cl) kreg[j] = *reinterpret_cast<const uint4 *>(newk + (p % LPC) * 16);
        }
#pragma unroll
        for (int i = 0; i < CPG; i++) {
            if (cg + NCG * i == own_cl) {
                if (TV == T_F16) vreg[i] = *reinterpret_cast<const uint2 *>(newv + dq * 8);
                else if (TV == T_Q4_0) {
                    const int e0 = (dq * 4) & 31;
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(newv + (dq >> 3) * 16 + (e0 & 15));
                    vreg[i].x = (e0 < 16 ? w : (w >> 4)) & 0x0f0f0f0fu; vreg[i].y = 0;
                }
                else { vreg[i].x = *reinterpret_cast<const uint32_t *>(newv + dq * 4); vreg[i].y = 0; }
            }
        }
    }
    if (TK != T_F16) {   // q8_0 of the rotated q: 4 values per thread, 8-lane groups
        for (int e0 = tid * 4; e0 < R * D; e0 += 1024) {
            const float4 v4 = *reinterpret_cast<const float4 *>(qf + e0);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            uint32_t packed; float d;
            wave_quant_q80(vv, packed, d);
            *reinterpret_cast<uint32_t *>(qc + e0) = packed;
            if ((lane & 7) == 0) qd[e0 >> 5] = h2f(f2h(d));
        }
        __syncthreads();
    }

    // ---- scores
#pragma unroll
    for (int j = 0; j < KP; j++) {
        const int p = tid + 256 * j;
        const int cl = p / LPC, piece = p % LPC;
        float sc[R];
        if (TK == T_F16) {
            const uint32_t kw[4] = {kreg[j].x, kreg[j].y, kreg[j].z, kreg[j].w};
#pragma unroll
            for (int r = 0; r < R; r++) {
                const float *qq = qf + r * D + piece * 8;
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    s += h2f((uint16_t)(kw[i] & 0xffff)) * qq[2 * i];
                    s += h2f((uint16_t)(kw[i] >> 16)) * qq[2 * i + 1];
                }
                s += dpp_f<DPP_QP_1032>(s); s += dpp_f<DPP_QP_2301>(s); s += dpp_f<DPP_HALF_MIRROR>(s);      // 8 lanes
                if (LPC == 16) s += dpp_f<DPP_MIRROR>(s);
                sc[r] = s;
            }
        } else if (TK == T_Q4_0) {   // vec_dot_q4_0_q8_0: piece = 32-block; codes nibble - 8 against the q8_0 codes of q, low nibbles = elements 0 .. 15
            const uint32_t kpair = ksc[cl * (NB / 2) + (piece >> 1)];
            const float dk = h2f((uint16_t)((piece & 1) ? (kpair >> 16) : (kpair & 0xffff)));
            const uint32_t kw[4] = {kreg[j].x, kreg[j].y, kreg[j].z, kreg[j].w};
            uint32_t klo[4], khi[4];
#pragma unroll
            for (int w = 0; w < 4; w++) { klo[w] = nib_to_i8x4(kw[w] & 0x0f0f0f0fu); khi[w] = nib_to_i8x4((kw[w] >> 4) & 0x0f0f0f0fu); }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint4 q0 = *reinterpret_cast<const uint4 *>(qc + r * D + piece * 32), q1 = *reinterpret_cast<const uint4 *>(qc + r * D + piece * 32 + 16);
                int s = 0;
                s = dot4(klo[0], q0.x, s); s = dot4(klo[1], q0.y, s); s = dot4(klo[2], q0.z, s); s = dot4(klo[3], q0.w, s);
                s = dot4(khi[0], q1.x, s); s = dot4(khi[1], q1.y, s); s = dot4(khi[2], q1.z, s); s = dot4(khi[3], q1.w, s);
                float f = (float)s * (dk * qd[r * NB + piece]);
                f += dpp_f<DPP_QP_1032>(f); f += dpp_f<DPP_QP_2301>(f);                                 // the cell's four blocks
                sc[r] = f;
            }
        } else {
            const uint32_t kpair = ksc[cl * (NB / 2) + (piece >> 2)];
            const float dk = h2f((uint16_t)(((piece >> 1) & 1) ? (kpair >> 16) : (kpair & 0xffff)));
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint4 qq = *reinterpret_cast<const uint4 *>(qc + r * D + piece * 16);
                int s = 0;
                s = dot4(kreg[j].x, qq.x, s); s = dot4(kreg[j].y, qq.y, s); s = dot4(kreg[j].z, qq.z, s); s = dot4(kreg[j].w, qq.w, s);
                s += dpp_i<DPP_QP_1032>(s);                    // both halves of the 32-block (integer)
                float f = (piece & 1) ? 0.0f : (float)s * (dk * qd[r * NB + (piece >> 1)]);
                f += dpp_f<DPP_QP_1032>(f); f += dpp_f<DPP_QP_2301>(f);                                 // 4 lanes
                if (LPC == 8) f += dpp_f<DPP_HALF_MIRROR>(f);                                            // 8 lanes
                sc[r] = f;
            }
        }
        if (piece == 0) {
            const bool v = vis[cl] != 0;
#pragma unroll
            for (int r = 0; r < R; r++) S[r * C + cl] = v ? sc[r] * a.scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- softmax of the chunk: wave w -> heads w, w+4, ...; lane = cell
    for (int r = wave; r < R; r += 4) {
        const float s = S[r * C + lane];
        const float m = wave_max(s);
        const float p = (s == -INFINITY) ? 0.0f : expf(s - m);
        const float l = wave_sum(p);
        S[r * C + lane] = p;
        if (lane == 0) { ml[2 * r] = m; ml[2 * r + 1] = l; }
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
            const uint32_t vpair = vsc[cl * (NB / 2) + (dq >> 4)];
            const float dv = h2f((uint16_t)(((dq >> 3) & 1) ? (vpair >> 16) : (vpair & 0xffff)));
            const uint32_t w = TV == T_Q4_0 ? nib_to_i8x4(vreg[i].x) : vreg[i].x;
            v4[0] = (float)(int8_t)(w & 0xff) * dv; v4[1] = (float)(int8_t)((w >> 8) & 0xff) * dv;
            v4[2] = (float)(int8_t)((w >> 16) & 0xff) * dv; v4[3] = (float)(int8_t)(w >> 24) * dv;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const float p = S[r * C + cl];
            acc[r][0] += v4[0] * p; acc[r][1] += v4[1] * p; acc[r][2] += v4[2] * p; acc[r][3] += v4[3] * p;
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++)
        *reinterpret_cast<float4 *>(accs + ((size_t)cg * R + r) * D + dq * 4) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    __syncthreads();
    for (int e = tid; e < R * D; e += 256) {
        const int r = e / D, d = e - r * D;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < NCG; j++) s += accs[((size_t)j * R + r) * D + d];
        float *dst = a.part + (((size_t)t * H + (size_t)g * R + r) * a.splits + sp) * (D + 2);
        cstf<COH>(dst + d, s);
        if (d == 0) { cstf<COH>(dst + D, ml[2 * r]); cstf<COH>(dst + D + 1, ml[2 * r + 1]); }
    }
    if (!FUSED) return;
    if (fz.counters == nullptr) return;               // store-fused only: the merge runs as its own launch
    if (fz.probe && tid == 0) fz.probe[2 * (sp * a.G + g) + 1] = wall_clock64();

    // ---- FUSED: ticket; the last workgroup of this kv head merges.  The merge also quantises whole 256-element blocks of
    // the attention output, so with one 128-wide query head per kv head (R = 1, multi-head attention) TWO neighbouring kv
    // heads share a ticket and the last arriver of the pair merges both: GP kv heads per ticket, RM = GP * R heads merged.
    // (in general the smallest number of kv heads whose R * D outputs fill whole blocks: 256 / gcd(R * D, 256) - two for 3, 5 or 7 query heads of 128 per kv head)
    constexpr int GP = 256 / ((R * D) % 256 == 0 ? 256 : (R * D) % 128 == 0 ? 128 : (R * D) % 64 == 0 ? 64 : 32);
    constexpr int RM = R * GP;
    const int gq = g / GP;                                     // ticket group
    const int hb = gq * RM;                                    // its first query head
    const int stride_s = a.splits;                             // workspace stride; with a chunk list fewer slots are in use
    const int splits = a.tok_nchunks ? a.tok_nchunks[0] : a.splits;
    // every wave's partial stores must have reached L2 before thread 0 releases them device-wide: __syncthreads() fences
    // LDS only (hipcc emitted no vmcnt wait before the barrier here), so each wave drains its own stores first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a workgroup-scope release fence compiles to nothing on this target)
    __syncthreads();
    if (tid == 0) {
        // ONE release per workgroup (the barrier ordered the other waves' stores before it), then the ticket
        const unsigned old = COH ? __hip_atomic_fetch_add(fz.counters + gq * ATT_SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                 : __hip_atomic_fetch_add(fz.counters + gq * ATT_SYNC_STRIDE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (old == (unsigned)(splits * GP) - 1u) ? 1 : 0;
        if (last_flag) __hip_atomic_store(fz.counters + gq * ATT_SYNC_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    if (!last_flag) return;
    if (fz.probe && tid == 0) fz.probe[4096 + 8 * g] = wall_clock64();
    float *merged = accs;                              // [RM * D]
    float *wgt = GP > 1 ? accs + RM * D : S;           // [RM][64] split weights (S is free now, but holds R * C floats only)
    // The merge is a chain of memory round trips (~1.5 us each: the partials come from other XCDs).  Every output still
    // sums its splits in order (the combine kernel's arithmetic), but the partials of the first 32 splits of all of this
    // thread's outputs are requested BEFORE the (m, l) pairs, so the weight computation runs under them: ticket, one trip
    // for (m, l) + 32 splits, one more per further 32 splits (it was ticket + (m, l) + one trip per 8 splits and output).
    constexpr int NE = (RM * D + 255) / 256, UB = NE > 4 ? 16 : 32;     // (NE * UB partials in registers per thread)
    float macc[NE], x[NE][UB];
    const float *pp[NE];
    int wr[NE];
#pragma unroll
    for (int j = 0; j < NE; j++) {
        const int e = tid + 256 * j < RM * D ? tid + 256 * j : 0, r = e / D, d = e - r * D;
        pp[j] = a.part + ((size_t)hb + r) * stride_s * (D + 2) + d;
        wr[j] = r * 64;
        macc[j] = 0.0f;
    }
    auto request = [&](int s0) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int s2 = s0 + u < splits ? s0 + u : splits - 1;              // clamped: straight-line loads
#pragma unroll
            for (int j = 0; j < NE; j++) x[j][u] = __hip_atomic_load(pp[j] + (size_t)s2 * (D + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    request(0);
    for (int r = wave; r < RM; r += 4) {               // same arithmetic as flash_attn_combine_kernel
        const float *p = a.part + ((size_t)hb + r) * stride_s * (D + 2);
        float m = -INFINITY, l = 0.0f;
        if (lane < splits) {
            m = __hip_atomic_load(p + (size_t)lane * (D + 2) + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            l = __hip_atomic_load(p + (size_t)lane * (D + 2) + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float M = wave_max(m);
        const float w = (lane < splits && m != -INFINITY) ? expf(m - M) : 0.0f;
        const float den = wave_sum(w * l);
        const float inv = 1.0f / den;
        wgt[r * 64 + lane] = w * inv;
    }
    __syncthreads();
    if (fz.probe && tid == 0) fz.probe[4096 + 8 * g + 1] = wall_clock64();
    const int E = H * D;
    for (int s0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            if (s0 + u < splits) {                                             // workgroup-uniform
#pragma unroll
                for (int j = 0; j < NE; j++) macc[j] += wgt[wr[j] + s0 + u] * x[j][u];
            }
        }
        s0 += UB;
        if (s0 >= splits) break;
        request(s0);
    }
#pragma unroll
    for (int j = 0; j < NE; j++) {
        const int e = tid + 256 * j;
        if (e < RM * D) { merged[e] = macc[j]; a.out[(size_t)hb * D + e] = macc[j]; }
    }
    if (fz.probe && tid == 0) fz.probe[4096 + 8 * g + 2] = wall_clock64();
    __syncthreads();
    constexpr int NBLK = (RM * D) >> 8;                // 256-blocks this ticket group owns in the H*D row
    if (wave < NBLK && (fz.want_q8k || fz.want_q80)) {
        for (int b = wave; b < NBLK; b += 4) {
            const float4 v4 = *reinterpret_cast<const float4 *>(merged + b * 256 + lane * 4);
            const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
            const int gb = ((hb * D) >> 8) + b;        // global block index
            const int e0 = gb * 256 + lane * 4;
            if (fz.want_q8k) {
                uint32_t packed; int bs; float dq8;
                wave_quant_q8k(vv, lane, packed, bs, dq8);
                cst4<COH>(fz.q.qs + e0, packed);
                if ((lane & 3) == 0) cst2<COH>(fz.q.bsums + gb * 16 + (lane >> 2), (unsigned short)(int16_t)bs);
                if (lane == 0) cstf<COH>(fz.q.d + gb, dq8);
            }
            if (fz.want_q80) {
                uint32_t packed; float dd;
                wave_quant_q80(vv, packed, dd);
                cst4<COH>(fz.q.qs0 + e0, packed);
                if ((lane & 7) == 0) cst2<COH>(fz.q.d0 + gb * 8 + (lane >> 3), f2h(dd));
            }
        }
    }
    (void)E;
    if (fz.probe) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) fz.probe[4096 + 8 * g + 3] = wall_clock64();
    }
}

template <int R, int TK, int TV, bool FUSED, bool COH = false, int D = 128>
__global__ __launch_bounds__(256) void flash_attn_decode_kernel(const AttnArgs a, const float *cs_table, int n_rot, const DecodeFuse fz) {
    flash_attn_decode_item<R, TK, TV, FUSED, COH, D>(a, cs_table, n_rot, fz, blockIdx.x, blockIdx.y, blockIdx.z);
}

}  // namespace mi355
