// mmq.hip — batched prefill contraction  Y[t][r] = dot(W[r,:], act_q8K[t,:])  on the MFMA matrix cores.
//
// Stands in for ggml's mul_mat_q / the CPU backend's chunked mul_mat for prompt processing,
// reached from the reference through llama_decode with n_tokens > 1 (src/llama_server_context.cc:1568-1607, 1635).
//
// Integer-exact by construction, and MFMA-bound rather than VALU-bound:
//   ggml_vec_dot_q{4,5,6}_K_q8_K computes, per (row, token, super-block of 256),   isum = sum_g sc_g * sum_{k in g} q_k a_k
//   with 6-bit (Q4_K/Q5_K, groups of 32) or int8 (Q6_K, groups of 16) group scales sc_g.  Applying sc_g to the int32
//   MFMA result costs one multiply-add per OUTPUT per group (16 VALU ops per MFMA — as long as the MFMA itself).  Here
//   the scale is folded into the WEIGHT instead:  p_k = sc_g * q_k  (10..13 bits) is split into two int8 planes
//   p_k = 2^S * hi_k + lo_k, and  isum = 2^S * (hi . a) + (lo . a)  is two MFMA accumulation chains over the whole
//   super-block with nothing in between.  The split is exact, so isum is bit-identical to the CPU's integer.
//   The expansion costs VALU work per WEIGHT (not per output) and is amortised over the wave's token tiles.
//   * Q4_K / Q5_K: q in [0,15] / [0,31], sc in [0,63]: S = 5, hi <= 29 / 61, lo <= 31.  The "mins" term
//     msum = sum_j m_j * bsum_j is one more pair of small MFMAs over the 16 per-16 block sums of Q8_K (m_j duplicated),
//     the int16 sums split into int8 planes bsum = 64 * bh + bl by mmq_prep_kernel.
//   * Q6_K: the unsigned code q in [0,63] times the signed int8 scale (16-bit signed lanes), S = 6, hi in [-128,126],
//     lo in [0,63]; the "-32" of the code is the same small MFMA:  isum = 64 H + L - 32 * sum_g sc_g * bsum16_g.
//     A lane's 16 weights of a K-step are exactly one 16-group, so the scale is a per-lane scalar.
//   Only the final f32 accumulation order over super-blocks differs from the CPU.
//
// Tile mapping (wave64, v_mfma_i32_32x32x32_i8): M = tokens (A operand from LDS), N = weight rows (B operand built in
// registers): lane = (row n = lane & 31, k-half kg = lane >> 5) holds its own row's 16 weights per K-step, so d / dmin /
// scales are per-lane scalars and weights go global -> registers with no LDS round trip.  Workgroup = 8 waves:
// RW row-waves x TW token-waves with MT token tiles of 32 per wave (MT = 4: 256 rows x 128 tokens; MT = 2: 128 x 128;
// MT = 1: 256 rows x 32 tokens).  The activation tile of one super-block (tokens x 256 int8, + block-sum planes + scales) is
// staged in LDS once per workgroup (row stride 272 B: ds_read_b128 conflict-free), the next tile's global loads are
// issued before the MFMA work of the current one, and the raw weights of the next super-block likewise.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace mi355 {

namespace {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int A_STRIDE = 272;          // 256 codes + 16 B pad per token row
constexpr int NTHREADS = 512;

__device__ __forceinline__ i32x16 mfma_i8(i32x4 a, i32x4 b, i32x16 c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ u32x4 ldg16(const void *p) { return *reinterpret_cast<const u32x4 *>(p); }
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// two int8 planes of  p = scale * code  for four codes held one per byte in `w` (codes < 64, 0 <= scale < 64 for the
// unsigned form): even / odd bytes are multiplied as 16-bit lanes of one 32-bit product (no carry: p < 2^16)
template <int S>
__device__ __forceinline__ void planes_u(uint32_t w, uint32_t sc, uint32_t &hi, uint32_t &lo) {
    constexpr uint32_t LM = ((1u << S) - 1u) * 0x00010001u;
    const uint32_t pe = (w & 0x00ff00ffu) * sc;               // products of bytes 0, 2
    const uint32_t po = ((w >> 8) & 0x00ff00ffu) * sc;        // products of bytes 1, 3
    hi = ((pe >> S) & 0x00ff00ffu) | (((po >> S) & 0x00ff00ffu) << 8);
    lo = (pe & LM) | ((po & LM) << 8);
}

typedef short i16x2 __attribute__((ext_vector_type(2)));
// signed scale (int8, replicated in both 16-bit lanes of scpk): p = sc * code as 16-bit signed lanes, hi = p >> 6 (floor), lo = p & 63
__device__ __forceinline__ void planes_s6(uint32_t w, uint32_t scpk, uint32_t &hi, uint32_t &lo) {
    const i16x2 e = __builtin_bit_cast(i16x2, w & 0x00ff00ffu), o = __builtin_bit_cast(i16x2, (w >> 8) & 0x00ff00ffu);
    const i16x2 s = __builtin_bit_cast(i16x2, scpk);
    const i16x2 pe = e * s, po = o * s;                       // v_pk_mul_lo_u16: low 16 bits = the signed product
    const i16x2 six = {6, 6};
    const uint32_t he = __builtin_bit_cast(uint32_t, pe >> six), ho = __builtin_bit_cast(uint32_t, po >> six);   // v_pk_ashrrev_i16
    const uint32_t le = __builtin_bit_cast(uint32_t, pe) & 0x003f003fu, lo_ = __builtin_bit_cast(uint32_t, po) & 0x003f003fu;
    hi = (he & 0x00ff00ffu) | ((ho & 0x00ff00ffu) << 8);
    lo = le | (lo_ << 8);
}

// the same for the signed Q6_K code (q - 32) in [-32, 31]
__device__ __forceinline__ void planes_s6m32(uint32_t w, uint32_t scpk, uint32_t &hi, uint32_t &lo) {
    const i16x2 k32 = {32, 32};
    const i16x2 e = __builtin_bit_cast(i16x2, w & 0x00ff00ffu) - k32, o = __builtin_bit_cast(i16x2, (w >> 8) & 0x00ff00ffu) - k32;
    const i16x2 s = __builtin_bit_cast(i16x2, scpk);
    const i16x2 pe = e * s, po = o * s;
    const i16x2 six = {6, 6};
    const uint32_t he = __builtin_bit_cast(uint32_t, pe >> six), ho = __builtin_bit_cast(uint32_t, po >> six);
    const uint32_t le = __builtin_bit_cast(uint32_t, pe) & 0x003f003fu, lo_ = __builtin_bit_cast(uint32_t, po) & 0x003f003fu;
    hi = (he & 0x00ff00ffu) | ((ho & 0x00ff00ffu) << 8);
    lo = le | (lo_ << 8);
}

// per-lane weight-row state of one super-block: raw bytes as loaded, then what the K-steps need
template <int TYPE> struct RowSB;

template <> struct RowSB<T_Q4_K> {
    u32x4 q0, q1, q2, q3;      // chunk c: 16 bytes = 16 low nibbles (sub-block 2c) and 16 high nibbles (2c+1) of this lane's k-half
    u32x4 h;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg) {
        const uint8_t *b = row + (size_t)sb * 144;
        h = ldg16(b);
        q0 = ldg16(b + 16 + 16 * kg); q1 = ldg16(b + 48 + 16 * kg); q2 = ldg16(b + 80 + 16 * kg); q3 = ldg16(b + 112 + 16 * kg);
    }
    __device__ __forceinline__ float d() const { return h2f((uint16_t)(h.x & 0xffff)); }
    __device__ __forceinline__ float dmin() const { return h2f((uint16_t)(h.x >> 16)); }
    // scales bytes s0..s11 in h.y h.z h.w; j<4: sc = s[j]&63, m = s[j+4]&63; j>=4: sc = (s[j+4]&15)|((s[j-4]>>6)<<4), m = (s[j+4]>>4)|((s[j]>>6)<<4)
    __device__ __forceinline__ void scales(uint32_t &sc_lo, uint32_t &sc_hi, uint32_t &mn_lo, uint32_t &mn_hi) const {
        sc_lo = h.y & 0x3f3f3f3f; mn_lo = h.z & 0x3f3f3f3f;
        sc_hi = (h.w & 0x0f0f0f0f) | ((h.y >> 2) & 0x30303030);
        mn_hi = ((h.w >> 4) & 0x0f0f0f0f) | ((h.z >> 2) & 0x30303030);
    }
    // the raw codes of group J (one per byte): the B operand of the group-scale form (scale applied to the MFMA result)
    template <int J> __device__ __forceinline__ i32x4 codes() const {
        const u32x4 v = (J >> 1) == 0 ? q0 : (J >> 1) == 1 ? q1 : (J >> 1) == 2 ? q2 : q3;
        constexpr int sh = (J & 1) * 4;
        return i32x4{(int)((v.x >> sh) & 0x0f0f0f0f), (int)((v.y >> sh) & 0x0f0f0f0f), (int)((v.z >> sh) & 0x0f0f0f0f), (int)((v.w >> sh) & 0x0f0f0f0f)};
    }
    template <int J> __device__ __forceinline__ void bop(uint32_t sc, i32x4 &bh, i32x4 &bl) const {
        const u32x4 v = (J >> 1) == 0 ? q0 : (J >> 1) == 1 ? q1 : (J >> 1) == 2 ? q2 : q3;
        constexpr int sh = (J & 1) * 4;
        uint32_t hx, lx;
        planes_u<5>((v.x >> sh) & 0x0f0f0f0f, sc, hx, lx); bh.x = (int)hx; bl.x = (int)lx;
        planes_u<5>((v.y >> sh) & 0x0f0f0f0f, sc, hx, lx); bh.y = (int)hx; bl.y = (int)lx;
        planes_u<5>((v.z >> sh) & 0x0f0f0f0f, sc, hx, lx); bh.z = (int)hx; bl.z = (int)lx;
        planes_u<5>((v.w >> sh) & 0x0f0f0f0f, sc, hx, lx); bh.w = (int)hx; bl.w = (int)lx;
    }
};

template <> struct RowSB<T_Q5_K> {
    u32x4 q0, q1, q2, q3, qh;
    u32x4 h;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg) {
        const uint8_t *b = row + (size_t)sb * 176;
        h = ldg16(b);
        qh = ldg16(b + 16 + 16 * kg);
        q0 = ldg16(b + 48 + 16 * kg); q1 = ldg16(b + 80 + 16 * kg); q2 = ldg16(b + 112 + 16 * kg); q3 = ldg16(b + 144 + 16 * kg);
    }
    __device__ __forceinline__ float d() const { return h2f((uint16_t)(h.x & 0xffff)); }
    __device__ __forceinline__ float dmin() const { return h2f((uint16_t)(h.x >> 16)); }
    __device__ __forceinline__ void scales(uint32_t &sc_lo, uint32_t &sc_hi, uint32_t &mn_lo, uint32_t &mn_hi) const {
        sc_lo = h.y & 0x3f3f3f3f; mn_lo = h.z & 0x3f3f3f3f;
        sc_hi = (h.w & 0x0f0f0f0f) | ((h.y >> 2) & 0x30303030);
        mn_hi = ((h.w >> 4) & 0x0f0f0f0f) | ((h.z >> 2) & 0x30303030);
    }
    template <int J> __device__ __forceinline__ i32x4 codes() const {
        const u32x4 v = (J >> 1) == 0 ? q0 : (J >> 1) == 1 ? q1 : (J >> 1) == 2 ? q2 : q3;
        constexpr int sh = (J & 1) * 4;
#define Q5C(vv, hh) (int)((((vv) >> sh) & 0x0f0f0f0f) | ((((hh) >> J) & 0x01010101u) << 4))
        return i32x4{Q5C(v.x, qh.x), Q5C(v.y, qh.y), Q5C(v.z, qh.z), Q5C(v.w, qh.w)};
#undef Q5C
    }
    template <int J> __device__ __forceinline__ void bop(uint32_t sc, i32x4 &bh, i32x4 &bl) const {
        const u32x4 v = (J >> 1) == 0 ? q0 : (J >> 1) == 1 ? q1 : (J >> 1) == 2 ? q2 : q3;
        constexpr int sh = (J & 1) * 4;
        uint32_t hx, lx;
#define Q5W(vv, hh) ((((vv) >> sh) & 0x0f0f0f0f) | ((((hh) >> J) & 0x01010101u) << 4))
        planes_u<5>(Q5W(v.x, qh.x), sc, hx, lx); bh.x = (int)hx; bl.x = (int)lx;
        planes_u<5>(Q5W(v.y, qh.y), sc, hx, lx); bh.y = (int)hx; bl.y = (int)lx;
        planes_u<5>(Q5W(v.z, qh.z), sc, hx, lx); bh.z = (int)hx; bl.z = (int)lx;
        planes_u<5>(Q5W(v.w, qh.w), sc, hx, lx); bh.w = (int)hx; bl.w = (int)lx;
#undef Q5W
    }
};

// Q6_K device row planes: ql | qh | scales | d.  K-step s (0..7) = half n = s>>2, quarter k = s&3: this lane's 16 weights
// l = 16kg.. of that quarter are one 16-group with scale sc[8n + 2k + kg].
template <> struct RowSB<T_Q6_K> {
    u32x4 l0, l1, l2, l3;      // ql: half 0 (l, l+32), half 1 (l, l+32)
    u32x4 h0, h1;              // qh: half 0, half 1
    u32x4 scv;                 // 16 int8 scales
    uint32_t d16;
    __device__ __forceinline__ void load(const uint8_t *row, int nb, int sb, int kg) {
        l0 = ldg16(row + (size_t)sb * 128 + 16 * kg);      l1 = ldg16(row + (size_t)sb * 128 + 32 + 16 * kg);
        l2 = ldg16(row + (size_t)sb * 128 + 64 + 16 * kg); l3 = ldg16(row + (size_t)sb * 128 + 96 + 16 * kg);
        h0 = ldg16(row + (size_t)nb * 128 + (size_t)sb * 64 + 16 * kg);
        h1 = ldg16(row + (size_t)nb * 128 + (size_t)sb * 64 + 32 + 16 * kg);
        scv = ldg16(row + (size_t)nb * 192 + (size_t)sb * 16);
        d16 = *reinterpret_cast<const uint16_t *>(row + (size_t)nb * 208 + (size_t)sb * 2);
    }
    __device__ __forceinline__ float d() const { return h2f((uint16_t)d16); }
    // signed scale of 16-group g, replicated into both 16-bit lanes
    __device__ __forceinline__ uint32_t scale_pk(int g) const {
        const uint32_t w = g < 4 ? scv.x : g < 8 ? scv.y : g < 12 ? scv.z : scv.w;
        const int s = (int)(int8_t)((w >> (8 * (g & 3))) & 0xff);
        return ((uint32_t)s & 0xffffu) | ((uint32_t)s << 16);
    }
    // same with the signed code (q - 32): 64 * hi + lo = sc * (q - 32), no block-sum correction needed afterwards
    template <int S> __device__ __forceinline__ void bop_signed(uint32_t scpk, i32x4 &bh, i32x4 &bl) const {
        constexpr int n = S >> 2, k = S & 3;
        const u32x4 lv = n == 0 ? ((k & 1) ? l1 : l0) : ((k & 1) ? l3 : l2);
        const u32x4 hv = n == 0 ? h0 : h1;
        constexpr int nsh = (k >> 1) * 4, hsh = 2 * k;
        uint32_t hx, lx;
#define Q6W(lw, hw) ((((lw) >> nsh) & 0x0f0f0f0f) | ((((hw) >> hsh) & 0x03030303u) << 4))
        planes_s6m32(Q6W(lv.x, hv.x), scpk, hx, lx); bh.x = (int)hx; bl.x = (int)lx;
        planes_s6m32(Q6W(lv.y, hv.y), scpk, hx, lx); bh.y = (int)hx; bl.y = (int)lx;
        planes_s6m32(Q6W(lv.z, hv.z), scpk, hx, lx); bh.z = (int)hx; bl.z = (int)lx;
        planes_s6m32(Q6W(lv.w, hv.w), scpk, hx, lx); bh.w = (int)hx; bl.w = (int)lx;
#undef Q6W
    }
    template <int S> __device__ __forceinline__ void bop(uint32_t scpk, i32x4 &bh, i32x4 &bl) const {
        constexpr int n = S >> 2, k = S & 3;
        const u32x4 lv = n == 0 ? ((k & 1) ? l1 : l0) : ((k & 1) ? l3 : l2);
        const u32x4 hv = n == 0 ? h0 : h1;
        constexpr int nsh = (k >> 1) * 4, hsh = 2 * k;
        uint32_t hx, lx;
#define Q6W(lw, hw) ((((lw) >> nsh) & 0x0f0f0f0f) | ((((hw) >> hsh) & 0x03030303u) << 4))
        planes_s6(Q6W(lv.x, hv.x), scpk, hx, lx); bh.x = (int)hx; bl.x = (int)lx;
        planes_s6(Q6W(lv.y, hv.y), scpk, hx, lx); bh.y = (int)hx; bl.y = (int)lx;
        planes_s6(Q6W(lv.z, hv.z), scpk, hx, lx); bh.z = (int)hx; bl.z = (int)lx;
        planes_s6(Q6W(lv.w, hv.w), scpk, hx, lx); bh.w = (int)hx; bl.w = (int)lx;
#undef Q6W
    }
};

// MT token tiles of 32 per wave; TW token-waves x RW row-waves = 8 waves
template <int MT> struct Geo {
    static constexpr int TW = MT == 2 ? 2 : 1, RW = 8 / TW;
    static constexpr int TOK_TILE = 32 * MT * TW, ROW_TILE = 32 * RW;
    static constexpr int LDS_A = TOK_TILE * A_STRIDE, LDS_P = TOK_TILE * 16;
    static constexpr int LDS_BYTES = LDS_A + 2 * LDS_P + TOK_TILE * 4;
    static constexpr int NST = (TOK_TILE * 16 + NTHREADS - 1) / NTHREADS;       // 16-byte staging pieces per thread
};

template <int TYPE, int MT>
__global__ __launch_bounds__(NTHREADS) void mmq_kernel(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, int n_row_tiles, int n_tok_tiles,
                                                       const int8_t *aq, const float *ad, const int8_t *abh, const int8_t *abl,
                                                       float *out, int ld_out, const float *resid) {
    using G = Geo<MT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int8_t *s_aq = reinterpret_cast<int8_t *>(smem);
    int8_t *s_bh = s_aq + G::LDS_A, *s_bl = s_bh + G::LDS_P;
    float *s_yd = reinterpret_cast<float *>(s_bl + G::LDS_P);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = K >> 8;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs; the token tiles of one row tile run on ONE
    // XCD back to back, so its weights are fetched from HBM once and re-read from that XCD's L2
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tok_tile = loc % n_tok_tiles;
    const int row_tile = (loc / n_tok_tiles) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;                       // workgroup-uniform
    const int rw = wave % G::RW, tw = wave / G::RW;
    const int row0 = row_tile * G::ROW_TILE + rw * 32;
    const int tok0 = tok_tile * G::TOK_TILE;
    const int n = lane & 31, kg = lane >> 5;
    int my_row = row0 + n;
    const bool row_ok = my_row < n_rows;
    if (!row_ok) my_row = n_rows - 1;
    const uint8_t *rowp = W + (size_t)my_row * row_bytes;

    // staging roles: piece p = tid + NTHREADS * i -> token p / 16, 16-byte column p % 16; tokens past T re-read token T-1
    unsigned st_g[G::NST], st_l[G::NST];
#pragma unroll
    for (int i = 0; i < G::NST; i++) {
        const int p = tid + NTHREADS * i;
        int tk = p >> 4;
        if (tk >= G::TOK_TILE) tk = G::TOK_TILE - 1;
        int gt = tok0 + tk;
        if (gt >= T) gt = T - 1;
        st_g[i] = (unsigned)gt * (unsigned)K + (unsigned)(p & 15) * 16u;
        st_l[i] = (unsigned)tk * A_STRIDE + (unsigned)(p & 15) * 16u;
    }
    int ptok = tok0 + (tid < G::TOK_TILE ? tid : 0);
    if (ptok >= T) ptok = T - 1;
    const unsigned pl_g = (unsigned)ptok * (unsigned)nb;      // per-token plane / scale index base (x 16 bytes for planes)

    float facc[MT][16];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) facc[t][r] = 0.0f;

    // ---- prologue: tile 0 into LDS, raw weights of super-block 0 into registers
    RowSB<TYPE> R;
    R.load(rowp, nb, 0, kg);
    {
        u32x4 tmp[G::NST];
#pragma unroll
        for (int i = 0; i < G::NST; i++) tmp[i] = ldg16(aq + st_g[i]);
#pragma unroll
        for (int i = 0; i < G::NST; i++) if (tid + NTHREADS * i < G::TOK_TILE * 16) *reinterpret_cast<u32x4 *>(s_aq + st_l[i]) = tmp[i];
        if (tid < G::TOK_TILE) {
            s_yd[tid] = ad[pl_g];
            *reinterpret_cast<u32x4 *>(s_bh + tid * 16) = ldg16(abh + (size_t)pl_g * 16);
            *reinterpret_cast<u32x4 *>(s_bl + tid * 16) = ldg16(abl + (size_t)pl_g * 16);
        }
    }
    __syncthreads();

    for (int sb = 0; sb < nb; sb++) {
        // ---- next activation tile: request now, park in registers across the MFMA work
        const bool more = sb + 1 < nb;
        const int sbn = more ? sb + 1 : sb;
        u32x4 nxt[G::NST];
#pragma unroll
        for (int i = 0; i < G::NST; i++) nxt[i] = ldg16(aq + st_g[i] + (unsigned)sbn * 256u);
        float nyd = 0.0f;
        u32x4 nbh = {0, 0, 0, 0}, nbl = {0, 0, 0, 0};
        if (tid < G::TOK_TILE) {
            nyd = ad[pl_g + (unsigned)sbn];
            nbh = ldg16(abh + ((size_t)pl_g + (unsigned)sbn) * 16);
            nbl = ldg16(abl + ((size_t)pl_g + (unsigned)sbn) * 16);
        }

        // ---- this super-block: scales, then 8 K-steps x MT tiles x {hi, lo}
        i32x16 H[MT], L[MT];
        const int8_t *abase = s_aq + (tw * MT * 32 + n) * A_STRIDE + 16 * kg;
        uint32_t sc_lo = 0, sc_hi = 0, mn_lo = 0, mn_hi = 0;
        if constexpr (TYPE != T_Q6_K) R.scales(sc_lo, sc_hi, mn_lo, mn_hi);
#define KSTEP(J)                                                                                              \
        {                                                                                                     \
            i32x4 bh, bl;                                                                                     \
            if constexpr (TYPE == T_Q6_K) R.template bop<J>(R.scale_pk(8 * (J >> 2) + 2 * (J & 3) + kg), bh, bl); \
            else R.template bop<J>(((J < 4 ? sc_lo : sc_hi) >> (8 * (J & 3))) & 0xff, bh, bl);                \
            _Pragma("unroll") for (int t = 0; t < MT; t++) {                                                  \
                const i32x4 a = *reinterpret_cast<const i32x4 *>(abase + t * 32 * A_STRIDE + 32 * J);          \
                if (J == 0) {                                                                                 \
                    i32x16 z;                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 16; r++) z[r] = 0;                                  \
                    H[t] = mfma_i8(a, bh, z); L[t] = mfma_i8(a, bl, z);                                       \
                } else { H[t] = mfma_i8(a, bh, H[t]); L[t] = mfma_i8(a, bl, L[t]); }                          \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);   /* one K-step's operands live at a time (register pressure) */ \
        }
        KSTEP(0) KSTEP(1) KSTEP(2) KSTEP(3) KSTEP(4) KSTEP(5) KSTEP(6) KSTEP(7)
#undef KSTEP
        // ---- block-sum term: 16 per-16 sums x (mins duplicated | scales); K = 16 of 32 used (lanes kg == 0)
        i32x4 bm = {0, 0, 0, 0};
        if (kg == 0) {
            if constexpr (TYPE == T_Q6_K) { bm.x = (int)R.scv.x; bm.y = (int)R.scv.y; bm.z = (int)R.scv.z; bm.w = (int)R.scv.w; }
            else {
                bm.x = (int)perm(0, mn_lo, 0x01010000u); bm.y = (int)perm(0, mn_lo, 0x03030202u);
                bm.z = (int)perm(0, mn_hi, 0x01010000u); bm.w = (int)perm(0, mn_hi, 0x03030202u);
            }
        }
        const float dd = R.d();
        float dm = 0.0f;
        if constexpr (TYPE != T_Q6_K) dm = R.dmin();
        // raw weights of the next super-block (R is dead from here on)
        if (more) R.load(rowp, nb, sb + 1, kg);
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int tl = (tw * MT + t) * 32;
            i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0};
            if (kg == 0) {
                ah = *reinterpret_cast<const i32x4 *>(s_bh + (tl + n) * 16);
                al = *reinterpret_cast<const i32x4 *>(s_bl + (tl + n) * 16);
            }
            i32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0;
            __builtin_amdgcn_sched_barrier(0);
            const i32x16 mh = mfma_i8(ah, bm, z), ml = mfma_i8(al, bm, z);
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
                const f32x4 yd4 = *reinterpret_cast<const f32x4 *>(s_yd + tl + 8 * rq + 4 * kg);   // tokens m = 8 rq + 4 kg + (0..3)
#pragma unroll
                for (int ri = 0; ri < 4; ri++) {
                    const int r = rq * 4 + ri;
                    const float yd = yd4[ri];
                    const int bsum = 64 * mh[r] + ml[r];
                    if constexpr (TYPE == T_Q6_K) {
                        const int isum = 64 * H[t][r] + L[t][r] - 32 * bsum;
                        facc[t][r] += (dd * yd) * (float)isum;
                    } else {
                        const int isum = 32 * H[t][r] + L[t][r];
                        facc[t][r] += (dd * yd) * (float)isum - (dm * yd) * (float)bsum;
                    }
                }
            }
        }
        // ---- swap in the next tile
        __syncthreads();                             // every wave is done reading this tile
        if (more) {
#pragma unroll
            for (int i = 0; i < G::NST; i++) if (tid + NTHREADS * i < G::TOK_TILE * 16) *reinterpret_cast<u32x4 *>(s_aq + st_l[i]) = nxt[i];
            if (tid < G::TOK_TILE) {
                s_yd[tid] = nyd;
                *reinterpret_cast<u32x4 *>(s_bh + tid * 16) = nbh;
                *reinterpret_cast<u32x4 *>(s_bl + tid * 16) = nbl;
            }
        }
        __syncthreads();
    }
    // store: lane = weight row n, regs = tokens; 32 consecutive rows per token -> 128-B coalesced
    if (row_ok) {
#pragma unroll
        for (int t = 0; t < MT; t++) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt >= T) gt = T - 1;
                rv[r] = resid ? resid[(size_t)gt * ld_out + row0 + n] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt < T) out[(size_t)gt * ld_out + row0 + n] = rv[r] + facc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Small token counts (continuous-batching decode steps, 8 <= T <= 64): the contraction is HBM-bound again, so the
// weights are read in their GGUF form (0.56 B / weight, expanded in registers — at 32 tokens that VALU work runs at about
// the HBM rate) and the parallelism comes from K instead of tokens: workgroup = 32 weight rows x (32 * MT tokens), its 8
// waves take the super-blocks round-robin, each wave stages its OWN activation tile (no workgroup barrier in the loop),
// and the eight partial f32 tiles are summed through LDS in fixed wave order (deterministic).
// One launch covers up to three tensors that share the activation (Q / K / V), or the gate / up pair with the SwiGLU
// applied in the epilogue (both rows of a pair are contracted against the same staged tile).
struct KSplitArgs {
    MMQSeg seg[3];
    int n_seg, K, T, swiglu;
    const int8_t *aq; const float *ad; const int8_t *abh, *abl;
};

// one super-block of one 32-row tile against the wave's staged activation tile: facc += fold(H, L, block sums)
template <int TYPE, int MT>
__device__ __forceinline__ void ksplit_superblock(const RowSB<TYPE> &R, const int8_t *s_aq, const int8_t *s_bh, const int8_t *s_bl, const float *s_yd,
                                                  int n, int kg, float (&facc)[MT][16]) {
    i32x16 H[MT], L[MT];
    const int8_t *abase = s_aq + n * A_STRIDE + 16 * kg;
    uint32_t sc_lo = 0, sc_hi = 0, mn_lo = 0, mn_hi = 0;
    if constexpr (TYPE != T_Q6_K) R.scales(sc_lo, sc_hi, mn_lo, mn_hi);
#define KSTEP(J)                                                                                              \
    {                                                                                                         \
        i32x4 bh, bl;                                                                                         \
        if constexpr (TYPE == T_Q6_K) R.template bop_signed<J>(R.scale_pk(8 * (J >> 2) + 2 * (J & 3) + kg), bh, bl); \
        else R.template bop<J>(((J < 4 ? sc_lo : sc_hi) >> (8 * (J & 3))) & 0xff, bh, bl);                    \
        _Pragma("unroll") for (int t = 0; t < MT; t++) {                                                      \
            const i32x4 a = *reinterpret_cast<const i32x4 *>(abase + t * 32 * A_STRIDE + 32 * J);              \
            if (J == 0) {                                                                                     \
                i32x16 z;                                                                                     \
                _Pragma("unroll") for (int r = 0; r < 16; r++) z[r] = 0;                                      \
                H[t] = mfma_i8(a, bh, z); L[t] = mfma_i8(a, bl, z);                                           \
            } else { H[t] = mfma_i8(a, bh, H[t]); L[t] = mfma_i8(a, bl, L[t]); }                              \
        }                                                                                                     \
    }
    KSTEP(0) KSTEP(1) KSTEP(2) KSTEP(3) KSTEP(4) KSTEP(5) KSTEP(6) KSTEP(7)
#undef KSTEP
    const float dd = R.d();
    float dm = 0.0f;
    i32x4 bm = {0, 0, 0, 0};
    if constexpr (TYPE != T_Q6_K) {
        dm = R.dmin();
        if (kg == 0) {
            bm.x = (int)perm(0, mn_lo, 0x01010000u); bm.y = (int)perm(0, mn_lo, 0x03030202u);
            bm.z = (int)perm(0, mn_hi, 0x01010000u); bm.w = (int)perm(0, mn_hi, 0x03030202u);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; t++) {
        i32x16 ms;
        if constexpr (TYPE != T_Q6_K) {
            i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0};
            if (kg == 0) {
                ah = *reinterpret_cast<const i32x4 *>(s_bh + (t * 32 + n) * 16);
                al = *reinterpret_cast<const i32x4 *>(s_bl + (t * 32 + n) * 16);
            }
            i32x16 z;
#pragma unroll
            for (int r = 0; r < 16; r++) z[r] = 0;
            ms = mfma_i8(ah, bm, z);
#pragma unroll
            for (int r = 0; r < 16; r++) ms[r] <<= 6;
            ms = mfma_i8(al, bm, ms);
        }
#pragma unroll
        for (int rq = 0; rq < 4; rq++) {
            const f32x4 yd4 = *reinterpret_cast<const f32x4 *>(s_yd + t * 32 + 8 * rq + 4 * kg);
#pragma unroll
            for (int ri = 0; ri < 4; ri++) {
                const int r = rq * 4 + ri;
                const float yd = yd4[ri];
                if constexpr (TYPE == T_Q6_K) {
                    const int isum = 64 * H[t][r] + L[t][r];
                    facc[t][r] += (dd * yd) * (float)isum;
                } else {
                    const int isum = 32 * H[t][r] + L[t][r];
                    facc[t][r] += (dd * yd) * (float)isum - (dm * yd) * (float)ms[r];
                }
            }
        }
    }
}

// The same super-block for one token tile (T <= 32) in the GROUP-SCALE form: a K-step of the MFMA is exactly one 32-weight group of Q4_K /
// Q5_K, so the raw 4- / 5-bit codes go in as the B operand (two VALU instructions per 8 weights instead of ~10 for the two scaled planes),
// ONE MFMA per K-step yields sum_k q_k a_k of the group, and the 6-bit group scale is applied to the 16 results per lane with one
// v_mad_i32_i24 each.  With a single token tile the expansion per weight dominated the planes form (4 VALU cycles per MFMA cycle); this
// form trades the second MFMA chain for 16 multiply-adds.  The integer isum is the same number, so everything downstream is unchanged.
template <int TYPE>
__device__ __forceinline__ void ksplit_superblock_gs(const RowSB<TYPE> &R, const int8_t *s_aq, const int8_t *s_bh, const int8_t *s_bl, const float *s_yd,
                                                     int n, int kg, float (&facc)[1][16]) {
    const int8_t *abase = s_aq + n * A_STRIDE + 16 * kg;
    uint32_t sc_lo = 0, sc_hi = 0, mn_lo = 0, mn_hi = 0;
    R.scales(sc_lo, sc_hi, mn_lo, mn_hi);
    int isum[16];
#pragma unroll
    for (int r = 0; r < 16; r++) isum[r] = 0;
    i32x16 z;
#pragma unroll
    for (int r = 0; r < 16; r++) z[r] = 0;
#define GSTEP(J)                                                                                              \
    {                                                                                                         \
        const i32x16 I = mfma_i8(*reinterpret_cast<const i32x4 *>(abase + 32 * J), R.template codes<J>(), z); \
        const int sc = (int)(((J < 4 ? sc_lo : sc_hi) >> (8 * (J & 3))) & 0xff);                              \
        _Pragma("unroll") for (int r = 0; r < 16; r++) { isum[r] = __mul24(I[r], sc) + isum[r]; asm volatile("" : "+v"(isum[r])); } \
        __builtin_amdgcn_sched_barrier(0);   /* (LLVM reassociates the eight-term integer sums otherwise: all eight MFMA results live, 217 registers) */ \
    }
    GSTEP(0) GSTEP(1) GSTEP(2) GSTEP(3) GSTEP(4) GSTEP(5) GSTEP(6) GSTEP(7)
#undef GSTEP
    const float dd = R.d(), dm = R.dmin();
    i32x4 bm = {0, 0, 0, 0};
    if (kg == 0) {
        bm.x = (int)perm(0, mn_lo, 0x01010000u); bm.y = (int)perm(0, mn_lo, 0x03030202u);
        bm.z = (int)perm(0, mn_hi, 0x01010000u); bm.w = (int)perm(0, mn_hi, 0x03030202u);
    }
    i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0};
    if (kg == 0) {
        ah = *reinterpret_cast<const i32x4 *>(s_bh + n * 16);
        al = *reinterpret_cast<const i32x4 *>(s_bl + n * 16);
    }
    i32x16 ms = mfma_i8(ah, bm, z);
#pragma unroll
    for (int r = 0; r < 16; r++) ms[r] <<= 6;
    ms = mfma_i8(al, bm, ms);
#pragma unroll
    for (int rq = 0; rq < 4; rq++) {
        const f32x4 yd4 = *reinterpret_cast<const f32x4 *>(s_yd + 8 * rq + 4 * kg);
#pragma unroll
        for (int ri = 0; ri < 4; ri++) {
            const int r = rq * 4 + ri;
            const float yd = yd4[ri];
            facc[0][r] += (dd * yd) * (float)isum[r] - (dm * yd) * (float)ms[r];
        }
    }
}

template <int TYPE, int MT, bool SWIGLU>
__device__ __forceinline__ void ksplit_body(const KSplitArgs &a, const MMQSeg &sg, int row_tile, uint8_t *smem) {
    constexpr int TOK = 32 * MT;
    constexpr int W_A = TOK * A_STRIDE, W_P = TOK * 16;
    constexpr int W_BYTES = W_A + 2 * W_P + TOK * 4;            // per-wave LDS region
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int8_t *s_aq = reinterpret_cast<int8_t *>(smem + wave * W_BYTES);
    int8_t *s_bh = s_aq + W_A, *s_bl = s_bh + W_P;
    float *s_yd = reinterpret_cast<float *>(s_bl + W_P);
    const int K = a.K, T = a.T, nb = K >> 8;
    const int row0 = row_tile * 32, tok0 = blockIdx.y * TOK;
    const int n = lane & 31, kg = lane >> 5;
    int my_row = row0 + n;
    if (my_row >= sg.n_rows) my_row = sg.n_rows - 1;
    const uint8_t *rowp = sg.W + (size_t)my_row * sg.row_bytes;
    const uint8_t *rowp2 = SWIGLU ? a.seg[1].W + (size_t)my_row * a.seg[1].row_bytes : nullptr;

    constexpr int NST = TOK * 16 / 64;
    int ptok = tok0 + (lane < TOK ? lane : 0);
    if (ptok >= T) ptok = T - 1;
    const unsigned pl_g = (unsigned)ptok * (unsigned)nb;

    float facc[MT][16], facc2[MT][16];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) { facc[t][r] = 0.0f; if (SWIGLU) facc2[t][r] = 0.0f; }

    for (int sb = wave; sb < nb; sb += 8) {
        RowSB<TYPE> R;
        R.load(rowp, nb, sb, kg);
        {   // this wave's activation tile of super-block sb
            u32x4 tmp[NST];
#pragma unroll
            for (int i = 0; i < NST; i++) {
                const int p = lane + 64 * i;
                int gt = tok0 + (p >> 4);
                if (gt >= T) gt = T - 1;
                tmp[i] = ldg16(a.aq + (size_t)gt * K + (size_t)sb * 256 + (p & 15) * 16);
            }
            float yd = 0.0f;
            u32x4 vbh = {0, 0, 0, 0}, vbl = {0, 0, 0, 0};
            if (lane < TOK) {
                yd = a.ad[pl_g + (unsigned)sb];
                if (TYPE != T_Q6_K) {
                    vbh = ldg16(a.abh + ((size_t)pl_g + (unsigned)sb) * 16);
                    vbl = ldg16(a.abl + ((size_t)pl_g + (unsigned)sb) * 16);
                }
            }
            __builtin_amdgcn_wave_barrier();                    // the previous tile's reads are done (in-order LDS per wave)
#pragma unroll
            for (int i = 0; i < NST; i++) {
                const int p = lane + 64 * i;
                *reinterpret_cast<u32x4 *>(s_aq + (p >> 4) * A_STRIDE + (p & 15) * 16) = tmp[i];
            }
            if (lane < TOK) {
                s_yd[lane] = yd;
                if (TYPE != T_Q6_K) {
                    *reinterpret_cast<u32x4 *>(s_bh + lane * 16) = vbh;
                    *reinterpret_cast<u32x4 *>(s_bl + lane * 16) = vbl;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#ifndef MI355_KSPLIT_TWO_PLANES                                  // (tools: the two-plane form for one token tile too, for comparison)
        constexpr bool GS_FORM = MT == 1 && TYPE != T_Q6_K;   // (Q6_K: 16-weight groups, a K-step spans two)
#else
        constexpr bool GS_FORM = false;
#endif
        if constexpr (GS_FORM) ksplit_superblock_gs<TYPE>(R, s_aq, s_bh, s_bl, s_yd, n, kg, facc);
        else ksplit_superblock<TYPE, MT>(R, s_aq, s_bh, s_bl, s_yd, n, kg, facc);
        if constexpr (SWIGLU) {                                 // the up row of the pair against the same tile
            RowSB<TYPE> R2;
            R2.load(rowp2, nb, sb, kg);
            if constexpr (GS_FORM) ksplit_superblock_gs<TYPE>(R2, s_aq, s_bh, s_bl, s_yd, n, kg, facc2);
            else ksplit_superblock<TYPE, MT>(R2, s_aq, s_bh, s_bl, s_yd, n, kg, facc2);
        }
    }
    // ---- sum the eight waves' partial tiles in wave order (deterministic), then store
    constexpr int NP = SWIGLU ? 2 : 1;
    __syncthreads();                                   // all activation tiles consumed: the LDS is reused for the partials
    float *part = reinterpret_cast<float *>(smem);     // [wave][NP][MT * 16][64]
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            part[((wave * NP + 0) * MT * 16 + t * 16 + r) * 64 + lane] = facc[t][r];
            if (SWIGLU) part[((wave * NP + 1) * MT * 16 + t * 16 + r) * 64 + lane] = facc2[t][r];
        }
    __syncthreads();
    for (int e = tid; e < MT * 16 * 64; e += NTHREADS) {
        float sum = 0.0f, sum2 = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; w++) {
            sum += part[(w * NP + 0) * MT * 16 * 64 + e];
            if (SWIGLU) sum2 += part[(w * NP + 1) * MT * 16 * 64 + e];
        }
        const int ln = e & 63, tr = e >> 6;            // tr = t * 16 + r
        const int t = tr >> 4, r = tr & 15;
        const int m = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        const int gt = tok0 + t * 32 + m, row = row0 + (ln & 31);
        if (gt < T && row < sg.n_rows) {
            const size_t o = (size_t)gt * sg.ld_out + row;
            if (SWIGLU) sg.out[o] = (sum / (1.0f + expf(-sum))) * sum2;
            else sg.out[o] = sg.resid ? sg.resid[o] + sum : sum;
        }
    }
}

template <int MT, bool SWIGLU>
__global__ __launch_bounds__(NTHREADS) void mmq_ksplit_kernel(const KSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int s = 0;
    if (a.n_seg > 1 && (int)blockIdx.x >= a.seg[1].tile0) s = 1;
    if (a.n_seg > 2 && (int)blockIdx.x >= a.seg[2].tile0) s = 2;
    if (SWIGLU) s = 0;
    const int row_tile = (int)blockIdx.x - a.seg[s].tile0;
    switch (a.seg[s].type) {
        case T_Q4_K: ksplit_body<T_Q4_K, MT, SWIGLU>(a, a.seg[s], row_tile, smem); break;
        case T_Q5_K: ksplit_body<T_Q5_K, MT, SWIGLU>(a, a.seg[s], row_tile, smem); break;
        case T_Q6_K: ksplit_body<T_Q6_K, MT, SWIGLU>(a, a.seg[s], row_tile, smem); break;
        default: break;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Pre-expanded weights ("planes"): the two int8 planes of every K-step are computed ONCE at model load and laid out
// exactly as the MFMA B operands are consumed, so the prefill kernel does no per-weight VALU work at all:
//   block(rt, sb) = 32 rows x one super-block:  [K-step J 0..7][plane 0 = hi, 1 = lo][lane 0..63][16 B]  (16 KB)
//                                               [row n 0..31][16 B meta: d | dmin << 16 (f16 bits), mins 0-3, mins 4-7, 0]
// 2 bytes per weight + 1/16 B meta (Q4_K: 3.7x the GGUF bytes).  Q6_K planes carry the SIGNED code (q - 32), so that
// type needs no block-sum term at all.
constexpr int PL_BLOCK = 8 * 2 * 1024 + 32 * 16;

template <int TYPE>
__global__ __launch_bounds__(256) void mmq_expand_kernel(const uint8_t *W, size_t row_bytes, int n_rows, int nb, int n_rt, uint8_t *planes) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);     // (row tile, super-block)
    if (idx >= n_rt * nb) return;
    const int rt = idx / nb, sb = idx - rt * nb;
    const int n = lane & 31, kg = lane >> 5;
    int row = rt * 32 + n;
    if (row >= n_rows) row = n_rows - 1;
    RowSB<TYPE> R;
    R.load(W + (size_t)row * row_bytes, nb, sb, kg);
    uint8_t *blk = planes + (size_t)idx * PL_BLOCK;
    uint32_t sc_lo = 0, sc_hi = 0, mn_lo = 0, mn_hi = 0;
    if constexpr (TYPE != T_Q6_K) R.scales(sc_lo, sc_hi, mn_lo, mn_hi);
#define EXP(J)                                                                                                 \
    {                                                                                                          \
        i32x4 bh, bl;                                                                                          \
        if constexpr (TYPE == T_Q6_K) R.template bop_signed<J>(R.scale_pk(8 * (J >> 2) + 2 * (J & 3) + kg), bh, bl); \
        else R.template bop<J>(((J < 4 ? sc_lo : sc_hi) >> (8 * (J & 3))) & 0xff, bh, bl);                     \
        *reinterpret_cast<i32x4 *>(blk + (J * 2) * 1024 + lane * 16) = bh;                                     \
        *reinterpret_cast<i32x4 *>(blk + (J * 2 + 1) * 1024 + lane * 16) = bl;                                 \
    }
    EXP(0) EXP(1) EXP(2) EXP(3) EXP(4) EXP(5) EXP(6) EXP(7)
#undef EXP
    if (kg == 0) {
        u32x4 m = {0, 0, 0, 0};
        if constexpr (TYPE == T_Q6_K) m.x = R.d16;
        else { m.x = R.h.x; m.y = mn_lo; m.z = mn_hi; }
        *reinterpret_cast<u32x4 *>(blk + 16384 + n * 16) = m;
    }
}

// Q2_K and Q3_K into the same plane format (their prompts then run on the kernels below; there is no expand-on-the-fly form for them):
//   Q2_K: p = scale * code <= 15 * 3, planes as Q4_K's (hi = p >> 5, lo = p & 31); sub-block 2 J + kg (16 weights) is exactly a lane's share of K-step J;
//         its sixteen 4-bit mins go into the meta words as nibbles (y: sub-blocks 0-7, z: 8-15) and w = 1 tells the fold to read them that way
//         (one min per block sum instead of one per two);
//   Q3_K: p = (scale - 32) * code with code in [-4, 3]: signed, 16-weight groups, no mins - planes as Q6_K's (hi = p >> 6 floored, lo = p & 63).
// device rows (dev_common.h): Q2_K [qs nb*64][scales nb*16][d, dmin nb*4];  Q3_K [hmask nb*32][qs nb*64][scales nb*12][d nb*2]
template <int TYPE>
__global__ __launch_bounds__(256) void mmq_expand_small_kernel(const uint8_t *W, size_t row_bytes, int n_rows, int nb, int n_rt, uint8_t *planes) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);     // (row tile, super-block)
    if (idx >= n_rt * nb) return;
    const int rt = idx / nb, sb = idx - rt * nb;
    const int n = lane & 31, kg = lane >> 5;
    int row = rt * 32 + n;
    if (row >= n_rows) row = n_rows - 1;
    const uint8_t *r = W + (size_t)row * row_bytes;
    uint8_t *blk = planes + (size_t)idx * PL_BLOCK;
    const size_t qs_off = TYPE == T_Q2_K ? 0 : (size_t)nb * 32;
    // this lane's code bytes: l = 16 kg .. 16 kg + 15 of both 32-byte halves
    const u32x4 q0 = ldg16(r + qs_off + (size_t)sb * 64 + 16 * kg), q1 = ldg16(r + qs_off + (size_t)sb * 64 + 32 + 16 * kg);
    u32x4 hm = {0, 0, 0, 0};
    int sc[16];
    uint32_t mn_lo = 0, mn_hi = 0;
    if (TYPE == T_Q2_K) {
        const uint8_t *scp = r + (size_t)nb * 64 + (size_t)sb * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int b = scp[i];
            sc[i] = b & 15;
            if (i < 8) mn_lo |= (uint32_t)(b >> 4) << (4 * i); else mn_hi |= (uint32_t)(b >> 4) << (4 * (i - 8));
        }
    } else {
        hm = ldg16(r + (size_t)sb * 32 + 16 * kg);
        const uint8_t *scp = r + (size_t)nb * 96 + (size_t)sb * 12;
#pragma unroll
        for (int i = 0; i < 16; i++) {   // 6-bit scale i: low 4 bits = nibble i of bytes 0-7 (low nibbles 0-7, high nibbles 8-15), high 2 bits = bit pair i / 4 of byte 8 + i % 4
            const int low = i < 8 ? (scp[i] & 15) : (scp[i - 8] >> 4);
            sc[i] = (low | (((scp[8 + (i & 3)] >> (2 * (i >> 2))) & 3) << 4)) - 32;
        }
    }
#pragma unroll
    for (int J = 0; J < 8; J++) {
        const u32x4 qv = (J >> 2) ? q1 : q0;
        const int sh = 2 * (J & 3);
        const int s = sc[2 * J + kg];
        const uint32_t cw[4] = {(qv.x >> sh) & 0x03030303u, (qv.y >> sh) & 0x03030303u, (qv.z >> sh) & 0x03030303u, (qv.w >> sh) & 0x03030303u};
        const uint32_t hw[4] = {hm.x, hm.y, hm.z, hm.w};
        uint32_t hv[4], lv[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t hx = 0, lx = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int code = (int)((cw[w] >> (8 * b)) & 3u);
                int p;
                if (TYPE == T_Q2_K) { p = s * code; hx |= (uint32_t)((p >> 5) & 0xff) << (8 * b); lx |= (uint32_t)(p & 31) << (8 * b); }
                else {
                    if (!((hw[w] >> (8 * b + J)) & 1u)) code -= 4;          // bit J = 4 (J >> 2) + (J & 3) of the element's hmask byte
                    p = s * code;
                    hx |= (uint32_t)((p >> 6) & 0xff) << (8 * b); lx |= (uint32_t)(p & 63) << (8 * b);
                }
            }
            hv[w] = hx; lv[w] = lx;
        }
        *reinterpret_cast<u32x4 *>(blk + (J * 2) * 1024 + lane * 16) = u32x4{hv[0], hv[1], hv[2], hv[3]};
        *reinterpret_cast<u32x4 *>(blk + (J * 2 + 1) * 1024 + lane * 16) = u32x4{lv[0], lv[1], lv[2], lv[3]};
    }
    if (kg == 0) {
        u32x4 m = {0, 0, 0, 0};
        if (TYPE == T_Q2_K) { m.x = *reinterpret_cast<const uint32_t *>(r + (size_t)nb * 80 + (size_t)sb * 4); m.y = mn_lo; m.z = mn_hi; m.w = 1u; }
        else m.x = *reinterpret_cast<const uint16_t *>(r + (size_t)nb * 108 + (size_t)sb * 2);
        *reinterpret_cast<u32x4 *>(blk + 16384 + n * 16) = m;
    }
}
// which fold a plane set takes: the block-sum ("mins") form with shift 5, or the signed form with shift 6
__host__ __device__ constexpr bool planes_have_mins(int type) { return type == T_Q4_K || type == T_Q5_K || type == T_Q2_K; }

// Several tensors that share the activation and whose plane sets lie back to back in memory (attn_q | attn_k | attn_v) run as
// ONE launch over the concatenated rows: rows [row_end[s-1], row_end[s]) go to out[s] with leading dimension ld[s] (segment
// boundaries are multiples of 32, so a wave's 32-row tile belongs to one segment).  A 1024-row tensor alone fills a quarter of
// the chip and takes as long as a 4096-row one (34 vs 38 us at 512 tokens); the three together take 59 us instead of 106.
struct PlanesOut {
    float *out[3];
    int ld[3];
    int row_end[3];
    int n_seg;
    unsigned mins_mask;     // bit s: segment s is Q4_K / Q5_K (block-sum term, shift 5); mmq_planes2_kernel<2> only
};

// MINS: Q4_K / Q5_K (block-sum term, shift 5); otherwise Q6_K (shift 6, nothing else)
template <bool MINS, int MT>
__global__ __launch_bounds__(NTHREADS) void mmq_planes_kernel(const uint8_t *planes, int n_rows, int K, int T, int n_row_tiles, int n_tok_tiles,
                                                              const int8_t *aq, const float *ad, const int8_t *abh, const int8_t *abl,
                                                              const PlanesOut po, const float *resid) {
    using G = Geo<MT>;
    constexpr int SH = MINS ? 5 : 6;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // two activation tiles: super-block sb is read from buffer sb & 1 while sb + 1 is written into the other one, so one
    // barrier per super-block suffices and the LDS writes overlap the other waves' MFMA work
    uint8_t *buf0 = smem, *buf1 = smem + G::LDS_BYTES;
    auto p_aq = [&](int b) { return reinterpret_cast<int8_t *>(b ? buf1 : buf0); };
    auto p_bh = [&](int b) { return p_aq(b) + G::LDS_A; };
    auto p_bl = [&](int b) { return p_aq(b) + G::LDS_A + G::LDS_P; };
    auto p_yd = [&](int b) { return reinterpret_cast<float *>(p_aq(b) + G::LDS_A + 2 * G::LDS_P); };
    int8_t *s_aq = p_aq(0), *s_bh = p_bh(0), *s_bl = p_bl(0);
    float *s_yd = p_yd(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = K >> 8;
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tok_tile = loc % n_tok_tiles;
    const int row_tile = (loc / n_tok_tiles) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;                       // workgroup-uniform
    const int rw = wave % G::RW, tw = wave / G::RW;
    const int rt32 = row_tile * G::RW + rw;                    // 32-row tile of this wave
    const int n_rt32 = (n_rows + 31) >> 5;
    const int row0 = rt32 * 32;
    const int tok0 = tok_tile * G::TOK_TILE;
    const int n = lane & 31, kg = lane >> 5;
    const bool tile_ok = rt32 < n_rt32;
    const uint8_t *blk0 = planes + (size_t)(tile_ok ? rt32 : n_rt32 - 1) * nb * PL_BLOCK;
    const unsigned b_off = (unsigned)lane * 16u;

    unsigned st_g[G::NST], st_l[G::NST];
#pragma unroll
    for (int i = 0; i < G::NST; i++) {
        const int p = tid + NTHREADS * i;
        int tk = p >> 4;
        if (tk >= G::TOK_TILE) tk = G::TOK_TILE - 1;
        int gt = tok0 + tk;
        if (gt >= T) gt = T - 1;
        st_g[i] = (unsigned)gt * (unsigned)K + (unsigned)(p & 15) * 16u;
        st_l[i] = (unsigned)tk * A_STRIDE + (unsigned)(p & 15) * 16u;
    }
    int ptok = tok0 + (tid < G::TOK_TILE ? tid : 0);
    if (ptok >= T) ptok = T - 1;
    const unsigned pl_g = (unsigned)ptok * (unsigned)nb;

    float facc[MT][16];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) facc[t][r] = 0.0f;

    // ---- prologue: B operands and meta of super-block 0 into registers, activation tile 0 into LDS
    // ring of K-steps: 4 (half a super-block ahead) where two token tiles per wave leave no registers for more; 8 (a whole
    // super-block ahead) with one tile per wave, whose 2 MFMAs per K-step otherwise cover only 256 cycles of the planes' latency
    constexpr int RING = MT == 1 ? 8 : 4;
    i32x4 Bh[RING], Bl[RING];
#pragma unroll
    for (int j = 0; j < RING; j++) {
        Bh[j] = *reinterpret_cast<const i32x4 *>(blk0 + (j * 2) * 1024 + b_off);
        Bl[j] = *reinterpret_cast<const i32x4 *>(blk0 + (j * 2 + 1) * 1024 + b_off);
    }
    u32x4 meta = ldg16(blk0 + 16384 + n * 16);
    {
        u32x4 tmp[G::NST];
#pragma unroll
        for (int i = 0; i < G::NST; i++) tmp[i] = ldg16(aq + st_g[i]);
#pragma unroll
        for (int i = 0; i < G::NST; i++) if (tid + NTHREADS * i < G::TOK_TILE * 16) *reinterpret_cast<u32x4 *>(s_aq + st_l[i]) = tmp[i];
        if (tid < G::TOK_TILE) {
            s_yd[tid] = ad[pl_g];
            if (MINS) {
                *reinterpret_cast<u32x4 *>(s_bh + tid * 16) = ldg16(abh + (size_t)pl_g * 16);
                *reinterpret_cast<u32x4 *>(s_bl + tid * 16) = ldg16(abl + (size_t)pl_g * 16);
            }
        }
    }
    __syncthreads();

    for (int sb = 0; sb < nb; sb++) {
        s_aq = p_aq(sb & 1); s_bh = p_bh(sb & 1); s_bl = p_bl(sb & 1); s_yd = p_yd(sb & 1);
        const bool more = sb + 1 < nb;
        const int sbn = more ? sb + 1 : sb;
        const uint8_t *blkn = blk0 + (size_t)sbn * PL_BLOCK;
        // next activation tile: request now, park in registers across the MFMA work
        u32x4 nxt[G::NST];
#pragma unroll
        for (int i = 0; i < G::NST; i++) nxt[i] = ldg16(aq + st_g[i] + (unsigned)sbn * 256u);
        float nyd = 0.0f;
        u32x4 nbh = {0, 0, 0, 0}, nbl = {0, 0, 0, 0};
        if (tid < G::TOK_TILE) {
            nyd = ad[pl_g + (unsigned)sbn];
            if (MINS) {
                nbh = ldg16(abh + ((size_t)pl_g + (unsigned)sbn) * 16);
                nbl = ldg16(abl + ((size_t)pl_g + (unsigned)sbn) * 16);
            }
        }
        const u32x4 mcur = meta;
        meta = ldg16(blkn + 16384 + n * 16);

        i32x16 H[MT], L[MT];
        const int8_t *abase = s_aq + (tw * MT * 32 + n) * A_STRIDE + 16 * kg;
        const uint8_t *blkc = blk0 + (size_t)sb * PL_BLOCK;
        // K-step J: MFMAs on ring slot J & 3, then the slot is refilled with the operand four K-steps ahead (same
        // super-block for J < 4, the next one otherwise): every B register waits half an iteration before its use
        // A fragments are read one K-step ahead of the MFMAs that use them (two register sets)
        i32x4 a0[MT], a1[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) a0[t] = *reinterpret_cast<const i32x4 *>(abase + t * 32 * A_STRIDE);
#define KSTEP(J, ACUR, ANXT)                                                                                  \
        {                                                                                                     \
            if (J < 7) {                                                                                      \
                _Pragma("unroll") for (int t = 0; t < MT; t++)                                                \
                    ANXT[t] = *reinterpret_cast<const i32x4 *>(abase + t * 32 * A_STRIDE + 32 * (J + 1));     \
            }                                                                                                 \
            _Pragma("unroll") for (int t = 0; t < MT; t++) {                                                  \
                if (J == 0) {                                                                                 \
                    i32x16 z;                                                                                 \
                    _Pragma("unroll") for (int r = 0; r < 16; r++) z[r] = 0;                                  \
                    H[t] = mfma_i8(ACUR[t], Bh[J & (RING - 1)], z); L[t] = mfma_i8(ACUR[t], Bl[J & (RING - 1)], z); \
                } else { H[t] = mfma_i8(ACUR[t], Bh[J & (RING - 1)], H[t]); L[t] = mfma_i8(ACUR[t], Bl[J & (RING - 1)], L[t]); } \
            }                                                                                                 \
            const uint8_t *src = RING == 8 ? blkn + (J * 2) * 1024 + b_off                                    \
                                           : (J < 4 ? blkc : blkn) + (((J + 4) & 7) * 2) * 1024 + b_off;       \
            Bh[J & (RING - 1)] = *reinterpret_cast<const i32x4 *>(src);                                       \
            Bl[J & (RING - 1)] = *reinterpret_cast<const i32x4 *>(src + 1024);                                \
        }
        KSTEP(0, a0, a1) KSTEP(1, a1, a0) KSTEP(2, a0, a1) KSTEP(3, a1, a0) KSTEP(4, a0, a1) KSTEP(5, a1, a0) KSTEP(6, a0, a1) KSTEP(7, a1, a0)
#undef KSTEP
        const float dd = h2f((uint16_t)(mcur.x & 0xffff));
        float dm = 0.0f;
        i32x4 bm = {0, 0, 0, 0};
        if (MINS) {
            dm = h2f((uint16_t)(mcur.x >> 16));
            if (kg == 0) {
                if (mcur.w) {           // sixteen 4-bit mins, one per block sum (Q2_K planes): nibbles -> bytes in block-sum order
                    const uint32_t ey = mcur.y & 0x0f0f0f0fu, oy = (mcur.y >> 4) & 0x0f0f0f0fu, ez = mcur.z & 0x0f0f0f0fu, oz = (mcur.z >> 4) & 0x0f0f0f0fu;
                    bm.x = (int)perm(oy, ey, 0x05010400u); bm.y = (int)perm(oy, ey, 0x07030602u);
                    bm.z = (int)perm(oz, ez, 0x05010400u); bm.w = (int)perm(oz, ez, 0x07030602u);
                } else {
                    bm.x = (int)perm(0, mcur.y, 0x01010000u); bm.y = (int)perm(0, mcur.y, 0x03030202u);
                    bm.z = (int)perm(0, mcur.z, 0x01010000u); bm.w = (int)perm(0, mcur.z, 0x03030202u);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int tl = (tw * MT + t) * 32;
            i32x16 ms;                                // block-sum term: 64 * (bh . m) + (bl . m) in ONE accumulator
            if (MINS) {
                i32x4 ah = {0, 0, 0, 0}, al = {0, 0, 0, 0};
                if (kg == 0) {
                    ah = *reinterpret_cast<const i32x4 *>(s_bh + (tl + n) * 16);
                    al = *reinterpret_cast<const i32x4 *>(s_bl + (tl + n) * 16);
                }
                i32x16 z;
#pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0;
                ms = mfma_i8(ah, bm, z);
#pragma unroll
                for (int r = 0; r < 16; r++) ms[r] <<= 6;
                ms = mfma_i8(al, bm, ms);
            }
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
                const f32x4 yd4 = *reinterpret_cast<const f32x4 *>(s_yd + tl + 8 * rq + 4 * kg);
#pragma unroll
                for (int ri = 0; ri < 4; ri++) {
                    const int r = rq * 4 + ri;
                    const float yd = yd4[ri];
                    const int isum = (H[t][r] << SH) + L[t][r];
                    // yd * (d * isum - dmin * msum) with explicit fmas: 6 VALU ops per output instead of 10 (this fold was the
                    // kernel's VALU bottleneck: PMC showed 9.5 VALU instructions per MFMA); one rounding fewer than the CPU's
                    // expression, far inside the 2e-5 parity bound
                    if (MINS) {
                        const float t0 = fmaf(dd, (float)isum, -(dm * (float)ms[r]));
                        facc[t][r] = fmaf(yd, t0, facc[t][r]);
                    } else {
                        facc[t][r] = fmaf(yd * dd, (float)isum, facc[t][r]);
                    }
                }
            }
        }
        if (more) {                                  // next tile into the other buffer (nobody reads it during this iteration)
            int8_t *w_aq = p_aq((sb + 1) & 1), *w_bh = p_bh((sb + 1) & 1), *w_bl = p_bl((sb + 1) & 1);
            float *w_yd = p_yd((sb + 1) & 1);
#pragma unroll
            for (int i = 0; i < G::NST; i++) if (tid + NTHREADS * i < G::TOK_TILE * 16) *reinterpret_cast<u32x4 *>(w_aq + st_l[i]) = nxt[i];
            if (tid < G::TOK_TILE) {
                w_yd[tid] = nyd;
                if (MINS) {
                    *reinterpret_cast<u32x4 *>(w_bh + tid * 16) = nbh;
                    *reinterpret_cast<u32x4 *>(w_bl + tid * 16) = nbl;
                }
            }
        }
        __syncthreads();                             // tile sb + 1 complete; every wave is done reading tile sb
    }
    // store: lane = weight row n, regs = tokens; 32 consecutive rows per token -> 128-B coalesced.  Residual values are
    // all requested before the first add (a load + wait per element serialised the epilogue).
    if (tile_ok && row0 + n < n_rows) {
        int sg = 0;
        if (po.n_seg > 1 && row0 >= po.row_end[0]) sg = 1;
        if (po.n_seg > 2 && row0 >= po.row_end[1]) sg = 2;
        float *out = po.out[sg] - (sg ? po.row_end[sg - 1] : 0);       // so that out[t * ld + row] addresses the segment's own row
        const int ld_out = po.ld[sg];
#pragma unroll
        for (int t = 0; t < MT; t++) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt >= T) gt = T - 1;
                rv[r] = resid ? resid[(size_t)gt * ld_out + row0 + n] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt < T) out[(size_t)gt * ld_out + row0 + n] = rv[r] + facc[t][r];
            }
        }
    }
}

// ---- the same contraction with BOTH operands through LDS (prompts of a few hundred tokens and more on the wide tensors).
// mmq_planes_kernel reads its B operand (the weight planes) per lane from L2: 2 KiB per K-step per wave for MT x 2 MFMAs, so
// at MT = 2 the eight waves of a CU ask the vector L1 for 64 B / clk at full matrix-core rate - the L1's whole bandwidth - and
// the matrix cores idle most of the time (16 % busy on gate/up of the 8B model).  Here a workgroup covers 128 rows x 256 tokens:
// a wave owns 32 rows x 128 tokens (four token tiles: 8 MFMAs per pair of B registers), and each plane byte is fetched ONCE per
// workgroup - by the LDS-DMA (global_load_lds_dwordx4, no register round trip), half a super-block (4 K-steps) per stage into one
// of two 64 KiB buffers, the other one being read by the MFMAs:
//   B  32 KiB = [row tile 0..3][K-step 0..3][plane hi, lo][lane][16 B]   - the layout mmq_expand_kernel writes, copied verbatim
//   A  32 KiB = [token 0..255][8 pieces of 16 B]; piece p of token t sits in slot p ^ ((t >> 1) & 7) (the swizzle is applied on
//               the SOURCE side of the DMA, whose LDS side is always lane-linear), so the 16 lanes of a ds_read_b128 phase hit 16
//               different 16-B bank groups
// Per stage and CU: 64 KiB over the L1 (32 B / clk at full rate), 192 KiB out of LDS (96 B / clk of 128), 256 MFMAs.
// Per-super-block activation scales and block-sum planes are staged by ordinary loads one super-block ahead (two parities).
// The integer sums, the fold and the f32 order over super-blocks are those of mmq_planes_kernel: results are bit-identical.
__device__ __forceinline__ unsigned lds_addr32(const void *p) { return (unsigned)(uintptr_t)p; }
// lane l: 16 B from gbase + voff (voff per lane) to LDS dst + 16 l
__device__ __forceinline__ void dma16_s(const void *gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}
// lane l: 4 x 16 B from g, g + 1 KiB, g + 2 KiB, g + 3 KiB (g per lane) to dst + {0, 1, 2, 3} KiB + 16 l: the instruction
// offset applies to the global and to the LDS address alike
__device__ __forceinline__ void dma_4k(const void *g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

// lane l: 16 B from g + OFF (g per lane) to LDS dst + OFF + 16 l
template <int OFF> __device__ __forceinline__ void dma16_o(const void *g, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%c3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst), "i"(OFF) : "memory");
}

constexpr int P2_ROWS = 128, P2_TOK = 256, P2_MT = 4;
constexpr int P2_BUF = 65536, P2_A = 32768;                         // per buffer: B planes | A codes
constexpr int P2_YD = 0, P2_BS = 1024, P2_META = 1024 + 8192;
constexpr int P2_SBARR = P2_META + 2048;                            // per parity: yd[256] | bsums[256][16] (int16, then f16) | meta[4][32][16]
constexpr int P2_LDS = 2 * P2_BUF + 2 * P2_SBARR;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// lane l: 4 B from gbase + voff to LDS dst + 4 l
__device__ __forceinline__ void dma4_s(const void *gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(lds_dst) : "memory");
}

#ifdef MI355_P2_PROBE
// tools/exp_p2.hip: per wave of workgroup 0, shader cycles spent per stage kind {first, second half} in {compute issue, DMA wait, barrier}
__device__ unsigned long long *g_p2_probe = nullptr;
#define P2_STAMP() __builtin_readcyclecounter()
#endif
#ifndef MI355_P2_PRIO
#define MI355_P2_PRIO 0      // tools: s_setprio level of a wave inside its MFMA phase (0 = no instruction)
#endif
#ifndef MI355_P2_LATE_DMA
#define MI355_P2_LATE_DMA 0  // tools: 1 = a stage's DMA issued behind the issuing half's MFMAs instead of in front of them
#endif
#ifndef MI355_P2_EXP
#define MI355_P2_EXP 0       // tools: 1 = no fold, 2 = no DMA, 3 = no MFMA (timing experiments; results are garbage)
#endif
// Schedule.  A stage is half a super-block (4 K-steps); stage s is computed from buffer s & 1 while the DMA fills the other one;
// one workgroup barrier per stage.  A SIMD holds two waves of the workgroup, one of each token half, and the f32 fold of a
// super-block (VALU work, 40 % of the MFMA time) would leave the matrix cores idle if both did it at the same point.  So the
// token halves fold at DIFFERENT points: waves 0-3 right after the super-block's last MFMAs (end of its second stage), waves 4-7
// at the start of the next super-block's first stage - each half's fold runs beside the other half's MFMAs.
// The block-sum ("mins") term is one v_mfma_f32_32x32x16_f16 per tile: the 16 block sums (|.| <= 2032) and the 6-bit mins are
// exact in f16 and their 16 products sum to < 2^24, so the f32 result IS the integer msum - no shift, no conversion.
// SWIGLU: `planes` = ffn_gate, `planes_up` = ffn_up (n_rows rows each, same type): a workgroup takes 64 rows of BOTH (row-waves 0, 1 the
// gate rows, 2, 3 the same rows of up), the up waves hand their tile over through LDS at the end and the gate waves store
// silu(gate . x) * (up . x) - one [T][n_rows] f32 result instead of two, and no separate SwiGLU pass over them
template <bool MINS, bool SWIGLU>
__device__ __forceinline__ void planes2_body(const uint8_t *planes, const uint8_t *planes_up, int n_rows, int K, int T, int n_row_tiles, int n_tok_tiles,
                                             const int8_t *aq, const float *ad, const int16_t *absum,
                                             const PlanesOut &po, const float *resid, float *ws, int n_split, int row_tile, int tok_tile, int zz) {
    constexpr int MT = P2_MT;
    constexpr int SH = MINS ? 5 : 6;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb_all = K >> 8;                                 // super-blocks per row (strides); this workgroup's share is [sb_lo, sb_lo + nb)
    const int sb_lo = (int)((long)nb_all * zz / n_split), nb = (int)((long)nb_all * (zz + 1) / n_split) - sb_lo;
    const int rw = wave & 3, tw = wave >> 2;
    const int n_rt32 = (n_rows + 31) >> 5;
    // 32-row tile of row-wave index p (compute: p = rw; plane DMA: p = wave >> 1; row-word DMA: p = 2 (wave & 1) + kg) and its tensor
    auto tile_of = [&](int p) { return SWIGLU ? row_tile * 2 + (p & 1) : row_tile * 4 + p; };
    auto base_of = [&](int p) { return SWIGLU && (p >> 1) ? planes_up : planes; };
    const int rt32 = tile_of(rw);
    const bool tile_ok = rt32 < n_rt32;
    const int row0 = rt32 * 32;
    const int tok0 = tok_tile * P2_TOK;
    const int n = lane & 31, kg = lane >> 5;
    // a token half (128 tokens) past the end of the batch - 320 tokens leave the second workgroup tile 64 of 256 - keeps the DMA and
    // barrier protocol and skips its MFMAs and folds: the other half then has the matrix cores to itself
    const bool idle_half = tok0 + tw * 128 >= T;                // (wave-uniform)

    // ---- the DMA, in shares of a "virtual wave" vw = 0..7.  Per stage: 4 KiB of B (row tile vw / 2, K-steps 2 (vw & 1) .. + 1, both
    // planes) and 4 KiB of A (32 tokens x 128 B, eight tokens per instruction).  Per super-block: 1 KiB of block sums (32 tokens x
    // 32 B), activation scales (vw 0-3, 64 tokens each) and the row tiles' d / dmin / mins words (vw 4, 5: two row tiles each).
    // Addresses are formed when a share is issued (a dozen VALU instructions) instead of living in eight registers, and WHO issues
    // follows the schedule below: an LDS-DMA instruction costs its wave ~100 cycles of issue in a busy phase, so a stage's 64 + 14
    // instructions go to the token half that is NOT folding in that stage (it waits at the barrier for the folding half anyway), or to
    // the half that has no tokens at all.
    const unsigned lds0 = lds_addr32(smem);
    auto issue_v = [&](int s, int vw) {                         // stage s = (super-block s >> 1, half s & 1) into buffer s & 1
        if (MI355_P2_EXP == 2) return;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));                          // (formed here, every time: hoisted out of the loop the offsets would cost 20 registers)
        const unsigned buf = lds0 + (unsigned)(s & 1) * P2_BUF;
        int drt = tile_of(vw >> 1);
        if (drt >= n_rt32) drt = n_rt32 - 1;
        const uint8_t *gb = base_of(vw >> 1) + ((size_t)drt * nb_all + sb_lo + (s >> 1)) * PL_BLOCK + (vw & 1) * 4096 + (s & 1) * 8192;
        dma_4k(gb + lane * 16, buf + (unsigned)vw * 4096u);
        const int8_t *ga = aq + (size_t)s * 128;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int tk = 8 * (4 * vw + k) + (lane >> 3);
            int gt = tok0 + tk;
            if (gt >= T) gt = T - 1;
            const int piece = (lane & 7) ^ ((tk >> 1) & 7);
            dma16_s(ga, (unsigned)gt * (unsigned)K + (unsigned)sb_lo * 256u + (unsigned)piece * 16u, buf + P2_A + (unsigned)(4 * vw + k) * 1024u);
        }
    };
    auto issue_sb_v = [&](int sb, int vw) {                     // per-super-block words into parity sb & 1
        if (MI355_P2_EXP == 2) return;
        int lane = tid & 63;
        asm volatile("" : "+v"(lane));
        const int n = lane & 31, kg = lane >> 5;
        const unsigned par = lds0 + 2 * P2_BUF + (unsigned)(sb & 1) * P2_SBARR;
        if (MINS) {
            int btok = tok0 + 32 * vw + (lane >> 1);
            if (btok >= T) btok = T - 1;
            dma16_s(absum + (size_t)sb * 16, ((unsigned)btok * (unsigned)nb_all + (unsigned)sb_lo) * 32u + (unsigned)(lane & 1) * 16u, par + P2_BS + (unsigned)vw * 1024u);
        }
        if (vw < 4) {
            int ptok = tok0 + 64 * vw + lane;
            if (ptok >= T) ptok = T - 1;
            dma4_s(ad + sb, ((unsigned)ptok * (unsigned)nb_all + (unsigned)sb_lo) * 4u, par + P2_YD + (unsigned)vw * 256u);
        } else if (vw < 6) {
            const int p0 = 2 * (vw & 1);                         // first tile of this share's pair (SwiGLU: the pair is one tensor's)
            int mb0 = tile_of(p0), mrt = tile_of(p0 + kg);
            if (mb0 >= n_rt32) mb0 = n_rt32 - 1;
            if (mrt >= n_rt32) mrt = n_rt32 - 1;
            const uint8_t *mb = base_of(p0) + ((size_t)mb0 * nb_all + sb_lo + sb) * PL_BLOCK + 16384;
            dma16_s(mb, (unsigned)(mrt - mb0) * (unsigned)nb_all * (unsigned)PL_BLOCK + (unsigned)n * 16u, par + P2_META + (unsigned)(vw - 4) * 1024u);
        }
    };
    // the issuing half of a stage takes two shares per wave
    auto issue = [&](int s, int issuer_tw) { if (tw == issuer_tw) { issue_v(s, rw); issue_v(s, rw + 4); } };
    auto issue_sb = [&](int sb, int issuer_tw) { if (tw == issuer_tw) { issue_sb_v(sb, rw); issue_sb_v(sb, rw + 4); } };
    auto convert_bs = [&](int sb) {                             // int16 block sums -> f16, in place (16 B per thread)
        if (!MINS) return;
        uint8_t *q = smem + 2 * P2_BUF + (sb & 1) * P2_SBARR + P2_BS + tid * 16;
        const i16x8 v = *reinterpret_cast<const i16x8 *>(q);
        f16x8 f;
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = (_Float16)v[e];
        *reinterpret_cast<f16x8 *>(q) = f;
    };

    float facc[MT][16];
#pragma unroll
    for (int t = 0; t < MT; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) facc[t][r] = 0.0f;

    // swizzled byte offset of this lane's A fragment for K-step j of a stage (same for every token tile: tiles start at multiples of 32)
    const unsigned a_lane = (unsigned)(tw * 128 + n) * 128u;
    const unsigned a_key = (unsigned)((n >> 1) & 7);
    i32x16 H[MT], L[MT];

    // fold of the super-block whose words sit in parity par, token tile by token tile
    auto fold_all = [&](int par) {
        if (MI355_P2_EXP == 1) return;
        const uint8_t *pp = smem + 2 * P2_BUF + par * P2_SBARR;
        const float *s_yd = reinterpret_cast<const float *>(pp + P2_YD);
        const u32x4 mcur = *reinterpret_cast<const u32x4 *>(pp + P2_META + rw * 512 + n * 16);
        const float dd = h2f((uint16_t)(mcur.x & 0xffff));
        float ndm = 0.0f;
        f16x8 bm8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (MINS) {
            ndm = -h2f((uint16_t)(mcur.x >> 16));
            const unsigned w = kg ? mcur.z : mcur.y;            // mins 4 kg .. 4 kg + 3, each for two consecutive block sums
            if (mcur.w) {                                       // Q2_K planes: eight 4-bit mins 8 kg .. 8 kg + 7, one per block sum
#pragma unroll
                for (int e = 0; e < 8; e++) bm8[e] = (_Float16)(float)((w >> (4 * e)) & 0xfu);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const _Float16 v = (_Float16)(float)((w >> (8 * e)) & 0xffu);
                    bm8[2 * e] = v; bm8[2 * e + 1] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < MT; t++) {
            const int tl = (tw * MT + t) * 32;
            f32x16 ms;
            if (MINS) {
                const f16x8 ab = *reinterpret_cast<const f16x8 *>(pp + P2_BS + (tl + n) * 32 + kg * 16);
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; r++) z[r] = 0.0f;
                ms = __builtin_amdgcn_mfma_f32_32x32x16_f16(ab, bm8, z, 0, 0, 0);
            }
#pragma unroll
            for (int rq = 0; rq < 4; rq++) {
                const f32x4 yd4 = *reinterpret_cast<const f32x4 *>(s_yd + tl + 8 * rq + 4 * kg);
#ifndef MI355_P2_SCALAR_FOLD
                // the fold's f32 operations on pairs (v_pk_mul_f32 / v_pk_fma_f32: identical IEEE results, bit-identical to the scalar form below and to the
                // per-lane kernel; measured 1 % (Q4_K) to 3 % (Q6_K) per launch)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int rp = 0; rp < 2; rp++) {
                    const int r0 = rq * 4 + 2 * rp, r1 = r0 + 1;
                    const f32x2 cf = {(float)((H[t][r0] << SH) + L[t][r0]), (float)((H[t][r1] << SH) + L[t][r1])};
                    const f32x2 yd2 = {yd4[2 * rp], yd4[2 * rp + 1]}, dd2 = {dd, dd};
                    f32x2 acc = {facc[t][r0], facc[t][r1]};
                    if (MINS) {
                        const f32x2 ndm2 = {ndm, ndm}, ms2 = {ms[r0], ms[r1]};
                        const f32x2 t0 = __builtin_elementwise_fma(dd2, cf, ndm2 * ms2);
                        acc = __builtin_elementwise_fma(yd2, t0, acc);
                    } else {
                        acc = __builtin_elementwise_fma(yd2 * dd2, cf, acc);
                    }
                    facc[t][r0] = acc.x; facc[t][r1] = acc.y;
                }
#else
#pragma unroll
                for (int ri = 0; ri < 4; ri++) {
                    const int r = rq * 4 + ri;
                    const float yd = yd4[ri];
                    const int isum = (H[t][r] << SH) + L[t][r];
                    if (MINS) {
                        // dd * isum - dmin * msum:  (-dmin) * msum is -(dmin * msum) bit for bit
                        const float t0 = fmaf(dd, (float)isum, ndm * ms[r]);
                        facc[t][r] = fmaf(yd, t0, facc[t][r]);
                    } else {
                        facc[t][r] = fmaf(yd * dd, (float)isum, facc[t][r]);
                    }
                }
#endif
            }
#ifndef MI355_P2_NOFB
            __builtin_amdgcn_sched_barrier(0);                   // one tile's block-sum accumulator at a time (four at once spill)
#endif
        }
    };
#ifdef MI355_P2_PROBE
    unsigned long long pr_acc[2][3] = {{0, 0, 0}, {0, 0, 0}}, pr_t0 = P2_STAMP();
    int pr_kind = 0;
#endif
    auto stage_end = [&]() {
#ifdef MI355_P2_PROBE
        const unsigned long long t1 = P2_STAMP();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = P2_STAMP();
        __syncthreads();
        const unsigned long long t3 = P2_STAMP();
        pr_acc[pr_kind][0] += t1 - pr_t0; pr_acc[pr_kind][1] += t2 - t1; pr_acc[pr_kind][2] += t3 - t2;
        pr_t0 = t3;
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the next stage has landed (this wave's share; the barrier covers the rest)
        __syncthreads();
#endif
    };
    // the 4 K-steps of buffer h, K-step-major, as 8 groups of {two token tiles x two planes}; the operands of group g + 1 (and,
    // every other group, the planes of the next K-step) are requested before the MFMAs of group g are issued: a lone wave
    // otherwise waits out the LDS latency once per group with its matrix-core slot empty
    auto mma_stage = [&](int h, bool first) {
#if MI355_P2_PRIO
        __builtin_amdgcn_s_setprio(MI355_P2_PRIO);              // tools: the wave in its MFMA phase ahead of the SIMD's other wave (folding / issuing DMA)
#endif
        const uint8_t *bs = smem + h * P2_BUF + rw * 8192 + lane * 16;
        const uint8_t *as = smem + h * P2_BUF + P2_A + a_lane;
        auto ld_a = [&](int t, int j) {
            const unsigned slot = ((unsigned)(2 * j + kg) ^ a_key) * 16u;
            return *reinterpret_cast<const i32x4 *>(as + t * 32 * 128 + slot);
        };
        i32x4 bh = *reinterpret_cast<const i32x4 *>(bs), bl = *reinterpret_cast<const i32x4 *>(bs + 1024);
        i32x4 a0 = ld_a(0, 0), a1 = ld_a(1, 0);
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const int j = g >> 1, t0 = (g & 1) * 2;
            i32x4 n0 = a0, n1 = a1, nbh = bh, nbl = bl;
            if (g < 7) {
                n0 = ld_a((t0 + 2) & 3, (g + 1) >> 1); n1 = ld_a((t0 + 3) & 3, (g + 1) >> 1);
                if (g & 1) { nbh = *reinterpret_cast<const i32x4 *>(bs + (j + 1) * 2048); nbl = *reinterpret_cast<const i32x4 *>(bs + (j + 1) * 2048 + 1024); }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int t = t0 + u;
                const i32x4 a = u ? a1 : a0;
                if (MI355_P2_EXP == 3) {
                    if (first && j == 0) { for (int r = 0; r < 16; r++) { H[t][r] = 0; L[t][r] = 0; } }
                    H[t][j] += a.x + bh.y; L[t][j] += a.z + bl.w;
                } else if (first && j == 0) {
                    i32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; r++) z[r] = 0;
                    H[t] = mfma_i8(a, bh, z); L[t] = mfma_i8(a, bl, z);
                } else { H[t] = mfma_i8(a, bh, H[t]); L[t] = mfma_i8(a, bl, L[t]); }
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0; a1 = n1; bh = nbh; bl = nbl;
        }
#if MI355_P2_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    issue_v(0, wave);
    issue_sb_v(0, wave);
    stage_end();
    // first-half stages: waves 4-7 fold there, so waves 0-3 issue - unless waves 4-7 have no tokens and nothing else to do;
    // second-half stages: waves 0-3 fold, waves 4-7 issue
    const int issuer0 = (tok0 + 128 >= T) ? 1 : 0, issuer1 = 1;
    if (idle_half) {
        for (int sb = 0; sb < nb; sb++) {
            issue(2 * sb + 1, issuer0);
            convert_bs(sb);
            stage_end();
            if (sb + 1 < nb) { issue(2 * sb + 2, issuer1); issue_sb(sb + 1, issuer1); }
            stage_end();
        }
    } else
    for (int sb = 0; sb < nb; sb++) {
        // ---- first half (buffer 0).  The block sums of sb (landed with the barrier just passed) become f16 here, one barrier
        // ahead of their first use; waves 4-7 fold super-block sb - 1 before they overwrite its accumulators.
#ifdef MI355_P2_PROBE
        pr_kind = 0;
#endif
#if MI355_P2_LATE_DMA
        // tools: the issuing half runs its MFMAs FIRST and issues the next stage's DMA behind them (the folding half's MFMAs then find the pipe free);
        // the DMA has less of the stage left to land in
        convert_bs(sb);
        if (tw == 1 && sb > 0) fold_all((sb - 1) & 1);
        mma_stage(0, true);
        issue(2 * sb + 1, issuer0);
        stage_end();
#else
        issue(2 * sb + 1, issuer0);
        convert_bs(sb);
        if (tw == 1 && sb > 0) fold_all((sb - 1) & 1);
        mma_stage(0, true);
        stage_end();
#endif
        // ---- second half (buffer 1).  The words of super-block sb + 1 go to the other parity, which waves 4-7 finished
        // reading in the stage just ended; waves 0-3 fold sb at the end of this one.
#ifdef MI355_P2_PROBE
        pr_kind = 1;
#endif
#if MI355_P2_LATE_DMA
        mma_stage(1, false);
        if (sb + 1 < nb) { issue(2 * sb + 2, issuer1); issue_sb(sb + 1, issuer1); }
        if (tw == 0 || sb + 1 == nb) fold_all(sb & 1);
        stage_end();
#else
        if (sb + 1 < nb) { issue(2 * sb + 2, issuer1); issue_sb(sb + 1, issuer1); }
        mma_stage(1, false);
        if (tw == 0 || sb + 1 == nb) fold_all(sb & 1);
        stage_end();
#endif
    }
    if (MI355_P2_EXP == 1) {
#pragma unroll
        for (int t = 0; t < MT; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) facc[t][r] = (float)(H[t][r] + L[t][r]);
    }

#ifdef MI355_P2_PROBE
    if (g_p2_probe && blockIdx.x == 0 && lane == 0) {
        const unsigned long long t4 = P2_STAMP();
        unsigned long long *o = g_p2_probe + wave * 8;
        for (int k = 0; k < 2; k++) for (int c = 0; c < 3; c++) o[k * 3 + c] = pr_acc[k][c];
        o[6] = t4 - pr_t0;                                      // last fold
    }
#endif
    if (SWIGLU) {                                               // the up tile crosses to the gate wave of the same rows and tokens
        float *exch = reinterpret_cast<float *>(smem) + (size_t)((rw & 1) * 2 + tw) * (MT * 16 * 64) + lane;   // (both buffers are idle now)
        if (rw >= 2) {
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) exch[(t * 16 + r) * 64] = facc[t][r];
        }
        __syncthreads();
        if (rw < 2 && tile_ok && row0 + n < n_rows) {
            float *out = po.out[0];
            const int ld_out = po.ld[0];
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int gt = tok0 + (tw * MT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    const float gv = facc[t][r], uv = exch[(t * 16 + r) * 64];
                    if (gt < T) out[(size_t)gt * ld_out + row0 + n] = (gv / (1.0f + expf(-gv))) * uv;
                }
        }
        return;
    }
    if (n_split > 1) {                                          // partial sums of this K range: ws[zz][token][row], no residual
        if (tile_ok && row0 + n < n_rows) {
            float *o = ws + (size_t)zz * T * n_rows;
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int gt = tok0 + (tw * MT + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (gt < T) o[(size_t)gt * n_rows + row0 + n] = facc[t][r];
                }
        }
        return;
    }
    if (tile_ok && row0 + n < n_rows) {
        int sg = 0;
        if (po.n_seg > 1 && row0 >= po.row_end[0]) sg = 1;
        if (po.n_seg > 2 && row0 >= po.row_end[1]) sg = 2;
        float *out = po.out[sg] - (sg ? po.row_end[sg - 1] : 0);
        const int ld_out = po.ld[sg];
#pragma unroll
        for (int t = 0; t < MT; t++) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt >= T) gt = T - 1;
                rv[r] = resid ? resid[(size_t)gt * ld_out + row0 + n] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * kg;
                const int gt = tok0 + (tw * MT + t) * 32 + m;
                if (gt < T) out[(size_t)gt * ld_out + row0 + n] = rv[r] + facc[t][r];
            }
        }
    }
}

// MODE 0: Q6_K planes (signed codes, no block-sum term); 1: Q4_K / Q5_K; 2: per segment (po.mins_mask; segment boundaries at multiples of
// 128 rows, so a workgroup's row tile is of one kind) - Q | K | V of the layers whose attn_v is Q6_K, in one launch.  MODE 2 carries both
// bodies behind one workgroup-uniform branch (a run-time flag inside ONE body costs 92 bytes of scratch per lane and 2.3x the time).
template <int MODE>
__global__ __launch_bounds__(NTHREADS) void mmq_planes2_kernel(const uint8_t *planes, int n_rows, int K, int T, int n_row_tiles, int n_tok_tiles,
                                                               const int8_t *aq, const float *ad, const int16_t *absum,
                                                               const PlanesOut po, const float *resid, float *ws, int n_split) {
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tok_tile = loc % n_tok_tiles;
    const int zz = (loc / n_tok_tiles) % n_split;              // K split: partial sums go to ws[zz], mmq_splitk_reduce_kernel adds them up
    const int row_tile = (loc / (n_tok_tiles * n_split)) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;                       // workgroup-uniform
    bool mins = MODE == 1;
    if (MODE == 2) {
        int sg0 = 0;
        if (po.n_seg > 1 && row_tile * P2_ROWS >= po.row_end[0]) sg0 = 1;
        if (po.n_seg > 2 && row_tile * P2_ROWS >= po.row_end[1]) sg0 = 2;
        mins = (po.mins_mask >> sg0) & 1u;
    }
    if (MODE == 1 || (MODE == 2 && mins)) planes2_body<true, false>(planes, nullptr, n_rows, K, T, n_row_tiles, n_tok_tiles, aq, ad, absum, po, resid, ws, n_split, row_tile, tok_tile, zz);
    else planes2_body<false, false>(planes, nullptr, n_rows, K, T, n_row_tiles, n_tok_tiles, aq, ad, absum, po, resid, ws, n_split, row_tile, tok_tile, zz);
}

// ffn_gate and ffn_up with SwiGLU in the epilogue: n_rows rows of each, 64 per workgroup, out[t][row] = silu(gate) * up
template <bool MINS>
__global__ __launch_bounds__(NTHREADS) void mmq_planes2_swiglu_kernel(const uint8_t *planes_gate, const uint8_t *planes_up, int n_rows, int K, int T,
                                                                      int n_row_tiles, int n_tok_tiles, const int8_t *aq, const float *ad,
                                                                      const int16_t *absum, const PlanesOut po) {
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tok_tile = loc % n_tok_tiles;
    const int row_tile = (loc / n_tok_tiles) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;                       // workgroup-uniform
    planes2_body<MINS, true>(planes_gate, planes_up, n_rows, K, T, n_row_tiles, n_tok_tiles, aq, ad, absum, po, nullptr, nullptr, 1, row_tile, tok_tile, 0);
}

// ---- ggml_mul_mat_id on a prompt batch as ONE launch per projection (round 5).  The (token, rank) pairs of the batch are sorted by expert (moe_group_kernel: expert
// e owns rows [meta[NE + e], + meta[e]) of the grouped activation / result arrays); a workgroup is (row tile, token-tile slot j): slot j is the j-th 256-token tile
// of the concatenation of the experts' batches, found by walking the per-expert counts ON THE DEVICE - no host synchronisation to size per-expert launches, and
// eight half-empty launches of ~128 tokens become one that fills the chip.  The body is planes2_body unchanged: the wrapper hands it the expert's planes, the
// batch's rows as "the" token range (pointers offset to the batch's first row, T = the batch's length) and the tile index within the batch.
struct MoeTiles { const int32_t *meta; int n_expert; size_t plane_stride; };     // meta: [0, NE) tokens per expert, [NE, 2 NE) first grouped row
__device__ __forceinline__ bool moe_tile_of(const MoeTiles &m, int j, int &e_out, int &tile_out, int &r0_out, int &n_out) {
    for (int e = 0; e < m.n_expert; e++) {                      // (scalar: the counts are a few words of one cache line)
        const int n_e = m.meta[e];
        const int tiles = (n_e + P2_TOK - 1) / P2_TOK;
        if (j < tiles) { e_out = e; tile_out = j; r0_out = m.meta[m.n_expert + e]; n_out = n_e; return true; }
        j -= tiles;
    }
    return false;
}
template <bool MINS>
__global__ __launch_bounds__(NTHREADS) void mmq_planes2_swiglu_moe_kernel(const uint8_t *planes_gate, const uint8_t *planes_up, int n_rows, int K, int n_row_tiles,
                                                                          int n_tile_slots, const int8_t *aq, const float *ad, const int16_t *absum, PlanesOut po,
                                                                          const MoeTiles mt) {
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int slot = loc % n_tile_slots;
    const int row_tile = (loc / n_tile_slots) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;                       // workgroup-uniform
    int e, tile, r0, n_e;
    if (!moe_tile_of(mt, slot, e, tile, r0, n_e)) return;
    const size_t nb = (size_t)(K >> 8);
    po.out[0] += (size_t)r0 * po.ld[0];
    planes2_body<MINS, true>(planes_gate + (size_t)e * mt.plane_stride, planes_up + (size_t)e * mt.plane_stride, n_rows, K, n_e, n_row_tiles, (n_e + P2_TOK - 1) / P2_TOK,
                             aq + (size_t)r0 * K, ad + (size_t)r0 * nb, absum + (size_t)r0 * (K >> 4), po, nullptr, nullptr, 1, row_tile, tile, 0);
}
template <bool MINS>
__global__ __launch_bounds__(NTHREADS) void mmq_planes2_moe_kernel(const uint8_t *planes, int n_rows, int K, int n_row_tiles, int n_tile_slots,
                                                                   const int8_t *aq, const float *ad, const int16_t *absum, PlanesOut po, const MoeTiles mt) {
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int slot = loc % n_tile_slots;
    const int row_tile = (loc / n_tile_slots) * 8 + xcd;
    if (row_tile >= n_row_tiles) return;
    int e, tile, r0, n_e;
    if (!moe_tile_of(mt, slot, e, tile, r0, n_e)) return;
    const size_t nb = (size_t)(K >> 8);
    po.out[0] += (size_t)r0 * po.ld[0];
    planes2_body<MINS, false>(planes + (size_t)e * mt.plane_stride, nullptr, n_rows, K, n_e, n_row_tiles, (n_e + P2_TOK - 1) / P2_TOK,
                              aq + (size_t)r0 * K, ad + (size_t)r0 * nb, absum + (size_t)r0 * (K >> 4), po, nullptr, nullptr, 1, row_tile, tile, 0);
}

// out[t][row] = (resid) + ws[0][t][row] + ws[1][t][row] + ... in split order (fixed, so results do not depend on timing)
__global__ __launch_bounds__(256) void mmq_splitk_reduce_kernel(const float *ws, int n_split, int T, int n_rows, const PlanesOut po, const float *resid) {
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x, per_tok = (size_t)n_rows >> 2;      // n_rows % 4 == 0 (launcher)
    if (i4 >= per_tok * T) return;
    const int t = (int)(i4 / per_tok), row = (int)(i4 % per_tok) * 4;
    const size_t plane = (size_t)T * n_rows, at = (size_t)t * n_rows + row;
    f32x4 acc = *reinterpret_cast<const f32x4 *>(ws + at);
    for (int z = 1; z < n_split; z++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(ws + z * plane + at);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    int sg = 0;
    if (po.n_seg > 1 && row >= po.row_end[0]) sg = 1;
    if (po.n_seg > 2 && row >= po.row_end[1]) sg = 2;
    float *o = po.out[sg] + (size_t)t * po.ld[sg] + (row - (sg ? po.row_end[sg - 1] : 0));
    if (resid) {
        const float *rp = resid + (size_t)t * po.ld[0] + row;
        acc.x = rp[0] + acc.x; acc.y = rp[1] + acc.y; acc.z = rp[2] + acc.z; acc.w = rp[3] + acc.w;
    }
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
}

// per-16 block sums of Q8_K split into int8 planes: bsum = 64*hi + lo, hi in [-32, 31], lo in [0, 63]
__global__ void mmq_prep_kernel(const int16_t *bsums, size_t n, int8_t *bh, int8_t *bl) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over T * K/16
    if (i >= n) return;
    const int s = (int)bsums[i];
    const int hi = s >> 6;                    // arithmetic shift: floor(s / 64)
    bh[i] = (int8_t)hi;
    bl[i] = (int8_t)(s - 64 * hi);
}

template <int TYPE, int MT>
hipError_t launch_one(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                      float *out, int ld_out, const float *resid, hipStream_t st) {
    using G = Geo<MT>;
    const int nrt = (n_rows + G::ROW_TILE - 1) / G::ROW_TILE, ntt = (T + G::TOK_TILE - 1) / G::TOK_TILE;
    const int groups = (nrt + 7) / 8;                                   // row tiles in groups of 8 (one per XCD)
    const dim3 grid((unsigned)(groups * ntt * 8));
    hipLaunchKernelGGL((mmq_kernel<TYPE, MT>), grid, dim3(NTHREADS), (size_t)G::LDS_BYTES, st, W, row_bytes, n_rows, K, T, nrt, ntt,
                       q.qs, q.d, bh, bl, out, ld_out, resid);
    return hipGetLastError();
}

int g_mmq_mt = 0;   // 0 = by T; tools may force 1 / 2 / 4
int g_mmq_split = 0;   // tools: force the K split of the 128 x 256 kernel (0 = by shape, 1 = none)
int g_mmq_p2 = -1;     // tools: 0 = never the 128 x 256 kernel, 1 = by shape, -1 = MI355_MMQ_PLANES2 / default

template <int TYPE>
hipError_t launch_type(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                       float *out, int ld_out, const float *resid, hipStream_t st) {
    // one token tile per wave: the expansion is VALU-bound and larger tiles spill (measured 34 ms vs 40 ms per 512-token
    // prompt of the 8B model); MI355_MMQ_MT / mmq_set_tiles(2) select the 128 x 128 workgroup tile
    static const int env_mt = getenv("MI355_MMQ_MT") ? atoi(getenv("MI355_MMQ_MT")) : 0;
    int mt = 1;
    if (env_mt == 1 || env_mt == 2) mt = env_mt;
    if (g_mmq_mt == 1 || g_mmq_mt == 2) mt = g_mmq_mt;
    switch (mt) {
        case 2: return launch_one<TYPE, 2>(W, row_bytes, n_rows, K, T, q, bh, bl, out, ld_out, resid, st);
        default: return launch_one<TYPE, 1>(W, row_bytes, n_rows, K, T, q, bh, bl, out, ld_out, resid, st);
    }
}

}  // namespace

void mmq_set_tiles(int mt) { g_mmq_mt = mt; }
void mmq_set_split(int n) { g_mmq_split = n; }
void mmq_set_lds_form(int on) { g_mmq_p2 = on; }
#ifdef MI355_P2_PROBE
void mmq_p2_set_probe(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_p2_probe), &p, sizeof(p)); }
#endif

// continuous-batching decode steps and prompts up to a few hundred tokens: 3 <= T <= g_ksplit_max.  This kernel reads the GGUF
// bytes (0.56 B / weight) and expands them in registers, the planes kernel reads 2 B / weight it need not expand: measured on the
// 8B model the whole prompt takes 8.3 vs 13.7 ms at 128 tokens, 13.6 vs 16.7 at 256, 19.4 vs 21.0 at 384 and 22.7 vs 21.1 at 448
// (round 2: the LDS-form planes kernel moved the crossover down - whole prompt, K-split against planes: 13.8 vs 14.1 ms at 256 tokens,
// 17.2 vs 15.2 at 320, 19.1 vs 15.5 at 384)
static int g_ksplit_max = getenv("MI355_KSPLIT_MAX") ? atoi(getenv("MI355_KSPLIT_MAX")) : 256;
bool mmq_ksplit_applicable(int type, int K, int T) {
    // from 3 tokens: the mat-vec takes 4 + 2 + 1 tokens per pass over the weights, so 3 sequences cost 4.5 ms and 6 cost 6.4 ms a step against
    // 3.7 ms through this kernel; for 2 the mat-vec pass is cheaper (3.2 ms)
    return (type == T_Q4_K || type == T_Q5_K || type == T_Q6_K) && (K % 256) == 0 && T >= 3 && T <= g_ksplit_max;
}

// Which kernel takes a 3..256-token launch is a matter of shape (tools/exp_planes_shapes.py, Q5_K, us per launch):
//     N x K          T = 48     64     96    128    160    192    256
//   14336 x 4096   K-split 29     30     56     58     83     86    103      planes 256x32: 34 34 35 37 67 68 70   128x128: 55 56 56 57 60 60 61
//    4096 x 14336  K-split 38     39     41     43     77     80     83      planes 256x32: 99 .. 102              128x128: 166 .. 172
// a tensor of >= 8192 rows has enough 256-row tiles for the planes kernels to fill the chip from 65 tokens on (56 x 3 workgroups) and
// they read pre-expanded operands; a tensor of 4096 rows does not, and the K-split kernel's waves share its long K instead.
// (pair: gate | up of a dense layer go out as ONE K-split launch of 2 n_rows rows, which holds its own against two planes launches up to
// 128 tokens - whole 8B prompt 8.8 vs 9.0 ms at 100 tokens, 9.0 vs 9.2 at 128; past that the LDS-form SwiGLU launch takes the pair:
// 13.6 -> 10.6 ms at 200 tokens, 14.1 -> 11.1 at 256.)
bool mmq_ksplit_preferred(int type, int n_rows, int K, int T, bool has_planes, bool pair) {
    static const int env = getenv("MI355_KSPLIT_SHAPE") ? atoi(getenv("MI355_KSPLIT_SHAPE")) : 1;
    if (!mmq_ksplit_applicable(type, K, T)) return false;
    if (env && has_planes && T > (pair ? 128 : 64) && n_rows >= 8192 && mmq_applicable(type, K, T)) return false;
    return true;
}

// segs: up to 3 tensors sharing the activation; swiglu: segs = {gate, up} of one type and shape, out = silu(gate.x) * (up.x) into segs[0].out
hipError_t launch_mmq_ksplit_multi(const MMQSeg *segs, int n_seg, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                                   bool swiglu, hipStream_t st) {
    if (n_seg < 1 || n_seg > 3) return hipErrorInvalidValue;
    if (swiglu && (n_seg != 2 || segs[0].type != segs[1].type || segs[0].n_rows != segs[1].n_rows)) return hipErrorInvalidValue;
    KSplitArgs a{};
    a.n_seg = n_seg; a.K = K; a.T = T; a.swiglu = swiglu ? 1 : 0;
    a.aq = q.qs; a.ad = q.d; a.abh = bh; a.abl = bl;
    int tiles = 0;
    for (int i = 0; i < n_seg; i++) {
        a.seg[i] = segs[i];
        a.seg[i].tile0 = tiles;
        if (!swiglu || i == 0) tiles += (segs[i].n_rows + 31) / 32;
    }
    const int mt = T <= 32 ? 1 : 2;
    const dim3 grid((unsigned)tiles, (unsigned)((T + 32 * mt - 1) / (32 * mt)));
#define KS(MTV, SW)                                                                                                  \
    {                                                                                                                \
        constexpr int TOKV = 32 * MTV;                                                                               \
        size_t lds = (size_t)8 * (TOKV * A_STRIDE + 2 * TOKV * 16 + TOKV * 4);                                        \
        const size_t red = (size_t)8 * (SW ? 2 : 1) * MTV * 16 * 64 * 4;                                             \
        if (red > lds) lds = red;                                                                                    \
        if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_ksplit_kernel<MTV, SW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((mmq_ksplit_kernel<MTV, SW>), grid, dim3(NTHREADS), lds, st, a);                          \
    }
    if (swiglu) { if (mt == 1) KS(1, true) else KS(2, true) }
    else { if (mt == 1) KS(1, false) else KS(2, false) }
#undef KS
    return hipGetLastError();
}

hipError_t launch_mmq_ksplit(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q,
                             const int8_t *bh, const int8_t *bl, float *out, int ld_out, const float *resid, hipStream_t st) {
    MMQSeg sg{};
    sg.W = W; sg.row_bytes = row_bytes; sg.n_rows = n_rows; sg.type = type; sg.out = out; sg.ld_out = ld_out; sg.resid = resid;
    return launch_mmq_ksplit_multi(&sg, 1, K, T, q, bh, bl, false, st);
}

size_t mmq_planes_bytes(int type, int64_t n_rows, int K) {
    if ((type != T_Q4_K && type != T_Q5_K && type != T_Q6_K && type != T_Q2_K && type != T_Q3_K) || (K % 256) != 0) return 0;
    return (size_t)((n_rows + 31) / 32) * (size_t)(K >> 8) * PL_BLOCK;
}

hipError_t launch_mmq_expand(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, uint8_t *planes, hipStream_t st) {
    const int nb = K >> 8, n_rt = (n_rows + 31) / 32;
    const dim3 grid((unsigned)((n_rt * nb + 3) / 4));
    switch (type) {
        case T_Q4_K: hipLaunchKernelGGL(mmq_expand_kernel<T_Q4_K>, grid, dim3(256), 0, st, W, row_bytes, n_rows, nb, n_rt, planes); break;
        case T_Q5_K: hipLaunchKernelGGL(mmq_expand_kernel<T_Q5_K>, grid, dim3(256), 0, st, W, row_bytes, n_rows, nb, n_rt, planes); break;
        case T_Q6_K: hipLaunchKernelGGL(mmq_expand_kernel<T_Q6_K>, grid, dim3(256), 0, st, W, row_bytes, n_rows, nb, n_rt, planes); break;
        case T_Q2_K: hipLaunchKernelGGL(mmq_expand_small_kernel<T_Q2_K>, grid, dim3(256), 0, st, W, row_bytes, n_rows, nb, n_rt, planes); break;
        case T_Q3_K: hipLaunchKernelGGL(mmq_expand_small_kernel<T_Q3_K>, grid, dim3(256), 0, st, W, row_bytes, n_rows, nb, n_rt, planes); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_mmq_planes(int type, const uint8_t *planes, int n_rows, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                             float *out, int ld_out, const float *resid, hipStream_t st, MMQWorkspace wsp) {
    return launch_mmq_planes_multi(type, planes, &n_rows, &out, &ld_out, 1, K, T, q, bh, bl, resid, st, wsp, nullptr);
}

// the K split the 128 x 256 kernel would run a tensor (or concatenation) of n_rows rows with, and whether it takes the launch at all
static int planes2_split(int n_rows, int K, int T, const MMQWorkspace &wsp, bool *takes) {
    static const int env_mt = getenv("MI355_MMQ_MT") ? atoi(getenv("MI355_MMQ_MT")) : 0;
    static const int env_p2_0 = getenv("MI355_MMQ_PLANES2") ? atoi(getenv("MI355_MMQ_PLANES2")) : 1;
    const int env_p2 = g_mmq_p2 >= 0 ? g_mmq_p2 : env_p2_0;
    static const int env_sk = getenv("MI355_MMQ_SPLITK") ? atoi(getenv("MI355_MMQ_SPLITK")) : -1;   // 0 off, 2..4 forced
    const int nb = K >> 8;
    const long wg4 = (long)((n_rows + P2_ROWS - 1) / P2_ROWS) * ((T + P2_TOK - 1) / P2_TOK);
    int n_split = 1;
    if (g_mmq_split > 1 || env_sk > 1) n_split = g_mmq_split > 1 ? g_mmq_split : env_sk;
    else if (wg4 * 4 < 3L * num_cu() && env_sk != 0 && g_mmq_split != 1) {
        n_split = (int)((num_cu() + wg4 - 1) / wg4);
        if (n_split > 4) n_split = 4;
        if (n_split > nb / 8) n_split = nb / 8;                 // at least eight super-blocks per workgroup: a 4096-wide K split four ways
                                                                // (40.9 us with its reduction) loses to the 256 x 32 tiles (37.9), a 14336-wide one wins (89.7 vs 120.9)
    }
    if (n_split > nb) n_split = nb;
    if (n_split < 1 || (n_rows % 4) != 0 || !wsp.p || (size_t)n_split * T * n_rows * sizeof(float) > wsp.bytes) n_split = 1;
    *takes = g_mmq_mt == 4 || (g_mmq_mt == 0 && env_mt == 0 && env_p2 != 0 && T > 128 && wg4 * n_split * 4 >= 3L * num_cu());
    return n_split;
}

// Q4_K / Q5_K segments beside Q6_K ones in one launch: only the 128 x 256 kernel has that form, and segments must end on its row tiles
bool mmq_planes_mixed_ok(const int *seg_rows, int n_seg, int K, int T, MMQWorkspace wsp) {
    static const bool env_on = !(getenv("MI355_MMQ_MIXED") && getenv("MI355_MMQ_MIXED")[0] == '0');
    if (!env_on) return false;
    int n_rows = 0;
    for (int i = 0; i < n_seg; i++) { if (i + 1 < n_seg && (seg_rows[i] % P2_ROWS) != 0) return false; n_rows += seg_rows[i]; }
    bool takes = false;
    (void)planes2_split(n_rows, K, T, wsp, &takes);
    return takes && n_seg >= 1 && n_seg <= 3;
}

// ffn_gate | ffn_up of one plane type with SwiGLU in the epilogue: 64 rows of each per workgroup
bool mmq_planes_swiglu_ok(int type_gate, int type_up, int n_rows, int K, int T) {
    static const bool env_on = !(getenv("MI355_MMQ_SWIGLU") && getenv("MI355_MMQ_SWIGLU")[0] == '0');
    if (!env_on || type_gate != type_up || mmq_planes_bytes(type_gate, n_rows, K) == 0 || (n_rows % 32) != 0) return false;
    bool takes = false;
    (void)planes2_split(2 * n_rows, K, T, MMQWorkspace(), &takes);
    return takes;
}

hipError_t launch_mmq_planes_swiglu(int type, const uint8_t *planes_gate, const uint8_t *planes_up, int n_rows, int K, int T, const ActQuant &q,
                                    float *out, int ld_out, hipStream_t st) {
    if (!mmq_planes_swiglu_ok(type, type, n_rows, K, T)) return hipErrorInvalidValue;
    PlanesOut po{};
    po.n_seg = 1; po.out[0] = out; po.ld[0] = ld_out; po.row_end[0] = n_rows;
    const int nrt = (n_rows + 63) / 64, ntt = (T + P2_TOK - 1) / P2_TOK;
    const dim3 grid((unsigned)(((nrt + 7) / 8) * ntt * 8));
    if (planes_have_mins(type)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_swiglu_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_swiglu_kernel<true>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes_gate, planes_up, n_rows, K, T, nrt, ntt, q.qs, q.d, q.bsums, po);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_swiglu_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_swiglu_kernel<false>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes_gate, planes_up, n_rows, K, T, nrt, ntt, q.qs, q.d, q.bsums, po);
    }
    return hipGetLastError();
}

// the grouped (all experts, one launch) forms: `rows_max` = rows of the grouped arrays (tokens x experts used), meta = moe_group_kernel's counts and offsets
bool mmq_planes_moe_ok(int type, int n_rows, int K) {
    static const bool env_on = !(getenv("MI355_MOE_GROUPED_LAUNCH") && getenv("MI355_MOE_GROUPED_LAUNCH")[0] == '0');
    return env_on && (type == T_Q4_K || type == T_Q5_K || type == T_Q6_K) && mmq_planes_bytes(type, n_rows, K) != 0 && (n_rows % 128) == 0 && (K % 256) == 0;
}
hipError_t launch_mmq_planes_swiglu_moe(int type, const uint8_t *planes_gate, const uint8_t *planes_up, size_t plane_stride, int n_expert, const int32_t *meta,
                                        int n_rows, int K, int rows_max, const ActQuant &q, float *out, int ld_out, hipStream_t st) {
    if (!mmq_planes_moe_ok(type, n_rows, K) || n_expert < 1 || !meta || !q.qs || !q.d || !q.bsums) return hipErrorInvalidValue;
    PlanesOut po{};
    po.n_seg = 1; po.out[0] = out; po.ld[0] = ld_out; po.row_end[0] = n_rows;
    const int nrt = (n_rows + 63) / 64, slots = (rows_max + P2_TOK - 1) / P2_TOK + n_expert;     // every expert may end in a partial tile
    const dim3 grid((unsigned)(((nrt + 7) / 8) * slots * 8));
    const MoeTiles mt{meta, n_expert, plane_stride};
    if (planes_have_mins(type)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_swiglu_moe_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_swiglu_moe_kernel<true>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes_gate, planes_up, n_rows, K, nrt, slots, q.qs, q.d, q.bsums, po, mt);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_swiglu_moe_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_swiglu_moe_kernel<false>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes_gate, planes_up, n_rows, K, nrt, slots, q.qs, q.d, q.bsums, po, mt);
    }
    return hipGetLastError();
}
hipError_t launch_mmq_planes_moe(int type, const uint8_t *planes, size_t plane_stride, int n_expert, const int32_t *meta, int n_rows, int K, int rows_max,
                                 const ActQuant &q, float *out, int ld_out, hipStream_t st) {
    if (!mmq_planes_moe_ok(type, n_rows, K) || n_expert < 1 || !meta || !q.qs || !q.d || !q.bsums) return hipErrorInvalidValue;
    PlanesOut po{};
    po.n_seg = 1; po.out[0] = out; po.ld[0] = ld_out; po.row_end[0] = n_rows;
    const int nrt = (n_rows + P2_ROWS - 1) / P2_ROWS, slots = (rows_max + P2_TOK - 1) / P2_TOK + n_expert;
    const dim3 grid((unsigned)(((nrt + 7) / 8) * slots * 8));
    const MoeTiles mt{meta, n_expert, plane_stride};
    if (planes_have_mins(type)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_moe_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_moe_kernel<true>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes, n_rows, K, nrt, slots, q.qs, q.d, q.bsums, po, mt);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_moe_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
        hipLaunchKernelGGL((mmq_planes2_moe_kernel<false>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes, n_rows, K, nrt, slots, q.qs, q.d, q.bsums, po, mt);
    }
    return hipGetLastError();
}

hipError_t launch_mmq_planes_multi(int type, const uint8_t *planes, const int *seg_rows, float *const *outs, const int *lds_out, int n_seg, int K, int T,
                                   const ActQuant &q, const int8_t *bh, const int8_t *bl, const float *resid, hipStream_t st, MMQWorkspace wsp,
                                   const int *seg_types) {
    if (n_seg < 1 || n_seg > 3 || (n_seg > 1 && resid)) return hipErrorInvalidValue;
    PlanesOut po{};
    po.n_seg = n_seg;
    int n_rows = 0;
    bool mixed = false;
    for (int i = 0; i < n_seg; i++) {
        if (i + 1 < n_seg && (seg_rows[i] % 32) != 0) return hipErrorInvalidValue;
        n_rows += seg_rows[i];
        po.out[i] = outs[i]; po.ld[i] = lds_out[i]; po.row_end[i] = n_rows;
        const int ti = seg_types ? seg_types[i] : type;
        if (planes_have_mins(ti)) po.mins_mask |= 1u << i;
        mixed = mixed || planes_have_mins(ti) != planes_have_mins(type);
    }
    if (mixed && !mmq_planes_mixed_ok(seg_rows, n_seg, K, T, wsp)) return hipErrorInvalidValue;
    static const int env_mt = getenv("MI355_MMQ_MT") ? atoi(getenv("MI355_MMQ_MT")) : 0;
    // 128 x 128 workgroup tiles (two token tiles per wave) when that still yields ~2 workgroups per CU, else 256 x 32
    // tiles, which quadruple the workgroup count (measured on the 8B shapes: gate/up 297 vs 383 us, N = 4096 tensors
    // 54-144 vs 75-204 us per layer)
    const long wg2 = (long)((n_rows + 127) / 128) * ((T + 127) / 128);
    int mt = (T <= 32 || wg2 < 3L * num_cu() / 2) ? 1 : 2;
    if (n_seg > 1 && T > 32 && wg2 * 4 >= 3L * num_cu()) mt = 2;       // concatenated Q | K | V: 192 workgroups of 128 x 128 beat 384 of 256 x 32 (59 vs 71 us)
    if (env_mt == 1 || env_mt == 2) mt = env_mt;
    if (g_mmq_mt == 1 || g_mmq_mt == 2) mt = g_mmq_mt;
    const bool mins = planes_have_mins(type);
    // both operands through LDS (128 rows x 256 tokens per workgroup) once that grid covers most of the chip; tensors with
    // too few rows for that split K over 2..4 workgroups (partial sums in the caller's workspace, added up in split order)
    bool p2 = false;
    const int n_split = planes2_split(n_rows, K, T, wsp, &p2);
    if (p2) {
        const int nrt = (n_rows + P2_ROWS - 1) / P2_ROWS, ntt = (T + P2_TOK - 1) / P2_TOK;
        const dim3 grid((unsigned)(((nrt + 7) / 8) * ntt * n_split * 8));
        if (mixed) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
            hipLaunchKernelGGL((mmq_planes2_kernel<2>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes, n_rows, K, T, nrt, ntt, q.qs, q.d, q.bsums, po, resid, wsp.p, n_split);
        } else if (mins) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
            hipLaunchKernelGGL((mmq_planes2_kernel<true>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes, n_rows, K, T, nrt, ntt, q.qs, q.d, q.bsums, po, resid, wsp.p, n_split);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS);
            hipLaunchKernelGGL((mmq_planes2_kernel<false>), grid, dim3(NTHREADS), (size_t)P2_LDS, st, planes, n_rows, K, T, nrt, ntt, q.qs, q.d, q.bsums, po, resid, wsp.p, n_split);
        }
        if (n_split > 1) {
            const size_t n4 = (size_t)T * (n_rows >> 2);
            hipLaunchKernelGGL(mmq_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, wsp.p, n_split, T, n_rows, po, resid);
        }
        return hipGetLastError();
    }
#define PLN(MINSV, MTV)                                                                                                   \
    {                                                                                                                     \
        using G = Geo<MTV>;                                                                                               \
        const int nrt = (n_rows + G::ROW_TILE - 1) / G::ROW_TILE, ntt = (T + G::TOK_TILE - 1) / G::TOK_TILE;              \
        const dim3 grid((unsigned)(((nrt + 7) / 8) * ntt * 8));                                                           \
        const size_t lds2 = 2 * (size_t)G::LDS_BYTES;                                                                     \
        if (lds2 > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mmq_planes_kernel<MINSV, MTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
        hipLaunchKernelGGL((mmq_planes_kernel<MINSV, MTV>), grid, dim3(NTHREADS), lds2, st, planes, n_rows, K, T, nrt, ntt, \
                           q.qs, q.d, bh, bl, po, resid);                                                                 \
    }
    if (mins) { if (mt == 1) PLN(true, 1) else PLN(true, 2) }
    else { if (mt == 1) PLN(false, 1) else PLN(false, 2) }
#undef PLN
    return hipGetLastError();
}

bool mmq_applicable(int type, int K, int T) {
    return (type == T_Q4_K || type == T_Q5_K || type == T_Q6_K) && (K % 256) == 0 && T >= 32 && (size_t)T * (size_t)K < (1ull << 32);
}
size_t mmq_prep_bytes(int K, int T) { return (size_t)T * (K >> 4); }

hipError_t launch_mmq_prep(const ActQuant &q, int K, int T, int8_t *bh, int8_t *bl, hipStream_t st) {
    const size_t n = (size_t)T * (K >> 4);
    hipLaunchKernelGGL(mmq_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q.bsums, n, bh, bl);
    return hipGetLastError();
}

// out[t * ld_out + r] = (resid ? resid[...] : 0) + dot(W[r], act[t]);  bh/bl from launch_mmq_prep
hipError_t launch_mmq(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q,
                      const int8_t *bh, const int8_t *bl, float *out, int ld_out, const float *resid, hipStream_t st) {
    switch (type) {
        case T_Q4_K: return launch_type<T_Q4_K>(W, row_bytes, n_rows, K, T, q, bh, bl, out, ld_out, resid, st);
        case T_Q5_K: return launch_type<T_Q5_K>(W, row_bytes, n_rows, K, T, q, bh, bl, out, ld_out, resid, st);
        case T_Q6_K: return launch_type<T_Q6_K>(W, row_bytes, n_rows, K, T, q, bh, bl, out, ld_out, resid, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace mi355
