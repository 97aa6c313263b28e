// dev_common.h — shared device/host helpers for the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace mi355 {

constexpr int WAVE = 64;
constexpr int QK_K = 256;

// ggml type ids (GGUF on-disk values)
enum : int {
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q5_0 = 6, T_Q8_0 = 8, T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15, T_IQ4_NL = 20,
};

// ---- ggml on-disk block sizes ------------------------------------------------------------
__host__ __device__ constexpr int ggml_block_elems(int t) {
    return (t == T_F32 || t == T_F16) ? 1 : (t == T_Q4_0 || t == T_Q5_0 || t == T_Q8_0 || t == T_IQ4_NL) ? 32 : 256;
}
__host__ __device__ constexpr int ggml_block_bytes(int t) {
    return t == T_F32 ? 4 : t == T_F16 ? 2 : t == T_Q4_0 ? 18 : t == T_Q8_0 ? 34 : t == T_Q4_K ? 144 :
           t == T_Q5_K ? 176 : t == T_Q6_K ? 210 : t == T_Q8_K ? 292 : t == T_Q2_K ? 84 : t == T_Q3_K ? 110 : t == T_Q5_0 ? 22 : t == T_IQ4_NL ? 18 : 0;
}
__host__ __device__ inline size_t ggml_row_bytes(int t, int64_t n) {
    return (size_t)(n / ggml_block_elems(t)) * (size_t)ggml_block_bytes(t);
}

// the CPU backend's activation format for a weight type: Q8_0 blocks for the 32-element formats, Q8_K for the K-quants
__host__ __device__ constexpr bool act_is_q80(int t) { return t == T_Q8_0 || t == T_Q4_0 || t == T_Q5_0 || t == T_IQ4_NL; }

// ---- device-resident weight row layouts --------------------------------------------------
// Q4_K / Q5_K rows stay in ggml order (144 / 176 B super-blocks are 16-B aligned: header | [qh] | qs).
// Q6_K (210 B) and Q8_0 (34 B) blocks are not 16-B aligned, so at upload each ROW is regrouped into
// aligned planes (same bytes, same total size up to padding):
//   Q6_K row: [ql: nb*128][qh: nb*64][scales: nb*16][d: nb*2] padded to 16
//   Q8_0 row: [qs: K][d: K/32*2] padded to 16
//   Q2_K row (84 B blocks: scales 16 | qs 64 | d | dmin):      [qs: nb*64][scales: nb*16][d, dmin: nb*4] padded to 16
//   Q3_K row (110 B blocks: hmask 32 | qs 64 | scales 12 | d): [hmask: nb*32][qs: nb*64][scales: nb*12][d: nb*2] padded to 16
//   Q4_0 / IQ4_NL row (18 B blocks: d | qs 16):              [qs: K/2][d: K/32*2] padded to 16
//   Q5_0 row (22 B blocks: d | qh 4 | qs 16):                 [qs: K/2][qh: K/32*4][d: K/32*2] padded to 16
// F16 / F32 rows are unchanged.
__host__ __device__ inline size_t dev_row_bytes(int t, int64_t K) {
    size_t b = ggml_row_bytes(t, K);
    return (b + 15) & ~(size_t)15;
}

// ---- wave-level reductions ---------------------------------------------------------------
// DPP butterflies inside each 16-lane row (VALU speed, no LDS crossbar), then the four row totals are read
// with v_readlane and combined in a fixed order; the result is wave-uniform (valid in every lane).
// ---- device-coherent accesses (agent scope, no cache-wide fence) for data exchanged between workgroups of ONE running
// kernel (decode_mega.hip).  MI355X has one L2 per XCD: a plain store may sit in the writer's L2 and a plain load may hit
// a stale line in the reader's.  An agent-scope release / acquire pair fixes that with buffer_wbl2 / buffer_inv over the
// whole L2, which costs ~25 us when 512 workgroups do it at once (tools/bench_gridbar.hip).  Marking just the exchanged
// accesses sc1 (write-through to / read from the coherence point) costs nothing extra: they are a few KB per phase.
// COH = false compiles to the plain access.  `base` must be wave-uniform; byte offsets below 2 GB.
typedef unsigned int coh_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int coh_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7ffffff0, 0x00020000);
}
template <bool COH> __device__ __forceinline__ coh_u32x4 cld16(const void *base, int byte_off) {
    if (COH) return __builtin_amdgcn_raw_buffer_load_b128(coh_rsrc(base), byte_off, 0, 16);
    return *reinterpret_cast<const coh_u32x4 *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <bool COH> __device__ __forceinline__ coh_u32x2 cld8(const void *base, int byte_off) {
    if (COH) return __builtin_amdgcn_raw_buffer_load_b64(coh_rsrc(base), byte_off, 0, 16);
    return *reinterpret_cast<const coh_u32x2 *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <bool COH> __device__ __forceinline__ unsigned cld4(const void *base, int byte_off) {
    if (COH) return __builtin_amdgcn_raw_buffer_load_b32(coh_rsrc(base), byte_off, 0, 16);
    return *reinterpret_cast<const unsigned *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <bool COH> __device__ __forceinline__ int cld2s(const void *base, int byte_off) {          // int16 -> int
    if (COH) return (int)(short)__builtin_amdgcn_raw_buffer_load_b16(coh_rsrc(base), byte_off, 0, 16);
    return (int)*reinterpret_cast<const short *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <bool COH> __device__ __forceinline__ float cldf(const float *p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> __device__ __forceinline__ void cstf(float *p, float v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH> __device__ __forceinline__ void cst4(void *p, unsigned v) {
    if (COH) __hip_atomic_store(reinterpret_cast<unsigned *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *reinterpret_cast<unsigned *>(p) = v;
}
template <bool COH> __device__ __forceinline__ void cst2(void *p, unsigned short v) {
    if (COH) __hip_atomic_store(reinterpret_cast<unsigned short *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *reinterpret_cast<unsigned short *>(p) = v;
}

// threadIdx.x through an opaque move: what is derived from it cannot be hoisted out of an enclosing loop or merged
// with another phase's copy.  decode_mega.hip runs several kernels' bodies inside one layer loop; with the plain builtin
// the compiler computed every phase's lane constants once at the top and kept them in scratch.
__device__ __forceinline__ int tid_now() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }

template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
constexpr int DPP_QP_1032 = 0xB1;    // quad_perm [1,0,3,2]
constexpr int DPP_QP_2301 = 0x4E;    // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;
constexpr int DPP_MIRROR = 0x140;

__device__ __forceinline__ float row_sum16(float v) {       // every lane of a 16-lane row gets the row total
    v += dpp_f<DPP_QP_1032>(v);
    v += dpp_f<DPP_QP_2301>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int wave_sum(int v) {
    v += dpp_i<DPP_QP_1032>(v);
    v += dpp_i<DPP_QP_2301>(v);
    v += dpp_i<DPP_HALF_MIRROR>(v);
    v += dpp_i<DPP_MIRROR>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ double dpp_d(double v, int which) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    switch (which) {
        case 0: lo = dpp_i<DPP_QP_1032>(lo); hi = dpp_i<DPP_QP_1032>(hi); break;
        case 1: lo = dpp_i<DPP_QP_2301>(lo); hi = dpp_i<DPP_QP_2301>(hi); break;
        case 2: lo = dpp_i<DPP_HALF_MIRROR>(lo); hi = dpp_i<DPP_HALF_MIRROR>(hi); break;
        default: lo = dpp_i<DPP_MIRROR>(lo); hi = dpp_i<DPP_MIRROR>(hi); break;
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d(v, 0); v += dpp_d(v, 1); v += dpp_d(v, 2); v += dpp_d(v, 3);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QP_1032>(v));
    v = fmaxf(v, dpp_f<DPP_QP_2301>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#define MI355_MAX64_STEP(CTRL)                                                                        \
    {                                                                                                 \
        const unsigned lo = (unsigned)dpp_i<CTRL>((int)(v & 0xffffffffull));                          \
        const unsigned hi = (unsigned)dpp_i<CTRL>((int)(v >> 32));                                    \
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;                             \
        v = w > v ? w : v;                                                                            \
    }
    MI355_MAX64_STEP(DPP_QP_1032) MI355_MAX64_STEP(DPP_QP_2301) MI355_MAX64_STEP(DPP_HALF_MIRROR) MI355_MAX64_STEP(DPP_MIRROR)
#undef MI355_MAX64_STEP
    unsigned long long r = 0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), l);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        r = w > r ? w : r;
    }
    return r;
}

__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(f)); }

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

}  // namespace mi355

#define HIP_CHECK_RET(expr, ret)                                                          \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            mi355::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return ret;                                                                   \
        }                                                                                 \
    } while (0)

namespace mi355 {
void set_error(const char *fmt, ...);
}
