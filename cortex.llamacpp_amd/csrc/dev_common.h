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
    T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q8_0 = 8, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15,
};

// ---- ggml on-disk block sizes ------------------------------------------------------------
__host__ __device__ constexpr int ggml_block_elems(int t) {
    return (t == T_F32 || t == T_F16) ? 1 : (t == T_Q4_0 || t == T_Q8_0) ? 32 : 256;
}
__host__ __device__ constexpr int ggml_block_bytes(int t) {
    return t == T_F32 ? 4 : t == T_F16 ? 2 : t == T_Q4_0 ? 18 : t == T_Q8_0 ? 34 : t == T_Q4_K ? 144 :
           t == T_Q5_K ? 176 : t == T_Q6_K ? 210 : t == T_Q8_K ? 292 : 0;
}
__host__ __device__ inline size_t ggml_row_bytes(int t, int64_t n) {
    return (size_t)(n / ggml_block_elems(t)) * (size_t)ggml_block_bytes(t);
}

// ---- device-resident weight row layouts --------------------------------------------------
// Q4_K / Q5_K rows stay in ggml order (144 / 176 B super-blocks are 16-B aligned: header | [qh] | qs).
// Q6_K (210 B) and Q8_0 (34 B) blocks are not 16-B aligned, so at upload each ROW is regrouped into
// aligned planes (same bytes, same total size up to padding):
//   Q6_K row: [ql: nb*128][qh: nb*64][scales: nb*16][d: nb*2] padded to 16
//   Q8_0 row: [qs: K][d: K/32*2] padded to 16
// F16 / F32 rows are unchanged.
__host__ __device__ inline size_t dev_row_bytes(int t, int64_t K) {
    size_t b = ggml_row_bytes(t, K);
    return (b + 15) & ~(size_t)15;
}

// ---- wave-level reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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
