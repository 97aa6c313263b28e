// tools/bench_gs.hip — register-resident inner loops of the prompt contraction, two waves per SIMD (not part of the product):
//   P2: two int8 planes with the 6-bit group scale folded into the weights: 2 MFMAs per (tile, group) accumulating over a super-block,
//       then shift-add + convert + 2 fma per output and super-block (the shipping mmq_planes2 arithmetic);
//   GS: raw codes as the operand: 1 MFMA per (tile, group) into a fresh accumulator, the group scale applied per output with v_mad_i32_i24
//       (one scale per lane: the lane's weight row), then convert + 2 fma per output and super-block.
// Operands come from a few KiB of random bytes and change every K-step (power draw depends on toggling bits); time per super-block step.
// build: hipcc --offload-arch=gfx950 -O3 tools/bench_gs.hip -o tools/bin/bench_gs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x16 mfma(i32x4 a, i32x4 b, i32x16 c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int mad24(int a, int b, int c) { int d; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ int mul24(int a, int b) { int d; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

template <int MODE>   // 0 = P2, 1 = GS, 2 = GS without the scale multiply (MFMA count of GS alone), 3 = P2 MFMAs alone
__global__ __launch_bounds__(512) void k(const i32x4 *ops, const float *scales, int n_sb, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float facc[4][16];
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) facc[t][r] = 0.0f;
    i32x4 a[4][2], bh[2], bl[2];
    for (int j = 0; j < 2; j++) {
        for (int t = 0; t < 4; t++) a[t][j] = ops[(wave * 16 + t * 2 + j) * 64 + lane];
        bh[j] = ops[(wave * 16 + 8 + j) * 64 + lane]; bl[j] = ops[(wave * 16 + 10 + j) * 64 + lane];
    }
    const float dd = scales[lane], ndm = scales[64 + lane];
    unsigned scw0 = (unsigned)ops[lane].x & 0x3f3f3f3fu, scw1 = (unsigned)ops[lane].y & 0x3f3f3f3fu;
    for (int sb = 0; sb < n_sb; sb++) {
        const float yd = scales[128 + ((sb + lane) & 63)];
        if (MODE == 0 || MODE == 3) {
            i32x16 H[4], L[4];
#pragma unroll
            for (int g = 0; g < 8; g++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (g == 0) { i32x16 z; for (int r = 0; r < 16; r++) z[r] = 0; H[t] = mfma(a[t][0], bh[0], z); L[t] = mfma(a[t][0], bl[0], z); }
                    else { H[t] = mfma(a[t][g & 1], bh[g & 1], H[t]); L[t] = mfma(a[t][g & 1], bl[g & 1], L[t]); }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    if (MODE == 3) { facc[t][r] += (float)(H[t][r] + L[t][r]); continue; }
                    const int isum = (H[t][r] << 5) + L[t][r];
                    const float t0 = fmaf(dd, (float)isum, ndm * (float)(r + sb));
                    facc[t][r] = fmaf(yd, t0, facc[t][r]);
                }
        } else {
            int isum[4][16];
#pragma unroll
            for (int g = 0; g < 8; g++) {
                const int sc = (int)(((g < 4 ? scw0 : scw1) >> (8 * (g & 3))) & 0xffu);
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    i32x16 z;
                    for (int r = 0; r < 16; r++) z[r] = 0;
                    const i32x16 res = mfma(a[t][g & 1], bh[g & 1], z);
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (MODE == 2) isum[t][r] = g == 0 ? res[r] : isum[t][r] + res[r];
                        else isum[t][r] = g == 0 ? mul24(res[r], sc) : mad24(res[r], sc, isum[t][r]);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float t0 = fmaf(dd, (float)isum[t][r], ndm * (float)(r + sb));
                    facc[t][r] = fmaf(yd, t0, facc[t][r]);
                }
        }
        // new operands every super-block (rotating through the table keeps the loop free of loads in the K-steps themselves)
        const int nx = ((sb + 1) & 7) * 8192;
        for (int j = 0; j < 2; j++) {
            for (int t = 0; t < 4; t++) a[t][j] = ops[nx + (wave * 16 + t * 2 + j) * 64 + lane];
            bh[j] = ops[nx + (wave * 16 + 8 + j) * 64 + lane]; bl[j] = ops[nx + (wave * 16 + 10 + j) * 64 + lane];
        }
    }
    float s = 0;
    for (int t = 0; t < 4; t++) for (int r = 0; r < 16; r++) s += facc[t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int n_sb = argc > 1 ? atoi(argv[1]) : 4096;
    const bool zeros = argc > 2 && atoi(argv[2]) == 1;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<int> h(8 * 8192 * 4);
    unsigned rs = 1234567u;
    for (auto &v : h) { rs = rs * 1664525u + 1013904223u; v = zeros ? 0x01010101 : (int)((rs >> 4) & 0x0f0f0f0fu) | (int)((rs << 9) & 0x70707070u & 0); }
    // weights: 4-bit codes (0..15) in every byte; activations: full int8 range in the A slots
    for (int s = 0; s < 8; s++) for (int w = 0; w < 8; w++) for (int t = 0; t < 8; t++) for (int l = 0; l < 64 * 4; l++) {
        rs = rs * 1664525u + 1013904223u;
        if (!zeros) h[(size_t)s * 8192 * 4 + ((w * 16 + t) * 64) * 4 + l] = (int)(rs ^ (rs >> 11));
    }
    std::vector<float> sc(256);
    for (int i = 0; i < 256; i++) sc[i] = 0.001f * (i + 1);
    i32x4 *ops; float *scales, *out;
    hipMalloc(&ops, h.size() * 4); hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&scales, 1024); hipMemcpy(scales, sc.data(), 1024, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)cus * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto kern, double mfma_per_sb) {
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, nullptr, ops, scales, n_sb, out); hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(512), 0, nullptr, ops, scales, n_sb, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // algorithmic work per super-block step per wave: 4 tiles x 32 x 32 x 256 MACs
        const double alg = (double)cus * 8 * n_sb * 4.0 * 32 * 32 * 256 * 2;
        printf("%-34s %8.1f us   %7.1f ns per super-block step   algorithmic %6.0f TOP/s   MFMA issue %6.0f TOP/s\n", name, ms * 1e3, ms * 1e6 / n_sb,
               alg / (ms * 1e-3) / 1e12, (double)cus * 8 * n_sb * mfma_per_sb * 65536.0 / (ms * 1e-3) / 1e12);
    };
    for (int rep = 0; rep < 2; rep++) {
        run("P2 two planes + fold", k<0>, 64);
        run("P2 MFMAs alone", k<3>, 64);
        run("GS one plane + mad24 + fold", k<1>, 32);
        run("GS MFMAs + plain add", k<2>, 32);
    }
    return 0;
}
