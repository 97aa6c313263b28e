// tools/bench_mfma.hip — issue-rate microbenchmark of the int8 and f16 MFMAs used by the prefill kernels (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 tools/bench_mfma.hip -o tools/bin/bench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_i8(int iters, int *sink) {
    i32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0;
    i32x4 x = {(int)threadIdx.x, 1, 2, 3}, y = {4, 5, 6, (int)blockIdx.x};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, acc[a], 0, 0, 0);
    }
    int s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    if (s == 0x12345678) *sink = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_f16(int iters, float *sink) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0;
    f16x8 x, y;
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(threadIdx.x & 3); y[i] = (_Float16)(i & 1); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    if (s == 1.2345f) *sink = s;
}
int main() {
    int *sink; hipMalloc(&sink, 16);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch, double ops_per_mfma, int nacc, int waves_per_simd) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)cus * 4 * waves_per_simd * iters * nacc;
        printf("%-28s acc=%d waves/SIMD=%d: %.1f TOP/s  (%.1f cycles per MFMA per SIMD at %.2f GHz)\n", name, nacc, waves_per_simd,
               n * ops_per_mfma / (ms * 1e-3) / 1e12, (ms * 1e-3) * p.clockRate * 1e3 / ((double)iters * nacc * waves_per_simd), p.clockRate / 1e6);
    };
    for (int w = 1; w <= 2; w++) {
        run("mfma_i32_32x32x32_i8", [&] { hipLaunchKernelGGL(k_i8<4>, dim3(cus * w), dim3(256), 0, nullptr, iters, sink); }, 2.0 * 32 * 32 * 32, 4, w);
        run("mfma_f32_32x32x16_f16", [&] { hipLaunchKernelGGL(k_f16<4>, dim3(cus * w), dim3(256), 0, nullptr, iters, (float *)sink); }, 2.0 * 32 * 32 * 16, 4, w);
    }
    run("mfma_i32_32x32x32_i8 (1 acc)", [&] { hipLaunchKernelGGL(k_i8<1>, dim3(cus), dim3(256), 0, nullptr, iters, sink); }, 2.0 * 32 * 32 * 32, 1, 1);
    return 0;
}
