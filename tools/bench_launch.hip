// tools/bench_launch.hip — fixed per-kernel costs on this GPU: launch-to-launch gap and one dependent memory round trip.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void rt_kernel(const int *src, int *dst, int hops) {   // `hops` dependent global loads per thread
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    for (int h = 0; h < hops; h++) idx = src[idx];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = idx;
}
int main() {
    const int N = 1 << 20;
    int *src, *dst; hipMalloc(&src, N * 4); hipMalloc(&dst, N * 4);
    int *h = new int[N]; for (int i = 0; i < N; i++) h[i] = (int)(((long)i * 7919 + 12345) % N);
    hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, int iters, auto f) {
        for (int i = 0; i < 20; i++) f();
        hipDeviceSynchronize(); hipEventRecord(e0);
        for (int i = 0; i < iters; i++) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-50s %7.2f us per launch\n", name, ms * 1e3 / iters);
    };
    timeit("empty kernel, 1 block", 2000, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, nullptr, (int *)nullptr); });
    timeit("empty kernel, 1024 blocks x 256", 2000, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, nullptr, (int *)nullptr); });
    for (int hops : {1, 2, 3, 4})
        for (int blocks : {64, 1024}) {
            char nm[96]; snprintf(nm, sizeof nm, "%d dependent load(s), %d blocks x 256", hops, blocks);
            timeit(nm, 1000, [&] { hipLaunchKernelGGL(rt_kernel, dim3(blocks), dim3(256), 0, nullptr, src, dst, hops); });
        }
    // graph of 64 empty kernels
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 64; i++) hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(256), 0, st, (int *)nullptr);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; i++) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st); hipEventRecord(e0, st);
    for (int i = 0; i < 50; i++) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-50s %7.2f us per kernel\n", "hipGraph of 64 empty kernels (1024x256)", ms * 1e3 / 50 / 64);
    return 0;
}
