// tools/bench_mmvq.hip — stand-alone timing harness for the quantised mat-vec kernel (not part of the product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cortex.llamacpp_amd/csrc tools/bench_mmvq.hip \
//        cortex.llamacpp_amd/csrc/mmvq.hip -o gpurun_out/bench_mmvq        (mmvq.hip provides set_error via stub below)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
namespace mi355 { void set_error(const char *, ...) {} }
using namespace mi355;

__global__ __launch_bounds__(256) void stream_kernel(const uint4 *src, size_t n16, unsigned *sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) *sink = acc;
}

int main(int argc, char **argv) {
    const int type = argc > 1 ? atoi(argv[1]) : T_Q4_K;
    const int N = argc > 2 ? atoi(argv[2]) : 14336;
    const int K = argc > 3 ? atoi(argv[3]) : 4096;
    const int epi = argc > 4 ? atoi(argv[4]) : EPI_STORE;       // 2 = swiglu (two matrices of N rows)
    const int fuse = argc > 5 ? atoi(argv[5]) : 0;
    const int nbuf = argc > 6 ? atoi(argv[6]) : 8;
    const int iters = 40;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); set_num_cu(prop.multiProcessorCount);
    const size_t rb = dev_row_bytes(type, K);
    const int nmat = epi == EPI_SWIGLU ? 2 : 1;
    const size_t wbytes = rb * (size_t)N;
    std::vector<uint8_t *> W(nbuf * nmat);
    std::vector<uint8_t> host(wbytes);
    for (size_t i = 0; i < wbytes; i++) host[i] = (uint8_t)(rand() & 0x3f);   // small fp16 exponents: finite scales
    for (auto &w : W) { hipMalloc(&w, wbytes); hipMemcpy(w, host.data(), wbytes, hipMemcpyHostToDevice); }
    int8_t *aq; float *ad; int16_t *abs_; uint16_t *ad0; float *out, *x, *nw; unsigned *sink;
    hipMalloc(&aq, K); hipMalloc(&ad, (K / 256) * 4); hipMalloc(&abs_, (K / 16) * 2); hipMalloc(&ad0, (K / 32) * 2);
    hipMalloc(&out, (size_t)N * 4 * 2); hipMalloc(&x, K * 4); hipMalloc(&nw, K * 4); hipMalloc(&sink, 16);
    hipMemset(aq, 1, K); hipMemset(ad, 0, (K / 256) * 4); hipMemset(abs_, 0, (K / 16) * 2); hipMemset(ad0, 0, (K / 32) * 2);
    std::vector<float> hx(K, 0.5f); hipMemcpy(x, hx.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(nw, hx.data(), K * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0, (size_t)N * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int i) {
        MMVQArgs a{};
        a.n_seg = nmat; a.K = K; a.T = 1; a.epi = epi;
        for (int s = 0; s < nmat; s++) {
            a.seg[s].W = W[(i % nbuf) * nmat + s]; a.seg[s].out = out + (size_t)s * N; a.seg[s].resid = out; a.seg[s].type = type;
            a.seg[s].n_rows = N; a.seg[s].ld_out = N; a.seg[s].row_bytes = rb;
        }
        a.aq = aq; a.ad = ad; a.abs = abs_; a.aq0 = aq; a.ad0 = ad0;
        a.fuse_mode = fuse; a.nx = x; a.nw = nw; a.neps = 1e-5f; a.nck = getenv("MI355_EXP") ? atoi(getenv("MI355_EXP")) : 0;
        hipError_t e = launch_mmvq(a, nullptr);
        if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
    };
    for (int i = 0; i < 8; i++) run(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < iters; i++) run(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, gb = (double)wbytes * nmat / 1e9;
    printf("mmvq type=%d N=%d K=%d epi=%d fuse=%d: %.2f us/launch, %.1f MB, %.0f GB/s", type, N, K, epi, fuse, us, gb * 1e3, gb / (us * 1e-6));
    {   // the same launches replayed from a hipGraph (what the decode step does): no host launch-rate floor
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t g; hipGraphExec_t ge;
        auto run_st = [&](int i) {
            MMVQArgs a{};
            a.n_seg = nmat; a.K = K; a.T = 1; a.epi = epi;
            for (int s = 0; s < nmat; s++) {
                a.seg[s].W = W[(i % nbuf) * nmat + s]; a.seg[s].out = out + (size_t)s * N; a.seg[s].resid = out; a.seg[s].type = type;
                a.seg[s].n_rows = N; a.seg[s].ld_out = N; a.seg[s].row_bytes = rb;
            }
            a.aq = aq; a.ad = ad; a.abs = abs_; a.aq0 = aq; a.ad0 = ad0;
            a.fuse_mode = fuse; a.nx = x; a.nw = nw; a.neps = 1e-5f; a.nck = getenv("MI355_EXP") ? atoi(getenv("MI355_EXP")) : 0;
            launch_mmvq(a, st);
        };
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 32; i++) run_st(i);
        hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; i++) hipGraphLaunch(ge, st);
        hipStreamSynchronize(st); hipEventRecord(e0, st);
        for (int i = 0; i < 10; i++) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double usg = ms * 1e3 / 320;
        printf("   | in hipGraph: %.2f us, %.0f GB/s\n", usg, gb / (usg * 1e-6));
    }
    // plain streaming read of the same buffers for comparison
    for (int blocks : {1024, 2048, 4096}) {
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++)
            for (int s = 0; s < nmat; s++)
                hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)W[(i % nbuf) * nmat + s], wbytes / 16, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double us2 = ms * 1e3 / iters;
        printf("  stream blocks=%d: %.2f us, %.0f GB/s\n", blocks, us2, gb / (us2 * 1e-6));
    }
    return 0;
}
