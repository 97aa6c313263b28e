// tools/exp_p2.hip — the prompt contraction on pre-expanded planes: per-lane planes kernel (128 x 128 tiles) against the kernel with both
// operands through LDS (128 x 256): bit-for-bit comparison on synthetic planes, time per launch (HIP events), and with
// -DMI355_P2_PROBE the per-wave cycle split of workgroup 0.
// build: tools/build_exp_p2.sh
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <algorithm>
#include "kernels.h"
namespace mi355 {
void set_error(const char *, ...) {}
static int g_ncu = 256;
void set_num_cu(int n) { g_ncu = n; }
int num_cu() { return g_ncu; }
void mmq_set_tiles(int mt);
void mmq_set_split(int n);
#ifdef MI355_P2_PROBE
void mmq_p2_set_probe(unsigned long long *p);
#endif
}
using namespace mi355;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static unsigned rs = 12345u;
static unsigned rnd() { rs = rs * 1664525u + 1013904223u; return rs >> 8; }
static uint16_t f2h_host(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }

__global__ void tick_kernel(unsigned long long *o) {          // s_memtime ticks per 100 us of the 100 MHz wall clock
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_readcyclecounter();
    while (__builtin_amdgcn_s_memrealtime() - r0 < 10000) {}
    o[0] = __builtin_readcyclecounter() - t0;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 4096, T = argc > 3 ? atoi(argv[3]) : 2048;
    const int type = argc > 4 ? atoi(argv[4]) : T_Q4_K, reps = argc > 5 ? atoi(argv[5]) : 20;
    const int nb = K / 256, n_rt = (N + 31) / 32;
    const size_t PLB = 8 * 2 * 1024 + 32 * 16;
    std::vector<uint8_t> hp((size_t)n_rt * nb * PLB);
    for (size_t b = 0; b < (size_t)n_rt * nb; b++) {
        uint8_t *blk = hp.data() + b * PLB;
        for (int i = 0; i < 16384; i++) blk[i] = (uint8_t)(rnd() & (type == T_Q6_K ? 0xff : 0x3f));
        for (int r = 0; r < 32; r++) {
            uint32_t *m = reinterpret_cast<uint32_t *>(blk + 16384 + r * 16);
            m[0] = f2h_host(0.001f + (rnd() & 255) * 1e-5f) | ((uint32_t)f2h_host(0.0005f + (rnd() & 255) * 1e-5f) << 16);
            m[1] = rnd() & 0x3f3f3f3f; m[2] = (rnd() * 77u) & 0x3f3f3f3f; m[3] = 0;
        }
    }
    uint8_t *planes; CK(hipMalloc(&planes, hp.size())); CK(hipMemcpy(planes, hp.data(), hp.size(), hipMemcpyHostToDevice));
    std::vector<int8_t> haq((size_t)T * K); std::vector<float> had((size_t)T * nb); std::vector<int16_t> hbs((size_t)T * K / 16);
    for (auto &v : haq) v = (int8_t)((int)(rnd() % 255) - 127);
    for (auto &v : had) v = 0.01f + (rnd() & 1023) * 1e-4f;
    for (size_t i = 0; i < hbs.size(); i++) { int sum = 0; for (int k = 0; k < 16; k++) sum += haq[i * 16 + k]; hbs[i] = (int16_t)sum; }
    ActQuant q{};
    CK(hipMalloc(&q.qs, haq.size())); CK(hipMemcpy(q.qs, haq.data(), haq.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&q.d, had.size() * 4)); CK(hipMemcpy(q.d, had.data(), had.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&q.bsums, hbs.size() * 2)); CK(hipMemcpy(q.bsums, hbs.data(), hbs.size() * 2, hipMemcpyHostToDevice));
    int8_t *bh, *bl; CK(hipMalloc(&bh, mmq_prep_bytes(K, T))); CK(hipMalloc(&bl, mmq_prep_bytes(K, T)));
    CK(launch_mmq_prep(q, K, T, bh, bl, nullptr));
    float *o_old, *o_new; CK(hipMalloc(&o_old, (size_t)T * N * 4)); CK(hipMalloc(&o_new, (size_t)T * N * 4));
    CK(hipMemset(o_old, 0, (size_t)T * N * 4)); CK(hipMemset(o_new, 0xff, (size_t)T * N * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    MMQWorkspace wsp; wsp.bytes = (size_t)4 * T * N * 4; CK(hipMalloc(&wsp.p, wsp.bytes));
    auto run = [&](int tiles, float *out, int split = 0) {
        mmq_set_tiles(tiles); mmq_set_split(split);
        CK(launch_mmq_planes(type, planes, N, K, T, q, bh, bl, out, N, nullptr, nullptr, wsp));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) CK(launch_mmq_planes(type, planes, N, K, T, q, bh, bl, out, N, nullptr, nullptr, wsp));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f / reps;
    };
    const float us1 = run(1, o_old, 1), us2 = run(2, o_old, 1);
#ifdef MI355_P2_PROBE
    unsigned long long *probe; CK(hipMalloc(&probe, 64 * 8)); CK(hipMemset(probe, 0, 64 * 8)); mmq_p2_set_probe(probe);
#endif
    const float us4 = run(4, o_new, 1);
#ifdef MI355_P2_PROBE
    unsigned long long hpz[64]; CK(hipMemcpy(hpz, probe, sizeof(hpz), hipMemcpyDeviceToHost));   // (of the unsplit launch)
#endif
    float *o_sk; CK(hipMalloc(&o_sk, (size_t)T * N * 4));
    float us_sk[5] = {0, 0, 0, 0, 0};
    for (int sp = 2; sp <= 4; sp++) us_sk[sp] = run(4, o_sk, sp);
    {
        std::vector<float> a((size_t)T * N), b((size_t)T * N);
        CK(hipMemcpy(a.data(), o_new, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o_sk, b.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; for (size_t i = 0; i < a.size(); i++) { md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
        printf("  split K (+ reduce): x2 %.1f us  x3 %.1f us  x4 %.1f us   max |diff| vs unsplit %.3g of max |y| %.3g\n", us_sk[2], us_sk[3], us_sk[4], md, mx);
    }
    std::vector<float> a((size_t)T * N), b((size_t)T * N);
    CK(hipMemcpy(a.data(), o_old, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), o_new, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < a.size(); i++) if (memcmp(&a[i], &b[i], 4)) bad++;
    const double top = 2.0 * N * K * T / 1e6;
    printf("N %d K %d T %d type %d: 256x32 %.1f us (%.0f TOP/s)  128x128 %.1f us (%.0f)  lds 128x256 %.1f us (%.0f)  mismatches %zu / %zu  sample %g\n",
           N, K, T, type, us1, top / us1, us2, top / us2, us4, top / us4, bad, a.size(), (double)b[12345 % b.size()]);
#ifdef MI355_P2_PROBE
    { unsigned long long *tk, htk = 0; CK(hipMalloc(&tk, 8)); hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(64), 0, 0, tk); CK(hipMemcpy(&htk, tk, 8, hipMemcpyDeviceToHost));
      printf("  s_memtime: %.1f ticks per us\n", htk / 100.0); }
    printf("  workgroup 0, cycles per wave over %d super-blocks:   first half: compute / dma wait / barrier   second half: compute / dma wait / barrier   last fold\n", nb);
    for (int w = 0; w < 8; w++) printf("   wave %d: %8llu %8llu %8llu    %8llu %8llu %8llu    %8llu\n", w, hpz[w * 8], hpz[w * 8 + 1], hpz[w * 8 + 2], hpz[w * 8 + 3], hpz[w * 8 + 4], hpz[w * 8 + 5], hpz[w * 8 + 6]);
#endif
    return bad ? 1 : 0;
}
