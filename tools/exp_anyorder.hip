// tools/exp_anyorder.hip — can a stream dispatch dependent kernels without the end-of-kernel barrier between them?
// Each kernel waits IN the kernel for its predecessor's completion word, reads `bytes` of memory, and the last workgroup to finish
// publishes its own completion word.  Launched (a) the ordinary way and (b) with hipExtAnyOrderLaunch; per-kernel time, host
// time per launch, and how far a kernel's first wave starts before its predecessor's completion (negative = no overlap).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Slot { unsigned done; unsigned arrived; unsigned long long t_first, t_done; unsigned timeout; unsigned pad[9]; };   // 64 B

__global__ __launch_bounds__(256) void step_kernel(const u32x4 *src, size_t n16, Slot *slots, int idx, unsigned *sink) {
    Slot *me = slots + idx;
    if (threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (atomicAdd(&me->arrived, 1u) == 0) me->t_first = t;
    }
    if (idx > 0) {                                              // wait for the predecessor (bounded)
        if (threadIdx.x == 0) {
            volatile unsigned *d = &slots[idx - 1].done;
            int spins = 0;
            while (__hip_atomic_load(d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                if (++spins > (1 << 20)) { me->timeout = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { const u32x4 a = __builtin_nontemporal_load(src + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) *sink = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&me->arrived, 0x10000u) >> 16 == gridDim.x - 1) {
            me->t_done = __builtin_amdgcn_s_memrealtime();
            __hip_atomic_store(&me->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 64;
    const size_t bytes = (argc > 2 ? atoi(argv[2]) : 16) * (size_t)1 << 20;
    u32x4 *src; CK(hipMalloc(&src, bytes * 8)); CK(hipMemset(src, 1, bytes * 8));
    Slot *slots; CK(hipMalloc(&slots, sizeof(Slot) * N));
    unsigned *sink; CK(hipMalloc(&sink, 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *name, float ms, double host_us) {
        std::vector<Slot> h(N); CK(hipMemcpy(h.data(), slots, sizeof(Slot) * N, hipMemcpyDeviceToHost));
        double early = 0, dur = 0; int to = 0;
        for (int i = 1; i < N; i++) { early += ((double)h[i - 1].t_done - (double)h[i].t_first) * 0.01; dur += ((double)h[i].t_done - (double)h[i - 1].t_done) * 0.01; to += h[i].timeout; }
        printf("%-44s %7.2f us per kernel (events)  host %5.2f us per launch   first wave enters %+6.2f us before the predecessor completes, completion to completion %5.2f us, timeouts %d\n",
               name, ms * 1e3 / N, host_us / N, early / (N - 1), dur / (N - 1), to);
    };
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemsetAsync(slots, 0, sizeof(Slot) * N, st));
            CK(hipStreamSynchronize(st));
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            const auto h0 = std::chrono::steady_clock::now();
            if (mode == 2) CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            else CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; i++) {
                const u32x4 *s = src + (size_t)(i & 7) * (bytes / 16);
                if (mode == 0) hipLaunchKernelGGL(step_kernel, dim3(256), dim3(256), 0, st, s, bytes / 16, slots, i, sink);
                else hipExtLaunchKernelGGL(step_kernel, dim3(256), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, s, bytes / 16, slots, i, sink);
            }
            const auto h1 = std::chrono::steady_clock::now();
            if (mode == 2) {
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st));
            }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) report(mode == 0 ? "ordinary launches" : mode == 1 ? "hipExtAnyOrderLaunch, eager" : "hipExtAnyOrderLaunch, captured in a graph", ms, std::chrono::duration<double, std::micro>(h1 - h0).count());
            if (ge) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
        }
    }
    return 0;
}
