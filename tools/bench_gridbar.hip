// bench_gridbar.hip — cost of a device-wide barrier inside a persistent kernel on MI355X (decode_mega.hip's building block).
// WPC workgroups per CU spin until everybody has arrived; variants:
//   mode 0: one counter, relaxed atomic add + relaxed poll, no cache maintenance
//   mode 3: mode 0 + agent-scope release fence before the add and acquire fence after the poll (every workgroup)
//   mode 5: two-level counters: 8 group counters (blockIdx % 8, one cache line each), the last arriver of a group bumps the
//           top counter, everybody polls the top counter
//   mode 6: no atomics: every workgroup stores its epoch into its own slot; workgroup 0 polls all slots and then stores a
//           release word that everybody else polls
//   mode 7: no atomics: slots as in 6, every workgroup polls all slots itself
//   mode 8: two-level slots: 8 group masters (blocks 0..7) poll their group's slots and write group-done words; everybody
//           polls the 8 group-done words (one 64-byte read)
// build: hipcc --offload-arch=gfx950 -O3 tools/bench_gridbar.hip -o tools/bin/bench_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LIMIT 300000

// words: [0] counter | [32..32+8*32) group counters (stride 32 words) | [512] release | [1024 .. 1024+grid) slots | [4096 .. +8*32) group-done
template <int MODE>
__global__ void bar_kernel(unsigned *w, int n_bar, float *sink, int payload) {
    extern __shared__ unsigned char smem[];
    unsigned epoch = 0;
    float acc = 0.0f;
    const int nwg = gridDim.x, me = blockIdx.x, lane = threadIdx.x;
    for (int b = 0; b < n_bar; b++) {
        if (payload) sink[(size_t)me * blockDim.x + threadIdx.x] = acc + (float)b;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        epoch++;
        if (threadIdx.x < 64) {                                   // wave 0
            int polls = 0;
            if (MODE == 0 || MODE == 3) {
                if (lane == 0) {
                    if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (LD(w) < epoch * nwg && ++polls < LIMIT) __builtin_amdgcn_s_sleep(2);
                    if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            } else if (MODE == 5) {
                if (lane == 0) {
                    const int g = me & 7, gsize = (nwg - g + 7) / 8;
                    const unsigned old = __hip_atomic_fetch_add(w + 32 + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old + 1 == epoch * gsize) __hip_atomic_fetch_add(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (LD(w) < epoch * 8 && ++polls < LIMIT) __builtin_amdgcn_s_sleep(2);
                }
            } else if (MODE == 6) {
                if (lane == 0) ST(w + 1024 + me, epoch);
                if (me == 0) {
                    bool all = false;
                    while (!all && ++polls < LIMIT) {
                        bool ok = true;
                        for (int i = lane; i < nwg; i += 64) ok &= LD(w + 1024 + i) >= epoch;
                        all = __all(ok);
                    }
                    if (lane == 0) ST(w + 512, epoch);
                } else if (lane == 0) {
                    while (LD(w + 512) < epoch && ++polls < LIMIT) __builtin_amdgcn_s_sleep(2);
                }
            } else if (MODE == 7) {
                if (lane == 0) ST(w + 1024 + me, epoch);
                bool all = false;
                while (!all && ++polls < LIMIT) {
                    bool ok = true;
                    for (int i = lane; i < nwg; i += 64) ok &= LD(w + 1024 + i) >= epoch;
                    all = __all(ok);
                    if (!all) __builtin_amdgcn_s_sleep(2);
                }
            } else if (MODE == 8) {
                if (lane == 0) ST(w + 1024 + me, epoch);
                if (me < 8) {                                      // master of group me: workgroups me, me + 8, ...
                    bool all = false;
                    while (!all && ++polls < LIMIT) {
                        bool ok = true;
                        for (int i = lane; me + 8 * i < nwg; i += 64) ok &= LD(w + 1024 + me + 8 * i) >= epoch;
                        all = __all(ok);
                    }
                    if (lane == 0) ST(w + 4096 + me, epoch);
                }
                bool all = false;
                while (!all && ++polls < LIMIT) {
                    const bool ok = lane < 8 ? LD(w + 4096 + lane) >= epoch : true;
                    all = __all(ok);
                    if (!all) __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (payload) acc += sink[(size_t)((me + 37) % nwg) * blockDim.x + threadIdx.x];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int n_bar = argc > 1 ? atoi(argv[1]) : 160;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    unsigned *words; float *sink;
    hipMalloc(&words, 8192 * 4); hipMalloc(&sink, (size_t)1024 * 512 * 4);
    hipMemset(sink, 0, (size_t)1024 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int modes[6] = {0, 3, 5, 6, 7, 8};
    for (int wpc = 1; wpc <= 2; wpc++) {
        const int grid = wpc * prop.multiProcessorCount, nt = wpc == 1 ? 512 : 256;
        for (int payload = 0; payload < 2; payload++)
        for (int mi = 0; mi < 6; mi++) {
            const int mode = modes[mi];
            float best = 1e9f;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(words, 0, 8192 * 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                const size_t lds = wpc == 1 ? 100 * 1024 : 60 * 1024;
                switch (mode) {
                    case 0: hipLaunchKernelGGL(bar_kernel<0>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                    case 3: hipLaunchKernelGGL(bar_kernel<3>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                    case 5: hipLaunchKernelGGL(bar_kernel<5>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                    case 6: hipLaunchKernelGGL(bar_kernel<6>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                    case 7: hipLaunchKernelGGL(bar_kernel<7>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                    default: hipLaunchKernelGGL(bar_kernel<8>, dim3(grid), dim3(nt), lds, 0, words, n_bar, sink, payload); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("wg/cu=%d payload=%d mode=%d grid=%d: %.2f us per barrier (%.3f ms)\n", wpc, payload, mode, grid, best * 1000.0f / n_bar, best);
        }
    }
    return 0;
}
