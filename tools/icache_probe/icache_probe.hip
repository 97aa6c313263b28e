// tools/icache_probe: what does code that runs ONCE per launch cost on this chip?  A kernel executes the same straight-line block of N VALU instructions
// (16 KB of code) twice: the first pass fetches every line of it, the second finds it in the instruction cache.  Launched several times back to back, it
// also shows whether a launch boundary leaves the instruction cache warm.  (Round 5: the quantising prologues of the decode mat-vecs run once per launch and
// take 3-4x what their instruction counts say.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int DEP>
__global__ void straight(unsigned long long *out, float *sink, int waves_per_wg) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (DEP) asm volatile(".rept 4096\n\tv_add_f32 %0, %0, %1\n\t.endr" : "+v"(a) : "v"(b));
    else asm volatile(".rept 1024\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    // the same block again through a backward branch is not possible with .rept text: a second copy would be other addresses; so loop over ONE copy instead
    out[((size_t)blockIdx.x * waves_per_wg + (threadIdx.x >> 6)) * 4 + 0] = t1 - t0;
    (void)t2;
    if (a + b + c + d == 12345.0f) *sink = a;
}
template <int DEP>
__global__ void looped(unsigned long long *out, float *sink, int waves_per_wg) {
    float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
    unsigned long long t[3];
    for (int it = 0; it < 2; it++) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[it]) :: "memory");
        if (DEP) asm volatile(".rept 4096\n\tv_add_f32 %0, %0, %1\n\t.endr" : "+v"(a) : "v"(b));
        else asm volatile(".rept 1024\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t.endr" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[2]) :: "memory");
    unsigned long long *o = out + ((size_t)blockIdx.x * waves_per_wg + (threadIdx.x >> 6)) * 4;
    if ((threadIdx.x & 63) == 0) { o[0] = t[1] - t[0]; o[1] = t[2] - t[1]; }
    if (a + b + c + d == 12345.0f) *sink = a;
}

static void report(const char *name, const std::vector<unsigned long long> &h, int n, int col) {
    std::vector<double> v;
    for (int i = 0; i < n; i++) v.push_back((double)h[(size_t)i * 4 + col]);
    std::sort(v.begin(), v.end());
    printf("  %-44s min %8.0f  median %8.0f  max %8.0f cycles  (%.2f cycles per instruction at the median)\n", name, v.front(), v[v.size() / 2], v.back(), v[v.size() / 2] / 4096.0);
}

int main() {
    unsigned long long *out; float *sink;
    CK(hipMalloc(&out, 256 * 16 * 4 * 8)); CK(hipMalloc(&sink, 4));
    for (int wpw : {1, 8}) {
        for (int dep = 0; dep < 2; dep++) {
            printf("%d wave(s) per workgroup, 256 workgroups, 4096 x v_add_f32 (16 KB of code), %s:\n", wpw, dep ? "one dependent chain" : "four independent chains");
            std::vector<unsigned long long> h((size_t)256 * wpw * 4);
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(out, 0, h.size() * 8));
                if (dep) hipLaunchKernelGGL(looped<1>, dim3(256), dim3(64 * wpw), 0, 0, out, sink, wpw);
                else hipLaunchKernelGGL(looped<0>, dim3(256), dim3(64 * wpw), 0, 0, out, sink, wpw);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
                char nm[96];
                snprintf(nm, sizeof nm, "launch %d, first pass (code not yet fetched?)", rep); report(nm, h, 256 * wpw, 0);
                snprintf(nm, sizeof nm, "launch %d, second pass (instruction cache warm)", rep); report(nm, h, 256 * wpw, 1);
            }
        }
    }
    return 0;
}
