// tools/preload_probe: does this box's firmware / runtime PRELOAD kernel arguments into SGPRs (hipcc -mllvm -amdgpu-kernarg-preload-count)?
// The compiler emits a 256-byte compatibility header in front of such a kernel that s_loads the preloaded arguments itself; hardware that preloads enters
// behind it.  build.sh patches the header of this kernel so that it loads 123.0 instead of the argument `e`: out = a * b + e  -> preload active,
// out = a * b + 123 -> the header ran (no preload).
#include <hip/hip_runtime.h>
extern "C" __global__ void preload_probe(const float *a, const float *b, float e, float *out) {
    const int i = threadIdx.x;
    out[i] = a[i] * b[i] + e;
}
