// loads probe_patched.hsaco (see probe_kernel.hip) and says whether the kernel arguments were preloaded by the hardware
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    hipModule_t m; hipFunction_t f;
    CK(hipModuleLoad(&m, argc > 1 ? argv[1] : "tools/preload_probe/probe_patched.hsaco"));
    CK(hipModuleGetFunction(&f, m, "preload_probe"));
    float *a, *b, *o; CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&o, 256));
    std::vector<float> h(64, 2.0f);
    CK(hipMemcpy(a, h.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), 256, hipMemcpyHostToDevice));
    float e = 7.0f;
    void *args[] = {&a, &b, &e, &o};
    CK(hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr));
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), o, 256, hipMemcpyDeviceToHost));
    printf("out[0] = %g -> %s\n", h[0], h[0] == 11.0f ? "kernel arguments PRELOADED by the hardware (entered behind the header)" : h[0] == 127.0f ? "NOT preloaded: the compatibility header ran" : "unexpected");
    return 0;
}
