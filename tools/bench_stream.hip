// tools/bench_stream.hip — access-pattern microbenchmarks behind the mmvq design (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// A: pattern of the register-path mmvq: per 144-B super-block one 16-B header (shared by 8 lanes) + 8 x 16 B
template <int NT_LOAD>
__global__ __launch_bounds__(256) void pattern_kernel(const uint8_t *W, size_t row_bytes, int n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    unsigned acc = 0;
    for (int r = gw * 2; r < n_rows; r += nw * 2) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t *row = W + (size_t)(r + (k & 1)) * row_bytes;
            const int sb = (k >> 1) * 8 + (lane >> 3);
            const uint8_t *b = row + (size_t)sb * 144;
            if (NT_LOAD) {
                const u32x4_t h = __builtin_nontemporal_load((const u32x4_t *)b), q = __builtin_nontemporal_load((const u32x4_t *)(b + 16 + (lane & 7) * 16));
                v[2 * k] = make_uint4(h.x, h.y, h.z, h.w); v[2 * k + 1] = make_uint4(q.x, q.y, q.z, q.w);
            } else {
                v[2 * k] = *(const uint4 *)b; v[2 * k + 1] = *(const uint4 *)(b + 16 + (lane & 7) * 16);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

// B: contiguous row streaming, 16 B per lane, registers
__global__ __launch_bounds__(256) void contig_kernel(const uint8_t *W, size_t row_bytes, int n_rows, unsigned *sink) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    unsigned acc = 0;
    for (int r = gw * 2; r < n_rows; r += nw * 2) {
        const uint8_t *row = W + (size_t)r * row_bytes;     // two consecutive rows = 2*row_bytes contiguous
        uint4 v[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const size_t off = (size_t)k * 1024 + lane * 16;
            v[k] = off < 2 * row_bytes ? *(const uint4 *)(row + off) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

// C: LDS-DMA ring per wave: chunk = one row (row_bytes), DEPTH chunks in flight, consumer xors the chunk from LDS
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define C(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
        C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32) C(33) C(34) C(35) C(36) C(37) C(38) C(39) C(40)
#undef C
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const uint8_t *W, size_t row_bytes, int n_rows, unsigned *sink, int ring_bytes, int nt) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    uint8_t *ring = smem + (size_t)wave * ring_bytes;
    const int slot = (int)row_bytes;                    // 16-B multiple
    const int ns = ring_bytes / slot;
    const int ni = (slot + 1023) / 1024;
    const int nrows_w = gw < n_rows ? (n_rows - gw + nw - 1) / nw : 0;
    auto issue = [&](int j) {
        const uint8_t *row = W + (size_t)(gw + (size_t)j * nw) * row_bytes;
        uint8_t *dst = ring + (size_t)(j % ns) * slot;
        for (int i = 0; i < ni; i++) {
            const int off = i * 1024 + lane * 16;
            if (off < slot) {
                if (nt) __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(row + off), (void __attribute__((address_space(3))) *)(dst + i * 1024), 16, 0, 2);
                else __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(row + off), (void __attribute__((address_space(3))) *)(dst + i * 1024), 16, 0, 0);
            }
        }
    };
    const int depth = ns - 1;
    int issued = 0;
    for (; issued < depth && issued < nrows_w; issued++) issue(issued);
    unsigned acc = 0;
    for (int j = 0; j < nrows_w; j++) {
        const int younger = issued - j - 1;              // chunks issued after chunk j
        wait_vm(younger * ni);
        const uint8_t *src = ring + (size_t)(j % ns) * slot;
        for (int off = lane * 16; off < slot; off += 1024) {
            const uint4 v = *(const uint4 *)(src + off);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
        asm volatile("" ::: "memory");
        if (issued < nrows_w) { issue(issued); issued++; }
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}

int main(int argc, char **argv) {
    const int n_rows = argc > 1 ? atoi(argv[1]) : 28672;
    const size_t row_bytes = argc > 2 ? atoi(argv[2]) : 2304;
    const int nbuf = 6, iters = 30;
    const size_t bytes = row_bytes * n_rows;
    std::vector<uint8_t *> W(nbuf);
    for (auto &w : W) { hipMalloc(&w, bytes + 4096); hipMemset(w, 0x11, bytes + 4096); }
    unsigned *sink; hipMalloc(&sink, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; i++) launch(i);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < iters; i++) launch(i);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters;
        printf("%-44s %8.2f us  %6.0f GB/s\n", name, us, bytes / (us * 1e-6) / 1e9);
    };
    char nm[128];
    for (int blocks : {512, 1024, 2048}) {
        snprintf(nm, sizeof nm, "A pattern hdr+qs regs       blocks=%d", blocks);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(pattern_kernel<0>, dim3(blocks), dim3(256), 0, nullptr, W[i % nbuf], row_bytes, n_rows, sink); });
        snprintf(nm, sizeof nm, "A pattern hdr+qs regs nt    blocks=%d", blocks);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(pattern_kernel<1>, dim3(blocks), dim3(256), 0, nullptr, W[i % nbuf], row_bytes, n_rows, sink); });
        snprintf(nm, sizeof nm, "B contiguous rows regs      blocks=%d", blocks);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(contig_kernel, dim3(blocks), dim3(256), 0, nullptr, W[i % nbuf], row_bytes, n_rows, sink); });
    }
    for (int ring : {7168, 14336}) for (int nt : {0, 1}) for (int bpc : {1, 2}) {
        const int blocks = 256 * bpc;
        snprintf(nm, sizeof nm, "C LDS-DMA ring=%d nt=%d 4 waves x %d blk/CU", ring, nt, bpc);
        timeit(nm, [&](int i) { hipLaunchKernelGGL(dma_kernel<4>, dim3(blocks), dim3(256), 4 * ring, nullptr, W[i % nbuf], row_bytes, n_rows, sink, ring, nt); });
        if (bpc == 1) {
            snprintf(nm, sizeof nm, "C LDS-DMA ring=%d nt=%d 8 waves x 1 blk/CU", ring, nt);
            timeit(nm, [&](int i) { hipLaunchKernelGGL(dma_kernel<8>, dim3(256), dim3(512), 8 * ring, nullptr, W[i % nbuf], row_bytes, n_rows, sink, ring, nt); });
        }
    }
    return 0;
}
