// tools/exp_stream.hip — the decode mat-vecs of one Llama-3-8B layer (+ lm-head), mmvq_fast (register ring) against
// mmvq_stream (LDS-DMA ring): bit-for-bit comparison of the outputs, then in-graph time per launch of each kernel over
// NL distinct weight sets (1 GB: nothing stays in the Infinity Cache) and of the four-kernel layer chain.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I cortex.llamacpp_amd/csrc tools/exp_stream.hip \
//        cortex.llamacpp_amd/csrc/mmvq.hip cortex.llamacpp_amd/csrc/mmvq_fast.hip cortex.llamacpp_amd/csrc/mmvq_stream.hip -o tools/bin/exp_stream
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "kernels.h"
namespace mi355 { void set_error(const char *, ...) {} }
using namespace mi355;
#ifdef MI355_STREAM_PROBE
namespace mi355 { void mmvq_stream_set_probe(unsigned long long *p); }
#endif
#include <algorithm>
#ifndef MI355_STREAM_NL
#define MI355_STREAM_NL 8
#endif

#ifndef MI355_ST_RING
#define MI355_ST_RING 131072
#endif
#define MI355_ST_RING_KIB (MI355_ST_RING / 1024)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint8_t *dev_rand_bytes(size_t n, unsigned seed, int mask) {
    std::vector<uint8_t> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)((s >> 24) & mask); }
    uint8_t *d; CK(hipMalloc(&d, n + 4096)); CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
    return d;
}
static float *dev_rand_f32(size_t n, unsigned seed, float amp) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 777u;
    for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; h[i] = amp * ((float)(int)(s >> 8) / 8388608.0f - 1.0f); }
    float *d; CK(hipMalloc(&d, n * 4 + 4096)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

// ---- read floors: the same bytes with no arithmetic
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_gs_kernel(const u32x4_t *src, size_t n16, unsigned *sink) {   // chip-wide moving window
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4_t a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                      c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const u32x4_t a = __builtin_nontemporal_load(src + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) *sink = acc;
}
__device__ __forceinline__ void x_dma16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void x_wait(int n) {
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// 8 waves per CU, a 16 KiB ring per wave filled 4 KiB at a time by DMA, nothing read back.  mode 0: each wave a contiguous
// run; 1: the 8 waves of a workgroup interleave 4 KiB groups over the workgroup's contiguous run; 2: all waves of the chip
// interleave (one moving window)
__global__ __launch_bounds__(512) void read_dma_kernel(const uint8_t *src, size_t bytes, int mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ngroups = (int)(bytes / 4096);
    const int nw = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
    int first, stride, count;
    if (mode == 0) { const int per = (ngroups + nw - 1) / nw; first = gw * per; stride = 1; count = first + per <= ngroups ? per : (ngroups > first ? ngroups - first : 0); }
    else if (mode == 1) { const int per = (ngroups + gridDim.x - 1) / gridDim.x; const int lo = blockIdx.x * per, hi = lo + per < ngroups ? lo + per : ngroups;
                          first = lo + wave; stride = 8; count = hi > first ? (hi - first + 7) / 8 : 0; }
    else { first = gw; stride = nw; count = ngroups > first ? (ngroups - first + nw - 1) / nw : 0; }
    const unsigned ring = (unsigned)(uintptr_t)(smem + wave * 16384);
    auto issue = [&](int k) {
        const uint8_t *g = src + (size_t)(first + (size_t)k * stride) * 4096 + lane * 16;
        const unsigned dst = ring + (k & 3) * 4096;
        x_dma16(g, dst); x_dma16(g + 1024, dst + 1024); x_dma16(g + 2048, dst + 2048); x_dma16(g + 3072, dst + 3072);
    };
    int issued = 0;
    for (; issued < count && issued < 4; issued++) issue(issued);
    for (int k = 0; k < count; k++) {
        x_wait((issued - 1 - k) * 4);
        if (issued < count) { issue(issued); issued++; }
    }
}

struct Op {
    std::string name;
    int n_seg, type[3], N[3], K, epi, fuse;
    std::vector<uint8_t *> W[3];     // per weight set
    float *out_a, *out_b;            // old / new outputs (sum of N floats)
    size_t bytes;
};

static const int E = 4096, FF = 14336, NL = MI355_STREAM_NL;   // 8 sets = 1 GB: nothing stays in the Infinity Cache; 1 set: everything does
static int8_t *g_aq; static float *g_ad; static int16_t *g_abs; static float *g_x, *g_nw, *g_ffn, *g_resid;

static MMVQArgs make_args(const Op &op, int set, float *out) {
    MMVQArgs a{};
    a.n_seg = op.n_seg; a.K = op.K; a.T = 1; a.epi = op.epi; a.fuse_mode = op.fuse;
    size_t o = 0;
    for (int s = 0; s < op.n_seg; s++) {
        a.seg[s].W = op.W[s][set]; a.seg[s].type = op.type[s]; a.seg[s].n_rows = op.N[s]; a.seg[s].ld_out = op.N[s];
        a.seg[s].row_bytes = dev_row_bytes(op.type[s], op.K);
        a.seg[s].out = out + o; a.seg[s].resid = g_resid;
        o += op.N[s];
    }
    a.aq = g_aq; a.ad = g_ad; a.abs = g_abs;
    a.nx = op.fuse == 2 ? g_ffn : g_x; a.nw = g_nw; a.neps = 1e-5f;
    return a;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); set_num_cu(prop.multiProcessorCount);
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    g_aq = (int8_t *)dev_rand_bytes(FF, 1, 0xff);
    g_ad = dev_rand_f32(FF / 256, 2, 0.01f);
    g_abs = (int16_t *)dev_rand_bytes(FF / 16 * 2, 3, 0x3f);
    g_x = dev_rand_f32(E, 4, 1.0f); g_nw = dev_rand_f32(E, 5, 1.0f); g_ffn = dev_rand_f32(FF, 6, 1.0f); g_resid = dev_rand_f32(131072, 7, 1.0f);

    std::vector<Op> ops;
    auto add = [&](const char *name, int n_seg, std::initializer_list<int> types, std::initializer_list<int> Ns, int K, int epi, int fuse, int nsets) {
        Op op; op.name = name; op.n_seg = n_seg; op.K = K; op.epi = epi; op.fuse = fuse; op.bytes = 0;
        int i = 0; for (int t : types) op.type[i++] = t;
        i = 0; for (int n : Ns) op.N[i++] = n;
        size_t tot = 0;
        for (int s = 0; s < n_seg; s++) {
            const size_t b = dev_row_bytes(op.type[s], K) * (size_t)op.N[s];
            op.bytes += b; tot += op.N[s];
            for (int l = 0; l < nsets; l++) op.W[s].push_back(dev_rand_bytes(b, 100 + (unsigned)ops.size() * 31 + s * 7 + l, 0x3f));
        }
        CK(hipMalloc(&op.out_a, tot * 4)); CK(hipMalloc(&op.out_b, tot * 4));
        ops.push_back(op);
    };
    add("qkv   q4k 4096|1024|1024 x4096 rmsnorm", 3, {T_Q4_K, T_Q4_K, T_Q4_K}, {4096, 1024, 1024}, E, EPI_STORE, 1, NL);
    add("qkv   q4k q,k + q6k v    x4096 rmsnorm", 3, {T_Q4_K, T_Q4_K, T_Q6_K}, {4096, 1024, 1024}, E, EPI_STORE, 1, 2);
    add("o     q4k 4096x4096 planes +resid     ", 1, {T_Q4_K}, {4096}, E, EPI_ADD, 0, NL);
    add("gateup q4k 2x14336x4096 rmsnorm swiglu", 2, {T_Q4_K, T_Q4_K}, {FF, FF}, E, EPI_SWIGLU, 1, NL);
    add("down  q4k 4096x14336 quant +resid     ", 1, {T_Q4_K}, {4096}, FF, EPI_ADD, 2, NL);
    add("down  q6k 4096x14336 quant +resid     ", 1, {T_Q6_K}, {4096}, FF, EPI_ADD, 2, NL);
    add("down  q4k 4096x14336 PLANES +resid    ", 1, {T_Q4_K}, {4096}, FF, EPI_ADD, 0, NL);      // (round 4: what ffn_down costs without its quantisation prologue)
    add("head  q6k 128256x4096 rmsnorm         ", 1, {T_Q6_K}, {128256}, E, EPI_STORE, 1, 2);
    add("o     q5k 4096x4096 planes +resid     ", 1, {T_Q5_K}, {4096}, E, EPI_ADD, 0, 2);
    add("o     q6k 4096x4096 planes +resid     ", 1, {T_Q6_K}, {4096}, E, EPI_ADD, 0, 2);

    hipStream_t st; CK(hipStreamCreate(&st));
    // ---- 1. bit-for-bit
    int bad_total = 0;
    for (auto &op : ops) {
        size_t tot = 0; for (int s = 0; s < (op.epi == EPI_SWIGLU ? 1 : op.n_seg); s++) tot += op.N[s];
        CK(hipMemset(op.out_a, 0xff, tot * 4)); CK(hipMemset(op.out_b, 0xee, tot * 4));
        MMVQArgs a = make_args(op, 0, op.out_a), b = make_args(op, 0, op.out_b);
        if (!mmvq_fast_applicable(a)) { printf("%s: fast kernel not applicable\n", op.name.c_str()); continue; }
        if (!mmvq_stream_applicable(b)) { printf("%s: stream kernel not applicable\n", op.name.c_str()); continue; }
        CK(launch_mmvq_fast(a, st));
        CK(hipStreamSynchronize(st));
        std::vector<uint32_t> ha(tot), hb(tot);
        CK(hipMemcpy(ha.data(), op.out_a, tot * 4, hipMemcpyDeviceToHost));
        for (int mode = 0; mode < 1; mode++) {
            CK(hipMemset(op.out_b, 0xee, tot * 4));
            CK(launch_mmvq_stream(b, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hb.data(), op.out_b, tot * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < tot; i++) if (ha[i] != hb[i]) { if (!bad) first = i; bad++; }
            float fa, fb; memcpy(&fa, &ha[first], 4); memcpy(&fb, &hb[first], 4);
            printf("%s: %zu outputs, %zu differ%s", op.name.c_str(), tot, bad, bad ? "" : "  (bit-identical)\n");
            if (bad) printf("  first at %zu: fast %g stream %g\n", first, fa, fb);
            bad_total += bad != 0;
        }
    }
    // ---- 2. time per launch inside a graph (chain over the weight sets)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_graph = [&](auto body, int launches) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        body();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; i++) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        return (double)ms * 1e3 / iters / launches;
    };
    printf("\n%-42s %10s %10s   %8s %8s\n", "in-graph us per launch (8 weight sets x 4)", "fast", "stream", "GB/s", "GB/s");
    for (auto &op : ops) {
        const int nsets = (int)op.W[0].size();
        if (nsets < NL) continue;
        MMVQArgs probe = make_args(op, 0, op.out_b);
        const bool ok_s = mmvq_stream_applicable(probe);
        const double ta = time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < nsets; l++) CK(launch_mmvq_fast(make_args(op, l, op.out_a), st)); }, 4 * nsets);
        const double tb = ok_s ? time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < nsets; l++) CK(launch_mmvq_stream(make_args(op, l, op.out_b), st)); }, 4 * nsets) : 0.0;
        printf("%-42s %10.2f %10.2f   %8.0f %8.0f\n", op.name.c_str(), ta, tb, op.bytes / ta * 1e-3, tb > 0 ? op.bytes / tb * 1e-3 : 0.0);
    }
    // read floors over the same buffers (cold: 8 sets)
    {
        unsigned *sink; CK(hipMalloc(&sink, 16));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&read_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        printf("\n%-42s %10s %10s %10s %10s %10s\n", "read floor, in-graph us per launch", "gs 2048x256", "gs 1024x256", "dma contig", "dma cu-win", "dma chip");
        for (int oi : {2, 0, 4, 3}) {
            Op &op = ops[oi];
            // one buffer per launch: segment 0 only for multi-segment ops, so compare by bytes of segment 0
            const size_t b0 = dev_row_bytes(op.type[0], op.K) * (size_t)op.N[0];
            double t[5];
            int vi = 0;
            for (int blocks : {2048, 1024})
                t[vi++] = time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < NL; l++) hipLaunchKernelGGL(read_gs_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4_t *)op.W[0][l], b0 / 16, sink); }, 4 * NL);
            for (int mode : {0, 1, 2})
                t[vi++] = time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < NL; l++) hipLaunchKernelGGL(read_dma_kernel, dim3(256), dim3(512), 131072, st, op.W[0][l], b0, mode); }, 4 * NL);
            printf("%-30s %8.1f MB  %10.2f %10.2f %10.2f %10.2f %10.2f   (best %.0f GB/s)\n", op.name.substr(0, 30).c_str(), b0 / 1e6, t[0], t[1], t[2], t[3], t[4],
                   b0 / std::min({t[0], t[1], t[2], t[3], t[4]}) * 1e-3);
        }
    }
    // the layer chain: qkv -> o -> gate/up -> down (q4k), 8 layers x 4
    {
        Op &qkv = ops[0], &o = ops[2], &gu = ops[3], &dn = ops[4];
        const size_t layer_bytes = qkv.bytes + o.bytes + gu.bytes + dn.bytes;
        const double ta = time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < NL; l++) {
            CK(launch_mmvq_fast(make_args(qkv, l, qkv.out_a), st)); CK(launch_mmvq_fast(make_args(o, l, o.out_a), st));
            CK(launch_mmvq_fast(make_args(gu, l, gu.out_a), st)); CK(launch_mmvq_fast(make_args(dn, l, dn.out_a), st)); } }, 4 * NL);
        const double tb = time_graph([&] { for (int r = 0; r < 4; r++) for (int l = 0; l < NL; l++) {
            CK(launch_mmvq_stream(make_args(qkv, l, qkv.out_b), st)); CK(launch_mmvq_stream(make_args(o, l, o.out_b), st));
            CK(launch_mmvq_stream(make_args(gu, l, gu.out_b), st)); CK(launch_mmvq_stream(make_args(dn, l, dn.out_b), st)); } }, 4 * NL);
        printf("%-42s %10.2f %10.2f   %8.0f %8.0f   (%.1f MB per layer, no attention)\n", "layer chain: qkv, o, gate/up, down", ta, tb, layer_bytes / ta * 1e-3, layer_bytes / tb * 1e-3, layer_bytes / 1e6);
    }
    // the same chain launched eagerly, with and without the barrier between launches (hipExtAnyOrderLaunch; a stream capture drops the flag).
    // Barrier-less, NOTHING orders a launch behind its predecessor's results: the time is an upper bound of what hiding a launch's ramp under
    // its predecessor's tail could gain (with the 64 KiB-ring build, exp_stream_r64, two workgroups fit a CU, so a successor can enter early)
    {
        Op &qkv = ops[0], &o = ops[2], &gu = ops[3], &dn = ops[4];
        auto time_eager = [&](bool anyorder) {
            mmvq_stream_set_anyorder_for_timing(anyorder);
            auto body = [&] { for (int r = 0; r < 4; r++) for (int l = 0; l < NL; l++) {
                CK(launch_mmvq_stream(make_args(qkv, l, qkv.out_b), st)); CK(launch_mmvq_stream(make_args(o, l, o.out_b), st));
                CK(launch_mmvq_stream(make_args(gu, l, gu.out_b), st)); CK(launch_mmvq_stream(make_args(dn, l, dn.out_b), st)); } };
            for (int i = 0; i < 3; i++) body();
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; i++) body();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            mmvq_stream_set_anyorder_for_timing(false);
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            return (double)ms * 1e3 / iters / (4 * NL);
        };
        const double t_bar = time_eager(false), t_any = time_eager(true);
        printf("%-42s %10.2f %10.2f   (us per layer of four launches, ring %d KiB; barrier-less: unordered, an upper bound)\n", "eager chain: with barrier / barrier-less", t_bar, t_any,
               MI355_ST_RING_KIB);
    }
#ifdef MI355_STREAM_PROBE
    {   // timeline of the stream kernels of one layer in the middle of a 3-layer chain (100 MHz wall clock)
        Op &qkv = ops[0], &o = ops[2], &gu = ops[3], &dn = ops[argc > 2 ? 6 : 4];   // (argv[2] given: ffn_down with ready-made codes)
        const int NK = 12;
        unsigned long long *probe; const int WPW = 2 + 8, WPK = 256 * WPW;   // waves per kernel (loaders + 8 consumers per workgroup)
        const size_t pn = (size_t)NK * WPK * 8;
        CK(hipMalloc(&probe, pn * 8)); CK(hipMemset(probe, 0, pn * 8));
        mmvq_stream_set_probe(probe);
        Op *chain[4] = {&qkv, &o, &gu, &dn};
        for (int rep = 0; rep < 2; rep++) {
            int seq = 0;
            for (int l = 0; l < 3; l++) for (int k = 0; k < 4; k++) { MMVQArgs a = make_args(*chain[k], l, chain[k]->out_b); a.nck = (seq++) << 2; CK(launch_mmvq_stream(a, st)); }
            CK(hipStreamSynchronize(st));
        }
        std::vector<unsigned long long> h(pn);
        CK(hipMemcpy(h.data(), probe, pn * 8, hipMemcpyDeviceToHost));
        mmvq_stream_set_probe(nullptr);
        unsigned long long base = ~0ull;
        for (int w = 0; w < WPK; w++) { const unsigned long long t = h[((size_t)4 * WPK + w) * 8]; if (t && t < base) base = t; }
        printf("\nstream kernel timeline, layer 2 of 3 (eager launches), us since the first wave of its qkv entered; min / median / max over waves\n");
        printf("(down = \"%s\", fuse mode %d)\n", dn.name.c_str(), dn.fuse);
        printf("%-8s %-18s %-18s %-18s %-18s %-18s | %-18s %-18s %-18s %-18s %-18s | %-16s %-16s\n", "kernel", "loader: top", "args loaded", "go", "all issued", "all landed",
               "consumers: top", "args loaded", "activation ready", "last decoded", "outputs stored", "sum wait-slot", "sum decode");
        for (int k = 4; k < 9; k++) {
            printf("%-8s", k == 4 ? "qkv" : k == 5 ? "o" : k == 6 ? "gate/up" : k == 7 ? "down" : "qkv(3)");
            for (int col = 0; col < 12; col++) {
                const bool loader = col < 5;
                static const int lmap[5] = {0, 7, 2, 1, 5}, cmap[7] = {0, 7, 2, 4, 5, 3, 6};
                const int sidx = loader ? lmap[col] : cmap[col - 5];
                const bool rel = col >= 10;
                std::vector<double> v;
                for (int w = 0; w < WPK; w++) {
                    if (loader != ((w % WPW) < 2)) continue;
                    const unsigned long long t = h[((size_t)k * WPK + w) * 8 + sidx];
                    if (t || rel) v.push_back(rel ? (double)t * 0.01 : (double)(long long)(t - base) * 0.01);
                }
                if (col == 5 || col == 10) printf(" |");
                if (v.empty()) { printf(" %-18s", "-"); continue; }
                std::sort(v.begin(), v.end());
                char buf[64]; snprintf(buf, sizeof buf, "%.2f/%.2f/%.2f", v.front(), v[v.size() / 2], v.back());
                printf(rel ? " %-16s" : " %-18s", buf);
            }
            printf("\n");
        }
    }
#endif
    printf(bad_total ? "\nMISMATCH in %d ops\n" : "\nall compared ops bit-identical\n", bad_total);
    return bad_total ? 2 : 0;
}
