// decode_mega.hip — every layer of a single-token decode step in ONE launch.
//
// Why: with one launch per mat-vec (mmvq_fast.hip) a Llama-3-8B step is 5 dependent launches per layer, and each pays
// a fixed ~4-5 us on top of its bytes: workgroup dispatch, the first activation round trip, the first weight round trip
// with an empty memory pipeline, and the tail where the last waves finish while HBM idles (profiles/r1_rocprof_*:
// attn_output moves 9.4 MB in 6.9 us, 1.4 TB/s, where the same loop streams 66 MB at 3.3 TB/s).  Weights do not
// depend on anything computed in the step, so nothing forces that pipeline to drain between mat-vecs except the launch
// boundary itself.
//
// How: a persistent grid, 2 workgroups of 256 threads per CU (co-resident by construction: LDS is sized so that exactly
// two fit), walks the layers.  A phase is one of the step's five operations
//     Q/K/V (RMSNorm + Q8_K quantise in the prologue) | attention (rope + cache store + chunk attention + merge + quantise)
//     | attn_output (+ residual) | gate/up (RMSNorm prologue, SwiGLU epilogue) | down (quantise prologue, + residual)
// and phases are separated by a device-wide barrier (two-level atomic counters, no cache-wide fences: the few KB the
// phases hand to each other are read and written device-coherently instead).  Each wave ARRIVES at the barrier as soon as its outputs are stored, then requests the first 4 units of
// the next phase's weights (the register ring of run_fast: 32 MB across the chip, i.e. all of attn_output or Q/K/V and
// half of gate/up), and only then waits for the others — so the barrier latency, the straggler tail of the previous phase
// and the activation round trip are all spent with weight requests in flight.
//
// The arithmetic is run_fast / flash_attn_decode_item unchanged: a mega step produces bit-identical results to the
// per-launch path (tests/test_gpu_model.py::test_mega_step_matches_per_launch_bitwise).
//
// Safety: a spinning barrier needs every workgroup resident.  If that were ever not the case the wait gives up after
// a bounded number of polls, raises the sticky flag sync[1] (the host checks it and reports an error instead of
// returning tokens) and every later barrier falls through, so the launch always terminates.
#include "mmvq_fast_dev.h"
#include "attn_decode_dev.h"

namespace mi355 {

namespace {

constexpr int MEGA_NT = 256, MEGA_NW = 4;
constexpr int MEGA_SPIN_LIMIT = 400000;     // polls of ~1 us: a healthy barrier takes a handful

// Device-wide barrier.  No cache-wide fences: everything the phases exchange is read and written device-coherently (COH
// accesses in mmvq_fast_dev.h / attn_decode_dev.h), so the barrier only has to order "my stores have completed" (vmcnt(0)
// before arriving) against "everybody has arrived".  Two-level arrival: the workgroup bumps one of 8 group counters
// (blockIdx % 8: one 128-byte line each, so the 512 same-address atomics of a flat counter, 8 us, become 8 x 64 in
// parallel), the last of a group bumps the top counter, everybody polls the top counter: 2.7 us (tools/bench_gridbar.hip).
// words: [0] top | [1] sticky time-out flag | [32 * (1 + g)] group counter g
struct GridBarrier {
    unsigned *words;
    unsigned n_wg;
    unsigned epoch;                          // barriers this workgroup has arrived at
    unsigned long long *probe;               // diagnosis: workgroup 0 stamps the 100 MHz wall clock when it arrives and when its wait ends
    int n_stamp;
    __device__ __forceinline__ void stamp() { if (probe && threadIdx.x == 0) probe[n_stamp++] = wall_clock64(); }

    __device__ __forceinline__ int *timeout_flag() const { return reinterpret_cast<int *>(words + 1); }
    // every wave, once its outputs of the phase are issued: wait until they have completed, then ONE arrival per workgroup
    __device__ __forceinline__ void arrive() {
        stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // __syncthreads() only orders LDS (see attn_decode_dev.h)
        __syncthreads();
        epoch++;
        if (threadIdx.x == 0)                                   // no return value: nothing to wait for
            (void)__hip_atomic_fetch_add(words + 32 * (1 + (blockIdx.x & 7u)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the 8 group counters through the SCALAR memory path (the words live in uncached device memory, glc skips the scalar
    // cache): vector memory returns in order, so a vector poll would only come back after this wave's weight prefetch
    __device__ __forceinline__ bool all_arrived() const {
        unsigned c0, c1, c2, c3, c4, c5, c6, c7;
        asm volatile("s_load_dword %0, %8, 0x80 glc\n\ts_load_dword %1, %8, 0x100 glc\n\ts_load_dword %2, %8, 0x180 glc\n\t"
                     "s_load_dword %3, %8, 0x200 glc\n\ts_load_dword %4, %8, 0x280 glc\n\ts_load_dword %5, %8, 0x300 glc\n\t"
                     "s_load_dword %6, %8, 0x380 glc\n\ts_load_dword %7, %8, 0x400 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3), "=&s"(c4), "=&s"(c5), "=&s"(c6), "=&s"(c7)
                     : "s"(words) : "memory");
        const unsigned base = n_wg >> 3, rem = n_wg & 7u;      // group g has base + (g < rem) workgroups
        const unsigned e = epoch;
        return c0 >= e * (base + (0u < rem)) && c1 >= e * (base + (1u < rem)) && c2 >= e * (base + (2u < rem)) && c3 >= e * (base + (3u < rem)) &&
               c4 >= e * (base + (4u < rem)) && c5 >= e * (base + (5u < rem)) && c6 >= e * (base + (6u < rem)) && c7 >= e * (base + (7u < rem));
    }
    __device__ __forceinline__ void wait() {
        if (threadIdx.x == 0) {
            if (__hip_atomic_load(timeout_flag(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                int polls = 0;
                while (!all_arrived()) {
                    if (++polls > MEGA_SPIN_LIMIT) {
                        __hip_atomic_store(timeout_flag(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            stamp();
        }
        __syncthreads();
    }
};

struct PhaseSync {                           // what run_fast<.., PRE = true> calls between its weight and activation requests
    GridBarrier *gb;
    bool skip;                               // first phase of the launch: its inputs were complete before the kernel started
    __device__ __forceinline__ void operator()() const { if (!skip) gb->wait(); else gb->stamp(); }
};

template <int KB, int FUSE>
__device__ __forceinline__ void mega_mmvq(const MMVQArgs &a, uint8_t *smem, const PhaseSync &sync) {
    const int b = (int)blockIdx.x;
    int s = 0;
    if (a.n_seg > 1 && b >= a.seg_block0[1]) s = 1;
    if (a.n_seg > 2 && b >= a.seg_block0[2]) s = 2;
    if (b >= a.seg_block0[3]) { sync(); return; }                // more workgroups than row pairs: nothing to do in this phase
    const int nblk = a.seg_block0[s + 1] - a.seg_block0[s];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = (b - a.seg_block0[s]) * MEGA_NW + wave;
    const int nw = nblk * MEGA_NW;
    switch (a.seg[s].type) {
        case T_Q4_K: run_fast<T_Q4_K, KB, MEGA_NT, FUSE, true, PhaseSync>(a, a.seg[s], smem, gw, nw, sync); break;
        case T_Q5_K: run_fast<T_Q5_K, KB, MEGA_NT, FUSE, true, PhaseSync>(a, a.seg[s], smem, gw, nw, sync); break;
        case T_Q6_K: run_fast<T_Q6_K, KB, MEGA_NT, FUSE, true, PhaseSync>(a, a.seg[s], smem, gw, nw, sync); break;
        default: sync(); break;
    }
}

// its own function (not inlined): the attention item and the mat-vec loops each want most of the register file, and
// inlined side by side the allocator kept mat-vec lane constants alive across the attention phase (scratch spills)
template <int R, int TK, int TV>
__device__ __forceinline__ void mega_attention(const AttnArgs &aa, const float *cs_table, int n_rot, const DecodeFuse &fz, int items) {
    for (int it = (int)blockIdx.x; it < items; it += (int)gridDim.x) {
        flash_attn_decode_item<R, TK, TV, true, true>(aa, cs_table, n_rot, fz, it % aa.G, it / aa.G, 0);
        __syncthreads();                                         // the item's LDS is reused by the next one / the next phase
    }
}

template <int KBE, int KBF, int R, int TK, int TV>
__global__ __launch_bounds__(MEGA_NT, 2) void decode_mega_kernel(const MegaLayer *__restrict__ layers, int n_layer, AttnArgs aa, const float *cs_table,
                                                              int n_rot, const DecodeFuse fz, unsigned *sync_words, int *host_flag, unsigned long long *probe) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GridBarrier gb{sync_words, gridDim.x, 0u, nullptr, 0};
    gb.probe = blockIdx.x == 0 ? probe : nullptr;   // diagnosis (MI355_MEGA_PROBE=1)
    gb.stamp();
    for (int il = 0; il < n_layer; il++) {
        const MegaLayer &L = layers[il];
        // ---- Q, K, V (previous layer's down projection + residual must be complete)
        mega_mmvq<KBE, 1>(L.qkv, smem, PhaseSync{&gb, il == 0});
        gb.arrive();
        gb.wait();
        // ---- attention: items (kv head, chunk slot) over the grid
        aa.kv = L.kv;
        const int items = aa.G * aa.splits;
        mega_attention<R, TK, TV>(aa, cs_table, n_rot, fz, items);
        gb.arrive();
        // ---- attn_output + residual
        mega_mmvq<KBE, 0>(L.wo, smem, PhaseSync{&gb, false});
        gb.arrive();
        // ---- gate / up with SwiGLU
        mega_mmvq<KBE, 1>(L.gate_up, smem, PhaseSync{&gb, false});
        gb.arrive();
        // ---- down + residual
        mega_mmvq<KBF, 2>(L.down, smem, PhaseSync{&gb, false});
        gb.arrive();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *host_flag = __hip_atomic_load(gb.timeout_flag(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

bool experiments_built() { return true; }      // (this file and decode_engine.hip are compiled together or not at all: build.py)
int mega_blocks() { return 2 * num_cu(); }

bool decode_mega_applicable(int kb_e, int kb_ff, int R, int type_k, int type_v) {
    return kb_e == 2 && kb_ff == 7 && R == 4 && (type_k == T_Q8_0 || type_k == T_F16) && type_k == type_v;
}

hipError_t launch_decode_mega(const MegaLayer *layers_dev, int n_layer, int kb_e, int kb_ff, const AttnArgs &a, const float *cs_table,
                              int n_rot, const float *knew, const float *vnew, const int32_t *tok_cell, unsigned *counters,
                              unsigned *sync, int *host_flag, unsigned long long *probe, size_t lds_mmvq, hipStream_t st) {
    const int R = a.H / a.G;
    if (!decode_mega_applicable(kb_e, kb_ff, R, a.type_k, a.type_v) || a.T != 1 || !counters) return hipErrorInvalidValue;
    DecodeFuse fz{};
    fz.knew = knew; fz.vnew = vnew; fz.tok_cell = tok_cell; fz.counters = counters;
    if (a.out_q) fz.q = *a.out_q;
    fz.want_q8k = (int)(a.out_q && a.out_q8k); fz.want_q80 = (int)(a.out_q && a.out_q80);
    fz.probe = nullptr;
    // exactly two workgroups per CU: with more than a third of the CU's 160 KB of LDS each, a third cannot be placed, and
    // 2 x 256 CUs = the whole grid is resident.  (The attention item's static LDS comes on top of the dynamic part.)
    size_t lds = lds_mmvq < 36 * 1024 ? 36 * 1024 : lds_mmvq;
    const dim3 grid(mega_blocks());
    if (a.type_k == T_Q8_0) hipLaunchKernelGGL((decode_mega_kernel<2, 7, 4, T_Q8_0, T_Q8_0>), grid, dim3(MEGA_NT), lds, st, layers_dev, n_layer, a, cs_table, n_rot, fz, sync, host_flag, probe);
    else hipLaunchKernelGGL((decode_mega_kernel<2, 7, 4, T_F16, T_F16>), grid, dim3(MEGA_NT), lds, st, layers_dev, n_layer, a, cs_table, n_rot, fz, sync, host_flag, probe);
    return hipGetLastError();
}

}  // namespace mi355
