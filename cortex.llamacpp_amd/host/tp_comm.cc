// tp_comm.cc — RCCL (over xGMI) transport of the row-split exchange step, plus the host-callback transport used to
// validate the sharding where every rank cannot own a GPU.  See tp_comm.h.
#include "tp_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>     // types and enums only: the functions are looked up in the library at run time

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <vector>

namespace mi355 {

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
};

struct Group {
    int rank = 0, size = 1;
    ncclComm_t comm = nullptr;
    tp_host_exchange_fn host_fn = nullptr;
    void *host_user = nullptr;
    float *pinned = nullptr;          // host transport staging
    size_t pinned_floats = 0;
    bool null_group = false;          // tp_set_null_group
};

RcclApi g_api;
Group g_grp;
std::mutex g_mu;

// ---- peer-to-peer exchange state (see tp_comm.h)
constexpr int P2P_MAX_RANKS = 8, P2P_WGS = 16;
// A wait for a PEER is bounded in time, not in polls (the 100 MHz wall clock every process on the device shares): ranks reach an exchange together because one
// driver steps them in lock-step, so seconds of skew mean a dead peer - but a poll count meant 1.4 s or 4 s depending on what else used the memory system
constexpr unsigned long long P2P_WAIT_TICKS = 10ull * 100000000ull;      // 10 s
constexpr int RSAG_MAX_WGS = 256;      // flag rows are laid out for this many slices; the launch uses g_p2p.rsag_wgs of them
struct P2PDev {                        // by value into the kernel
    float *data[P2P_MAX_RANKS];        // base of every rank's slot area: [2 sets][P slots][max_floats]
    unsigned *flags[P2P_MAX_RANKS];    // base of every rank's flag area: [2 sets][P][P2P_WGS]
    unsigned *epoch;                   // this rank's private counters [P2P_WGS]
    unsigned *err;                     // nullable: pinned host word
    int rank, size;
    size_t max_floats;
    // prompt-sized messages (p2p_rsag_kernel): per rank [2 sets][P sources][seg_max] partial segments, then [2 sets][P owners][seg_max] reduced segments
    float *big[P2P_MAX_RANKS];
    unsigned *bflags[P2P_MAX_RANKS];   // [2 sets][2 phases][P][RSAG_MAX_WGS]
    unsigned *bepoch;                  // private [RSAG_MAX_WGS]
    size_t seg_max;                    // floats per segment slot (a multiple of 4)
    // "yield" form (ranks that SHARE a device - the host-callback rigs - or MI355_TP_YIELD=1): a wait for a peer gives up after yield_polls polls, the workgroup
    // notes how far it got (state[w]: 0 fresh, 1 = its part is out and signalled, 2 = reduce-scatter: its reduced segment is out and signalled), counts itself in
    // *pending and ENDS; the host re-enqueues the kernel until nothing is pending.  A kernel that spins for a peer holds the device, and whether another
    // process's kernel gets in beside it is the hardware scheduler's decision (round 6: eight processes behind one MI355X went eight for ten into the wait's
    // bound, profiles/r6_tp_shared_device_trace.txt) - a launch that ends lets everybody run.  yield_polls == 0: one launch, waits bounded in time (production).
    unsigned *state;                   // [P2P_WGS + RSAG_MAX_WGS] resume states (this rank's private memory)
    unsigned *pending;                 // pinned host word: workgroups of the current launch that left unfinished
    int yield_polls;
    unsigned target_e;                 // yield form: the serial of THIS exchange (= exchanges of its kind so far); a workgroup whose epoch has reached it is done and must not start the next one when the launch is repeated for the others
    unsigned long long *trace;         // nullable (MI355_TP_TRACE=1): per launch of either kernel 4 stamps of the 100 MHz wall clock, written by workgroup 0
    unsigned trace_slot;               // ... at trace[4 * trace_slot ..]: entered, own flags out, first wait over, done
};
struct P2PState {
    bool on = false;
    uint8_t *local = nullptr;
    size_t bytes = 0, flags_off = 0;
    void *peer[P2P_MAX_RANKS] = {};
    unsigned *epoch = nullptr, *err = nullptr;
    size_t max_floats = 0;
    int64_t exchanges = 0;
    size_t big_off = 0, bflags_off = 0, prompt_floats = 0;
    unsigned *bepoch = nullptr;
    unsigned *state = nullptr, *pending = nullptr;      // the yield form's resume states (device) and its pending word (pinned host)
    bool prompt_on = false;
    int rsag_wgs = 64;
    int64_t prompt_exchanges = 0;
    P2PDev dev{};
};
P2PState g_p2p;

// a wait for the P flags `f(q)` of this workgroup to carry e: true when they all do.  yield_polls > 0: gives up after that many polls (the caller ends the
// launch); else bounded in time, and the error word is raised when the bound runs out (the caller goes on: what it computes is discarded with the step)
template <class F>
__device__ __forceinline__ bool p2p_wait_flags(const P2PDev &a, unsigned e, F flag_of) {
    __shared__ int gave_up;
    const int tid = (int)threadIdx.x, P = a.size;
    if (tid == 0) gave_up = 0;
    __syncthreads();
    if (tid < P) {
        const unsigned *f = flag_of(tid);
        const unsigned long long t0 = wall_clock64();
        int spins = 0;
        while (f && __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            ++spins;
            if (a.yield_polls > 0) { if (spins >= a.yield_polls) { gave_up = 1; break; } }
            else if ((spins & 255) == 0 && wall_clock64() - t0 > P2P_WAIT_TICKS) { if (a.err) __hip_atomic_fetch_or(a.err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    return gave_up == 0;
}
// one workgroup = one slice of the message, start to finish: no device-wide step inside the kernel
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(const float *send, float *recv, int n, const P2PDev a) {
    __shared__ unsigned e_sh, st_sh;
    const int tid = (int)threadIdx.x, w = (int)blockIdx.x, P = a.size;
    if (tid == 0) { e_sh = a.epoch[w] + 1u; st_sh = a.yield_polls > 0 ? a.state[w] : 0u; }
    if (a.trace && w == 0 && tid == 0 && a.trace[4 * a.trace_slot] == 0) a.trace[4 * a.trace_slot] = wall_clock64();
    __syncthreads();
    const unsigned e = e_sh, stt = st_sh;
    if (a.yield_polls > 0 && e > a.target_e) return;          // (a repeated launch: this workgroup finished the exchange in an earlier one)
    const size_t set = e & 1u;
    const int per = (((n + 3) / 4 + (int)gridDim.x - 1) / (int)gridDim.x) * 4;
    const int lo = w * per, hi = lo + per < n ? lo + per : n;
    if (stt == 0) {
        // 1. this rank's slice into slot `rank` of every rank's buffer (peer stores over xGMI; the own copy too), then one flag per peer
        const size_t my_slot = (set * (size_t)P + (size_t)a.rank) * a.max_floats;
        for (int i = lo + tid; i < hi; i += 256) {
            const float v = send[i];
            for (int q = 0; q < P; q++) __hip_atomic_store(a.data[q] + my_slot + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __threadfence_system();                               // every thread: its stores have left for their owners before the flags do
        __syncthreads();
        if (tid < P) __hip_atomic_store(a.flags[tid] + (set * (size_t)P + (size_t)a.rank) * P2P_WGS + w, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.yield_polls > 0 && tid == 0) a.state[w] = 1u;
    }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 1] = wall_clock64();
    // 2. everybody's slice w has arrived in MY buffer once my P flags of this set carry e
    if (!p2p_wait_flags(a, e, [&](int q) { return a.flags[a.rank] + (set * (size_t)P + (size_t)q) * P2P_WGS + w; })) {
        if (tid == 0) __hip_atomic_fetch_add(a.pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;                                               // (the epoch stays: the next launch resumes this exchange)
    }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 2] = wall_clock64();
    // 3. the P slots added in rank order (system-scope loads: the lines were written by other devices)
    const float *mine = a.data[a.rank] + set * (size_t)P * a.max_floats;
    for (int i = lo + tid; i < hi; i += 256) {
        float acc = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int q = 1; q < P; q++) acc += __hip_atomic_load(mine + (size_t)q * a.max_floats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        recv[i] = acc;
    }
    if (tid == 0) { a.epoch[w] = e; if (a.yield_polls > 0) a.state[w] = 0u; }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 3] = wall_clock64();
}

// ---- prompt-sized messages: reduce-scatter + all-gather on all links at once (no reference counterpart: upstream copies whole
// activations between peers).  The message is cut into P segments, rank r owns segment r.  Phase 1: every rank stores its part of segment q
// straight into rank q's buffer (slot = source rank) - P - 1 peer streams leaving on P - 1 different links, (P - 1) / P of the message in total.
// Phase 2: the owner adds the P parts of its segment in rank order (the same association as the one-shot kernel above, and p0 + p1 for two ranks)
// and stores the sum into every rank's buffer; each rank copies the P - 1 foreign segments from its own buffer into `recv`.  A ring all-reduce
// moves 2 (P - 1) / P of the message over ONE link direction per rank; this moves the same bytes over seven.
// A workgroup owns the same sub-slice of every segment from start to finish and waits only for the workgroups of the same index on the other
// ranks: no device-wide step, no co-residency requirement.  Two buffer sets alternate as above.  Remote data is stored write-through (sc0 sc1) and
// read back past the L2 (the lines were written by other devices); flags are system-scope atomics behind a system fence.
typedef unsigned int tp_u32x4 __attribute__((ext_vector_type(4)));
typedef float tp_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tp_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7ffffff0, 0x00020000);
}
constexpr int TP_SYS = 17;             // cache policy bits sc0 | sc1: system scope
__device__ __forceinline__ tp_f32x4 tp_ld_sys(__amdgpu_buffer_rsrc_t r, int float_off) {
    return __builtin_bit_cast(tp_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, TP_SYS));
}
__device__ __forceinline__ void tp_st_sys(__amdgpu_buffer_rsrc_t r, int float_off, tp_f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tp_u32x4, v), r, float_off * 4, 0, TP_SYS);
}
__device__ __forceinline__ void rsag_signal(const P2PDev &a, size_t set, int phase, int w, unsigned e) {
    const int tid = (int)threadIdx.x, P = a.size;
    __threadfence_system();                               // every thread: its stores have reached their owners before the flags leave
    __syncthreads();
    const size_t row = (set * 2 + (size_t)phase) * (size_t)P;
    if (tid < P && tid != a.rank) __hip_atomic_store(a.bflags[tid] + (row + (size_t)a.rank) * RSAG_MAX_WGS + w, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool rsag_wait(const P2PDev &a, size_t set, int phase, int w, unsigned e) {
    const size_t row = (set * 2 + (size_t)phase) * (size_t)a.size;
    return p2p_wait_flags(a, e, [&](int q) -> const unsigned * { return q == a.rank ? nullptr : a.bflags[a.rank] + (row + (size_t)q) * RSAG_MAX_WGS + w; });
}
__global__ __launch_bounds__(256) void p2p_rsag_kernel(const float *send, float *recv, int n, const P2PDev a) {
    __shared__ unsigned e_sh, st_sh;
    const int tid = (int)threadIdx.x, w = (int)blockIdx.x, G = (int)gridDim.x, P = a.size, R = a.rank;
    unsigned *const st_w = a.state + P2P_WGS + w;
    if (tid == 0) { e_sh = a.bepoch[w] + 1u; st_sh = a.yield_polls > 0 ? *st_w : 0u; }
    if (a.trace && w == 0 && tid == 0 && a.trace[4 * a.trace_slot] == 0) a.trace[4 * a.trace_slot] = wall_clock64();
    __syncthreads();
    const unsigned e = e_sh, stt = st_sh;
    if (a.yield_polls > 0 && e > a.target_e) return;          // (a repeated launch: this workgroup finished the exchange in an earlier one)
    auto leave = [&]() { if (tid == 0) __hip_atomic_fetch_add(a.pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); };
    const size_t set = e & 1u;
    const int seg = ((n / 4 + P - 1) / P) * 4;                       // floats per segment (the last one may be shorter, or empty)
    const int per = ((seg / 4 + G - 1) / G) * 4;                     // floats of a segment that one workgroup carries
    const int s_lo = w * per, s_hi = s_lo + per < seg ? s_lo + per : seg;
    const size_t slot = a.seg_max, half = 2 * (size_t)P * slot;      // [partials: 2 sets][reduced: 2 sets]
    // 1. my part of every foreign segment into its owner's buffer, slot R; the nearest owner first so that the P ranks start on P different links
    if (stt == 0) {
        for (int d = 1; d < P; d++) {
            const int q = R + d < P ? R + d : R + d - P;
            const int hi = q * seg + s_hi < n ? s_hi : n - q * seg;
            const __amdgpu_buffer_rsrc_t dst = tp_rsrc(a.big[q] + (set * (size_t)P + (size_t)R) * slot);
            const tp_f32x4 *src = reinterpret_cast<const tp_f32x4 *>(send + (size_t)q * seg);
            for (int i = s_lo + tid * 4; i < hi; i += 1024) tp_st_sys(dst, i, src[i >> 2]);
        }
        rsag_signal(a, set, 0, w, e);
        if (a.yield_polls > 0 && tid == 0) *st_w = 1u;
    }
    if (stt <= 1) {
    if (!rsag_wait(a, set, 0, w, e)) { leave(); return; }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 1] = wall_clock64();
    // 2. my segment: the P parts in rank order; the sum to `recv` and into every rank's reduced area, slot R
    {
        const int hi = R * seg + s_hi < n ? s_hi : n - R * seg;
        const tp_f32x4 *own = reinterpret_cast<const tp_f32x4 *>(send + (size_t)R * seg);
        tp_f32x4 *out = reinterpret_cast<tp_f32x4 *>(recv + (size_t)R * seg);
        const float *parts = a.big[R] + set * (size_t)P * slot;
        for (int i = s_lo + tid * 4; i < hi; i += 1024) {
            tp_f32x4 acc = R == 0 ? own[i >> 2] : tp_ld_sys(tp_rsrc(parts), i);
            for (int q = 1; q < P; q++) acc += q == R ? own[i >> 2] : tp_ld_sys(tp_rsrc(parts + (size_t)q * slot), i);
            out[i >> 2] = acc;
            for (int d = 1; d < P; d++) {
                const int q = R + d < P ? R + d : R + d - P;
                tp_st_sys(tp_rsrc(a.big[q] + half + (set * (size_t)P + (size_t)R) * slot), i, acc);
            }
        }
    }
    rsag_signal(a, set, 1, w, e);
    if (a.yield_polls > 0 && tid == 0) *st_w = 2u;
    }
    if (!rsag_wait(a, set, 1, w, e)) { leave(); return; }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 2] = wall_clock64();
    // 3. the foreign segments out of my own buffer
    for (int d = 1; d < P; d++) {
        const int q = R + d < P ? R + d : R + d - P;
        const int hi = q * seg + s_hi < n ? s_hi : n - q * seg;
        const __amdgpu_buffer_rsrc_t src = tp_rsrc(a.big[R] + half + (set * (size_t)P + (size_t)q) * slot);
        tp_f32x4 *out = reinterpret_cast<tp_f32x4 *>(recv + (size_t)q * seg);
        for (int i = s_lo + tid * 4; i < hi; i += 1024) out[i >> 2] = tp_ld_sys(src, i);
    }
    if (tid == 0) { a.bepoch[w] = e; if (a.yield_polls > 0) *st_w = 0u; }
    if (a.trace && w == 0 && tid == 0) a.trace[4 * a.trace_slot + 3] = wall_clock64();
}

// ---- MI355_TP_TRACE=1: when each exchange was launched (host clock) and when its kernel entered / signalled / stopped waiting / ended (the device's 100 MHz
// wall clock, one counter for every process on the GPU), printed per rank at exit - who is late for whom, and whether the host or the device held it back
constexpr int TRACE_MAX = 4096;
struct TraceRec { double host_us; int n; int kind; };
std::vector<TraceRec> g_trace;
unsigned long long *g_trace_dev = nullptr;       // pinned host memory, device-visible
bool g_trace_on = false, g_trace_init = false;
double now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3; }
void trace_dump() {
    if (!g_trace_dev || g_trace.empty()) return;
    (void)hipDeviceSynchronize();
    const double h0 = g_trace[0].host_us;
    fprintf(stderr, "[tp trace] rank %d of %d: %zu exchanges (host us since the first launch | device wall clock us: entered, signalled, waited, done)\n", g_grp.rank, g_grp.size, g_trace.size());
    for (size_t i = 0; i < g_trace.size() && i < (size_t)TRACE_MAX; i++) {
        const unsigned long long *t = g_trace_dev + 4 * i;
        const bool slow = (t[3] - t[0]) > 100ull * 1000 || (i > 0 && g_trace[i].host_us - g_trace[i - 1].host_us > 50e3);   // > 1 ms in the kernel, or > 50 ms between two launches
        if (i < 8 || slow)
            fprintf(stderr, "[tp trace] rank %d #%zu %s n=%d host %.0f | dev %.2f %+.2f %+.2f %+.2f%s\n", g_grp.rank, i, g_trace[i].kind ? "rsag" : "p2p", g_trace[i].n, g_trace[i].host_us - h0,
                    (double)t[0] / 100.0, (double)(t[1] - t[0]) / 100.0, (double)(t[2] - t[0]) / 100.0, (double)(t[3] - t[0]) / 100.0, slow ? "  <-- slow" : "");
    }
}
unsigned trace_next(int n, int kind) {
    if (!g_trace_init) {
        g_trace_init = true;
        const char *ev = getenv("MI355_TP_TRACE");
        if (ev && ev[0] == '1' && hipHostMalloc((void **)&g_trace_dev, (size_t)TRACE_MAX * 4 * 8, hipHostMallocDefault) == hipSuccess) {
            memset(g_trace_dev, 0, (size_t)TRACE_MAX * 4 * 8);
            g_trace_on = true;
            atexit(trace_dump);
        }
    }
    if (!g_trace_on || g_trace.size() >= (size_t)TRACE_MAX) return 0;
    g_trace.push_back({now_us(), n, kind});
    return (unsigned)g_trace.size() - 1;
}

bool load_rccl(std::string &err) {
    if (g_api.lib) return true;
    // the soname first: a process that already holds an RCCL (e.g. through torch) gets that copy back
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) { err = std::string("cannot open librccl: ") + dlerror(); return false; }
    RcclApi a;
    a.lib = h;
#define SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name)); if (!a.field) { err = std::string("librccl lacks ") + name; dlclose(h); return false; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(h, "ncclCommAbort"));
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
#undef SYM
    g_api = a;
    return true;
}

hipError_t host_exchange(float *dev_send, float *dev_recv, size_t n_total, size_t n_arg, size_t my_off, size_t n_mine, int op, hipStream_t st) {
    Group &g = g_grp;
    if (g.pinned_floats < n_total) {
        if (g.pinned) (void)hipHostFree(g.pinned);
        g.pinned = nullptr; g.pinned_floats = 0;
        if (hipHostMalloc((void **)&g.pinned, n_total * sizeof(float), hipHostMallocDefault) != hipSuccess) return hipErrorOutOfMemory;
        g.pinned_floats = n_total;
    }
    hipError_t e = hipMemcpyAsync(g.pinned + my_off, dev_send, n_mine * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (g.host_fn(g.host_user, g.pinned, n_arg, op) != 0) return hipErrorUnknown;
    e = hipMemcpyAsync(dev_recv, g.pinned, n_total * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    return e;
}

// an exchange that ran out of time: which peers' flags never arrived here (this rank's own flag area and counters, read back), on stderr
void p2p_report_missing(bool rsag, int attempts) {
    const int P = g_grp.size, R = g_grp.rank;
    const int n_wg = rsag ? g_p2p.rsag_wgs : P2P_WGS, stride = rsag ? RSAG_MAX_WGS : P2P_WGS;
    std::vector<unsigned> ep((size_t)(P2P_WGS + RSAG_MAX_WGS) * 2), fl;
    if (hipMemcpy(ep.data(), g_p2p.epoch, ep.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
    const unsigned *epoch = ep.data() + (rsag ? P2P_WGS : 0), *state = ep.data() + (P2P_WGS + RSAG_MAX_WGS) + (rsag ? P2P_WGS : 0);
    const size_t fl_words = rsag ? (size_t)4 * P * RSAG_MAX_WGS : (size_t)2 * P * P2P_WGS;
    fl.resize(fl_words);
    if (hipMemcpy(fl.data(), g_p2p.local + (rsag ? g_p2p.bflags_off : g_p2p.flags_off), fl_words * 4, hipMemcpyDeviceToHost) != hipSuccess) return;
    std::string line = "[tp] rank " + std::to_string(R) + ": " + (rsag ? "reduce-scatter" : "all-reduce") + " exchange abandoned after " + std::to_string(attempts + 1) + " launches;";
    for (int w = 0; w < n_wg && w < 4; w++) {
        const unsigned e = epoch[w] + 1u, set = e & 1u;
        line += " wg" + std::to_string(w) + " epoch " + std::to_string(e) + " state " + std::to_string(state[w]) + " flags";
        const int phases = rsag ? 2 : 1;
        for (int ph = 0; ph < phases; ph++) {
            line += ph ? " | " : " ";
            for (int q = 0; q < P; q++) {
                const size_t row = rsag ? ((size_t)set * 2 + ph) * P + q : (size_t)set * P + q;
                line += (q == R && rsag ? std::string("-") : std::to_string(fl[row * stride + w])) + (q + 1 < P ? "," : "");
            }
        }
        line += ";";
    }
    fprintf(stderr, "%s\n", line.c_str());
}

// the yield form's launch: the kernel again and again until no workgroup left unfinished.  Every launch ends within a few milliseconds whatever the peers do, so
// the processes that share the device all get to run; the exchange as a whole is bounded in time (MI355_TP_YIELD_TIMEOUT_S, default 60) and raises the error
// word when the bound runs out.  Only where the stream may be drained inside a step (the host-callback transport does that for its own exchanges anyway).
hipError_t p2p_launch_until_done(bool rsag, const float *send, float *recv, int n, hipStream_t st) {
    static const double limit_s = [] { const char *e = getenv("MI355_TP_YIELD_TIMEOUT_S"); const double v = e ? atof(e) : 60.0; return v > 0.0 ? v : 60.0; }();
    const double t0 = now_us();
    g_p2p.dev.target_e = (unsigned)(rsag ? g_p2p.prompt_exchanges : g_p2p.exchanges);      // (counted by the caller before it came here)
    for (int attempt = 0;; attempt++) {
        *g_p2p.pending = 0;
        if (rsag) hipLaunchKernelGGL(p2p_rsag_kernel, dim3(g_p2p.rsag_wgs), dim3(256), 0, st, send, recv, n, g_p2p.dev);
        else hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_WGS), dim3(256), 0, st, send, recv, n, g_p2p.dev);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return e;
        if (*reinterpret_cast<volatile unsigned *>(g_p2p.pending) == 0) return hipSuccess;
        if ((now_us() - t0) * 1e-6 > limit_s) {
            if (g_p2p.err) __atomic_fetch_or(g_p2p.err, 32u, __ATOMIC_RELAXED);
            p2p_report_missing(rsag, attempt);
            (void)hipMemsetAsync(g_p2p.state, 0, (P2P_WGS + RSAG_MAX_WGS) * sizeof(unsigned), st);      // (the step is lost; what follows must not resume it)
            return hipSuccess;
        }
        struct timespec ts = {0, attempt < 20 ? 50000 : 500000};
        nanosleep(&ts, nullptr);
    }
}

}  // namespace

int tp_unique_id(void *out, size_t cap, std::string &err) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || cap < sizeof(ncclUniqueId)) { err = "unique id buffer must hold 128 bytes"; return -1; }
    if (!load_rccl(err)) return -1;
    ncclUniqueId id;
    const ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + g_api.GetErrorString(r); return -1; }
    std::memcpy(out, &id, sizeof id);
    return (int)sizeof id;
}

int tp_init(int rank, int size, const void *id, size_t id_len, std::string &err) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (size < 1 || rank < 0 || rank >= size) { err = "bad rank / size"; return -1; }
    if (g_grp.comm || g_grp.host_fn) { err = "a row-split group is already active in this process"; return -1; }
    if (!id || id_len < sizeof(ncclUniqueId)) { err = "unique id must be 128 bytes"; return -1; }
    if (!load_rccl(err)) return -1;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_api.CommInitRank(&comm, size, uid, rank);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + g_api.GetErrorString(r); return -1; }
    g_grp.comm = comm; g_grp.rank = rank; g_grp.size = size;
    return 0;
}

void tp_set_host_exchange(tp_host_exchange_fn fn, void *user, int rank, int size) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_grp.host_fn = fn; g_grp.host_user = user;
    if (fn) { g_grp.rank = rank; g_grp.size = size; }
    else if (!g_grp.comm) { g_grp.rank = 0; g_grp.size = 1; }
}

void tp_set_null_group(int rank, int size) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_grp.comm || g_grp.host_fn) return;
    g_grp.null_group = size >= 1; g_grp.rank = rank; g_grp.size = size >= 1 ? size : 1;
}

int tp_p2p_local_handle(void *out, size_t cap, size_t max_floats, size_t prompt_floats, std::string &err) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || cap < (size_t)TP_P2P_HANDLE_BYTES) { err = "handle buffer must hold 64 bytes"; return -1; }
    if (!(g_grp.comm || g_grp.host_fn) || g_grp.size < 2 || g_grp.size > P2P_MAX_RANKS) { err = "peer-to-peer exchange needs a row-split group of 2..8 ranks first"; return -1; }
    if (max_floats == 0 || (max_floats & 3)) { err = "max_floats must be a positive multiple of 4"; return -1; }
    static_assert(sizeof(hipIpcMemHandle_t) == TP_P2P_HANDLE_BYTES, "IPC handle size");
    P2PState &p = g_p2p;
    if (!p.local) {
        const size_t P = (size_t)g_grp.size;
        if (prompt_floats & 3) { err = "prompt_floats must be a multiple of 4"; return -1; }
        p.flags_off = (2 * P * max_floats * sizeof(float) + 255) & ~(size_t)255;
        p.bytes = p.flags_off + 2 * P * P2P_WGS * sizeof(unsigned);
        size_t seg_max = 0;
        if (prompt_floats > max_floats) {                  // the prompt-sized area: 4 P segment slots (2 sets x {partials, reduced}) + its flag rows
            seg_max = ((prompt_floats / 4 + P - 1) / P) * 4;
            if (4 * P * seg_max * sizeof(float) >= ((size_t)1 << 31)) { err = "prompt_floats: the exchange area must stay below 2 GB"; return -1; }
            p.big_off = (p.bytes + 255) & ~(size_t)255;
            p.bflags_off = p.big_off + 4 * P * seg_max * sizeof(float);
            p.bytes = p.bflags_off + 4 * P * RSAG_MAX_WGS * sizeof(unsigned);
        }
        if (hipMalloc((void **)&p.local, p.bytes) != hipSuccess || hipMalloc((void **)&p.epoch, 2 * (P2P_WGS + RSAG_MAX_WGS) * sizeof(unsigned)) != hipSuccess) { err = "hipMalloc of the exchange buffer failed"; return -1; }
        if (hipMemset(p.local, 0, p.bytes) != hipSuccess || hipMemset(p.epoch, 0, 2 * (P2P_WGS + RSAG_MAX_WGS) * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { err = "clearing the exchange buffer failed"; return -1; }
        p.state = p.epoch + (P2P_WGS + RSAG_MAX_WGS);
        if (!p.pending && hipHostMalloc((void **)&p.pending, 64, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc of the exchange's pending word failed"; return -1; }
        *p.pending = 0;
        p.max_floats = max_floats;
        p.prompt_floats = seg_max ? prompt_floats : 0;
        p.bepoch = p.epoch + P2P_WGS;
        p.dev.seg_max = seg_max;
    } else if (p.max_floats != max_floats || (prompt_floats > max_floats ? prompt_floats : 0) != p.prompt_floats) { err = "the exchange buffer exists with another size"; return -1; }
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, p.local);
    if (e != hipSuccess) { err = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e); return -1; }
    std::memcpy(out, &h, sizeof h);
    return TP_P2P_HANDLE_BYTES;
}

int tp_p2p_enable(const void *handles, size_t len, std::string &err) {
    std::lock_guard<std::mutex> lk(g_mu);
    P2PState &p = g_p2p;
    const int P = g_grp.size;
    if (!p.local) { err = "tp_p2p_local_handle first"; return -1; }
    if (!handles || len != (size_t)P * TP_P2P_HANDLE_BYTES) { err = "need one 64-byte handle per rank"; return -1; }
    for (int q = 0; q < P; q++) {
        if (q == g_grp.rank) { p.peer[q] = p.local; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const uint8_t *)handles + (size_t)q * TP_P2P_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { err = "hipIpcOpenMemHandle(rank " + std::to_string(q) + "): " + hipGetErrorString(e); return -1; }
        p.peer[q] = ptr;
    }
    P2PDev d{};
    for (int q = 0; q < P; q++) {
        d.data[q] = (float *)p.peer[q]; d.flags[q] = (unsigned *)((uint8_t *)p.peer[q] + p.flags_off);
        d.big[q] = (float *)((uint8_t *)p.peer[q] + p.big_off); d.bflags[q] = (unsigned *)((uint8_t *)p.peer[q] + p.bflags_off);
    }
    d.epoch = p.epoch; d.bepoch = p.bepoch; d.err = p.err; d.rank = g_grp.rank; d.size = P; d.max_floats = p.max_floats; d.seg_max = p.dev.seg_max;
    d.state = p.state; d.pending = p.pending;
    // ranks that exchange through the host callback share a device: their waits for a peer must END the launch instead of holding the device (P2PDev::state)
    { const char *ev = getenv("MI355_TP_YIELD"); d.yield_polls = (ev ? ev[0] == '1' : g_grp.host_fn != nullptr) ? 4096 : 0; }
    p.dev = d;
    p.on = true;
    p.prompt_on = p.prompt_floats > 0;
    if (const char *ev = getenv("MI355_TP_RSAG_WGS")) { const int v = atoi(ev); if (v >= 1 && v <= RSAG_MAX_WGS) p.rsag_wgs = v; }
    return 0;
}
bool tp_p2p_active() { return g_p2p.on; }
void tp_p2p_use(bool on) { g_p2p.on = on && g_p2p.peer[g_grp.rank] != nullptr; }
void tp_p2p_use_prompt(bool on) { g_p2p.prompt_on = on && g_p2p.prompt_floats > 0 && g_p2p.peer[g_grp.rank] != nullptr; }
void tp_p2p_set_error_word(unsigned *w) { g_p2p.err = w; g_p2p.dev.err = w; }
int64_t tp_p2p_exchanges() { return g_p2p.exchanges; }
int64_t tp_p2p_prompt_exchanges() { return g_p2p.prompt_exchanges; }

void tp_shutdown() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_p2p.local) {
        (void)hipDeviceSynchronize();
        for (int q = 0; q < P2P_MAX_RANKS; q++) if (g_p2p.peer[q] && g_p2p.peer[q] != g_p2p.local) (void)hipIpcCloseMemHandle(g_p2p.peer[q]);
        (void)hipFree(g_p2p.local);
        if (g_p2p.epoch) (void)hipFree(g_p2p.epoch);
        if (g_p2p.pending) (void)hipHostFree(g_p2p.pending);
    }
    g_p2p = P2PState();
    if (g_grp.comm && g_api.CommDestroy) (void)g_api.CommDestroy(g_grp.comm);
    if (g_grp.pinned) (void)hipHostFree(g_grp.pinned);
    g_grp = Group();
}

// A peer of the group is gone (the engine-formed row split, tp_split.cc): collectives already queued on this rank's stream would wait for it for ever - RCCL has no
// time-out of its own - and so would everything that drains the stream afterwards (the context's destructor).  ncclCommAbort ends them with an error.
void tp_abort() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_grp.comm && g_api.CommAbort) { (void)g_api.CommAbort(g_grp.comm); g_grp.comm = nullptr; }
}

bool tp_active() { return g_grp.comm != nullptr || g_grp.host_fn != nullptr || g_grp.null_group; }
int tp_rank() { return g_grp.rank; }
int tp_size() { return (g_grp.comm || g_grp.host_fn || g_grp.null_group) ? g_grp.size : 1; }
bool tp_uses_host() { return g_grp.host_fn != nullptr; }

hipError_t tp_all_reduce_sum(const float *send, float *recv, size_t n, hipStream_t st) {
    Group &g = g_grp;
    if (g_p2p.on && n <= g_p2p.max_floats && n >= 4) {          // decode-sized message: the one-shot peer-to-peer kernel
        g_p2p.exchanges++;
        g_p2p.dev.trace_slot = trace_next((int)n, 0); g_p2p.dev.trace = g_trace_on && g_trace.size() < (size_t)TRACE_MAX ? g_trace_dev : nullptr;
        if (g_p2p.dev.yield_polls > 0) return p2p_launch_until_done(false, send, recv, (int)n, st);
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_WGS), dim3(256), 0, st, send, recv, (int)n, g_p2p.dev);
        return hipGetLastError();
    }
    if (g_p2p.on && g_p2p.prompt_on && n > g_p2p.max_floats && n <= g_p2p.prompt_floats && !(n & 3)) {   // prompt-sized: reduce-scatter + all-gather over all links
        g_p2p.prompt_exchanges++;
        g_p2p.dev.trace_slot = trace_next((int)n, 1); g_p2p.dev.trace = g_trace_on && g_trace.size() < (size_t)TRACE_MAX ? g_trace_dev : nullptr;
        if (g_p2p.dev.yield_polls > 0) return p2p_launch_until_done(true, send, recv, (int)n, st);
        hipLaunchKernelGGL(p2p_rsag_kernel, dim3(g_p2p.rsag_wgs), dim3(256), 0, st, send, recv, (int)n, g_p2p.dev);
        return hipGetLastError();
    }
    if (g.host_fn) return host_exchange(const_cast<float *>(send), recv, n, n, 0, n, 0, st);
    if (!g.comm) return send == recv ? hipSuccess : hipMemcpyAsync(recv, send, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    return g_api.AllReduce(send, recv, n, ncclFloat32, ncclSum, g.comm, st) == ncclSuccess ? hipSuccess : hipErrorUnknown;
}

hipError_t tp_all_gather(const float *send, float *recv, size_t n, hipStream_t st) {
    Group &g = g_grp;
    if (g.host_fn) return host_exchange(const_cast<float *>(send), recv, n * (size_t)g.size, n, n * (size_t)g.rank, n, 1, st);
    if (!g.comm) return send == recv ? hipSuccess : hipMemcpyAsync(recv, send, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    return g_api.AllGather(send, recv, n, ncclFloat32, g.comm, st) == ncclSuccess ? hipSuccess : hipErrorUnknown;
}

}  // namespace mi355
