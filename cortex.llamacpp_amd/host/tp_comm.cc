// tp_comm.cc — RCCL (over xGMI) transport of the row-split exchange step, plus the host-callback transport used to
// validate the sharding where every rank cannot own a GPU.  See tp_comm.h.
#include "tp_comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>     // types and enums only: the functions are looked up in the library at run time

#include <cstring>
#include <mutex>

namespace mi355 {

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
constexpr int P2P_MAX_RANKS = 8, P2P_WGS = 16, P2P_SPIN_LIMIT = 1 << 22;
struct P2PDev {                        // by value into the kernel
    float *data[P2P_MAX_RANKS];        // base of every rank's slot area: [2 sets][P slots][max_floats]
    unsigned *flags[P2P_MAX_RANKS];    // base of every rank's flag area: [2 sets][P][P2P_WGS]
    unsigned *epoch;                   // this rank's private counters [P2P_WGS]
    unsigned *err;                     // nullable: pinned host word
    int rank, size;
    size_t max_floats;
};
struct P2PState {
    bool on = false;
    uint8_t *local = nullptr;
    size_t bytes = 0, flags_off = 0;
    void *peer[P2P_MAX_RANKS] = {};
    unsigned *epoch = nullptr, *err = nullptr;
    size_t max_floats = 0;
    int64_t exchanges = 0;
    P2PDev dev{};
};
P2PState g_p2p;

// one workgroup = one slice of the message, start to finish: no device-wide step inside the kernel
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(const float *send, float *recv, int n, const P2PDev a) {
    __shared__ unsigned e_sh;
    const int tid = (int)threadIdx.x, w = (int)blockIdx.x, P = a.size;
    if (tid == 0) e_sh = a.epoch[w] + 1u;
    __syncthreads();
    const unsigned e = e_sh;
    const size_t set = e & 1u;
    const int per = (((n + 3) / 4 + (int)gridDim.x - 1) / (int)gridDim.x) * 4;
    const int lo = w * per, hi = lo + per < n ? lo + per : n;
    // 1. this rank's slice into slot `rank` of every rank's buffer (peer stores over xGMI; the own copy too), then one flag per peer
    const size_t my_slot = (set * (size_t)P + (size_t)a.rank) * a.max_floats;
    for (int i = lo + tid; i < hi; i += 256) {
        const float v = send[i];
        for (int q = 0; q < P; q++) __hip_atomic_store(a.data[q] + my_slot + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();                               // every thread: its stores have left for their owners before the flags do
    __syncthreads();
    if (tid < P) __hip_atomic_store(a.flags[tid] + (set * (size_t)P + (size_t)a.rank) * P2P_WGS + w, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. everybody's slice w has arrived in MY buffer once my P flags of this set carry e
    if (tid < P) {
        const unsigned *f = a.flags[a.rank] + (set * (size_t)P + (size_t)tid) * P2P_WGS + w;
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            if (++spins >= P2P_SPIN_LIMIT) { if (a.err) __hip_atomic_fetch_or(a.err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    // 3. the P slots added in rank order (system-scope loads: the lines were written by other devices)
    const float *mine = a.data[a.rank] + set * (size_t)P * a.max_floats;
    for (int i = lo + tid; i < hi; i += 256) {
        float acc = __hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int q = 1; q < P; q++) acc += __hip_atomic_load(mine + (size_t)q * a.max_floats + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        recv[i] = acc;
    }
    if (tid == 0) a.epoch[w] = e;
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

int tp_p2p_local_handle(void *out, size_t cap, size_t max_floats, std::string &err) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || cap < (size_t)TP_P2P_HANDLE_BYTES) { err = "handle buffer must hold 64 bytes"; return -1; }
    if (!(g_grp.comm || g_grp.host_fn) || g_grp.size < 2 || g_grp.size > P2P_MAX_RANKS) { err = "peer-to-peer exchange needs a row-split group of 2..8 ranks first"; return -1; }
    if (max_floats == 0 || (max_floats & 3)) { err = "max_floats must be a positive multiple of 4"; return -1; }
    static_assert(sizeof(hipIpcMemHandle_t) == TP_P2P_HANDLE_BYTES, "IPC handle size");
    P2PState &p = g_p2p;
    if (!p.local) {
        const size_t P = (size_t)g_grp.size;
        p.flags_off = (2 * P * max_floats * sizeof(float) + 255) & ~(size_t)255;
        p.bytes = p.flags_off + 2 * P * P2P_WGS * sizeof(unsigned);
        if (hipMalloc((void **)&p.local, p.bytes) != hipSuccess || hipMalloc((void **)&p.epoch, P2P_WGS * sizeof(unsigned)) != hipSuccess) { err = "hipMalloc of the exchange buffer failed"; return -1; }
        if (hipMemset(p.local, 0, p.bytes) != hipSuccess || hipMemset(p.epoch, 0, P2P_WGS * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { err = "clearing the exchange buffer failed"; return -1; }
        p.max_floats = max_floats;
    } else if (p.max_floats != max_floats) { err = "the exchange buffer exists with another size"; return -1; }
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
    for (int q = 0; q < P; q++) { d.data[q] = (float *)p.peer[q]; d.flags[q] = (unsigned *)((uint8_t *)p.peer[q] + p.flags_off); }
    d.epoch = p.epoch; d.err = p.err; d.rank = g_grp.rank; d.size = P; d.max_floats = p.max_floats;
    p.dev = d;
    p.on = true;
    return 0;
}
bool tp_p2p_active() { return g_p2p.on; }
void tp_p2p_use(bool on) { g_p2p.on = on && g_p2p.peer[g_grp.rank] != nullptr; }
void tp_p2p_set_error_word(unsigned *w) { g_p2p.err = w; g_p2p.dev.err = w; }
int64_t tp_p2p_exchanges() { return g_p2p.exchanges; }

void tp_shutdown() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_p2p.local) {
        (void)hipDeviceSynchronize();
        for (int q = 0; q < P2P_MAX_RANKS; q++) if (g_p2p.peer[q] && g_p2p.peer[q] != g_p2p.local) (void)hipIpcCloseMemHandle(g_p2p.peer[q]);
        (void)hipFree(g_p2p.local);
        if (g_p2p.epoch) (void)hipFree(g_p2p.epoch);
    }
    g_p2p = P2PState();
    if (g_grp.comm && g_api.CommDestroy) (void)g_api.CommDestroy(g_grp.comm);
    if (g_grp.pinned) (void)hipHostFree(g_grp.pinned);
    g_grp = Group();
}

bool tp_active() { return g_grp.comm != nullptr || g_grp.host_fn != nullptr || g_grp.null_group; }
int tp_rank() { return g_grp.rank; }
int tp_size() { return (g_grp.comm || g_grp.host_fn || g_grp.null_group) ? g_grp.size : 1; }
bool tp_uses_host() { return g_grp.host_fn != nullptr; }

hipError_t tp_all_reduce_sum(const float *send, float *recv, size_t n, hipStream_t st) {
    Group &g = g_grp;
    if (g_p2p.on && n <= g_p2p.max_floats && n >= 4) {          // decode-sized message: the one-shot peer-to-peer kernel
        g_p2p.exchanges++;
        hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(P2P_WGS), dim3(256), 0, st, send, recv, (int)n, g_p2p.dev);
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
