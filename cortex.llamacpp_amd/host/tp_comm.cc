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

void tp_shutdown() {
    std::lock_guard<std::mutex> lk(g_mu);
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
