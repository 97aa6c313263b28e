// tp_comm.h — the one exchange step of the row-split (tensor-parallel) path: one process per GPU, every rank holds a
// row / column slice of each projection (runtime.cc model_load), and the partial sums of attn_output and ffn_down are
// summed across ranks twice per layer.  Transport = RCCL over xGMI (librccl is opened on first use, so a
// single-GPU process never loads it); collectives are issued on the context's stream and are captured into its
// hipGraphs like any other node.
//
// A second transport exists for validation on a box where the ranks cannot each own a GPU (RCCL refuses two ranks on one
// device): a host exchange callback (the tests implement it over torch.distributed/gloo).  The buffer is staged through
// pinned memory around the callback, the stream is drained, graphs are off.  It moves the same bytes at the same places.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace mi355 {

// op: 0 = sum in place over `n` floats; 1 = all-gather: `n` floats per rank, this rank's part already sits at
// buf + rank * n of the size * n float buffer.  Returns 0 on success.
typedef int (*tp_host_exchange_fn)(void *user, float *buf, size_t n, int op);

int tp_unique_id(void *out, size_t cap, std::string &err);                       // 128 bytes (ncclGetUniqueId), made by rank 0
int tp_init(int rank, int size, const void *id, size_t id_len, std::string &err); // ncclCommInitRank on the current device
void tp_set_host_exchange(tp_host_exchange_fn fn, void *user, int rank, int size);
// Measurement aid: a group of `size` ranks of which only this one exists — every exchange degenerates to a device copy of
// this rank's own part (results are NOT the model's; timings are this rank's compute without communication).
void tp_set_null_group(int rank, int size);
void tp_shutdown();
int tp_rank();
bool tp_active();         // a group exists (possibly of one rank)
int tp_size();            // 1 = no group, or a group of one
bool tp_uses_host();      // host transport: no stream capture

// recv[i] = sum over ranks of send[i]; send may equal recv
hipError_t tp_all_reduce_sum(const float *send, float *recv, size_t n, hipStream_t st);
// recv[r * n + i] = rank r's send[i]
hipError_t tp_all_gather(const float *send, float *recv, size_t n, hipStream_t st);

}  // namespace mi355
