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
void tp_abort();            // a peer is gone: end this rank's queued RCCL collectives with an error (ncclCommAbort) instead of waiting for it for ever
int tp_rank();
bool tp_active();         // a group exists (possibly of one rank)
int tp_size();            // 1 = no group, or a group of one
bool tp_uses_host();      // host transport: no stream capture

// ---- one-shot peer-to-peer all-reduce for the decode-sized messages (SURVEY.md §8e "fast path"): 2 L exchanges of 16 - 32 KB per token are
// latency-bound, and a collective library call costs 10 - 20 us each.  Every rank owns one buffer (IPC-exported device memory) of 2 sets x P slots;
// an exchange = each rank writes its partial into slot `rank` of EVERY rank's buffer (xGMI peer stores), raises a per-slice flag there, waits for
// the P flags of its own buffer and adds the P slots in rank order (the same order on every rank: all ranks hold the same bits).  One kernel per
// exchange, capturable into the decode graphs; the two sets alternate, so a rank that runs one exchange ahead never overwrites what a peer still
// reads.  Messages above `max_floats` (prompt batches) keep the base transport (RCCL / host).
// Bootstrap: tp_p2p_local_handle on every rank -> all-gather the 64-byte handles over any side channel (the one that carried the RCCL id) ->
// tp_p2p_enable with all of them, after tp_init / tp_set_host_exchange.
constexpr int TP_P2P_HANDLE_BYTES = 64;
// prompt_floats > max_floats additionally sizes the buffer for the reduce-scatter + all-gather kernel (tp_comm.cc p2p_rsag_kernel): messages of
// (max_floats, prompt_floats] floats - the prompt batches' n_embd x n_ubatch partial sums - are cut into one segment per rank, every rank stores its
// part of segment q into rank q's buffer (all links busy at once), the owner adds in rank order and stores the sum back to everybody.
int tp_p2p_local_handle(void *out, size_t cap, size_t max_floats, size_t prompt_floats, std::string &err);   // returns TP_P2P_HANDLE_BYTES or < 0
int tp_p2p_enable(const void *handles, size_t len, std::string &err);                  // len = size * TP_P2P_HANDLE_BYTES, rank-major
bool tp_p2p_active();
void tp_p2p_use(bool on);                            // measurement: route the small all-reduces back to the base transport (contexts created afterwards)
void tp_p2p_set_error_word(unsigned *host_word);     // pinned host word ORed with 32 when a bounded wait of the exchange kernel gives up
void tp_p2p_use_prompt(bool on);                     // the prompt-sized kernel alone (on after tp_p2p_enable when the buffer was sized for it)
int64_t tp_p2p_exchanges();                          // diagnosis: all-reduces that took the peer-to-peer kernel
int64_t tp_p2p_prompt_exchanges();                   // ... the reduce-scatter + all-gather kernel

// recv[i] = sum over ranks of send[i]; send may equal recv
hipError_t tp_all_reduce_sum(const float *send, float *recv, size_t n, hipStream_t st);
// recv[r * n + i] = rank r's send[i]
hipError_t tp_all_gather(const float *send, float *recv, size_t n, hipStream_t st);

}  // namespace mi355
