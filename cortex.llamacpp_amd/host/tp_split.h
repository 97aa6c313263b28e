// tp_split.h — ONE /loadmodel for the row split: `"split_mode": "row"` in the load body (the reference's engine drives every visible device from one
// call, src/llama_engine.cc:609-611; SURVEY.md §2b proposes split_mode / tensor_split / main_gpu for choosing how).  The engine process is rank 0 and forms the
// group ITSELF: it starts one worker process per further rank (bin/mi355_tp_worker, a fresh process that touches its GPU only after it has been told which one),
// hands it the RCCL id - or, where ranks must share a device, the descriptor of a shared-memory exchange segment (shm_exchange.h) - over a socket pair, and
// from then on sends every batch and every KV operation to the workers before it runs it itself: the ranks step in lock-step, the exchange kernels meet, the
// logits rows are gathered on every rank, rank 0 samples, and the sampled token reaches the workers as part of the next batch.  Every wait for a worker is
// bounded; the error names the rank.
#pragma once

#include <memory>
#include <string>

#include "engine.h"

namespace mi355 {

// true when the load body asks for the row split (split_mode == "row")
bool tp_split_requested(const Json &load_body);
// rank 0's side: workers + this process's own shard, as one IBackend.  `make_local` builds a backend from a (rewritten) load body in this process.
std::unique_ptr<IBackend> make_split_backend(const Json &load_body, BackendInfo &info, std::string &err);
// a worker process's whole life (bin/mi355_tp_worker -> mi355_tp_worker_main): returns the process's exit code
int tp_split_worker_main(int sock_fd);

}  // namespace mi355
