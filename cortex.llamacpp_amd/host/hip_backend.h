// hip_backend.h — the product IBackend: mi355::Model + mi355::Context (HIP arithmetic) + the GGUF tokenizer, created from a
// /loadmodel body the way the reference's LlamaEngine::LoadModelImpl fills common_params (src/llama_engine.cc:587-700).
#pragma once

#include "engine.h"

namespace mi355 {

// BackendFactory for LlamaEngine; device-only (no CPU fallback): ngl is accepted as a placement hint and every layer is placed in HBM.
std::unique_ptr<IBackend> make_hip_backend(const Json &load_body, BackendInfo &info, std::string &err);

}  // namespace mi355
