// mi355_engine_adapter.cc — the translation unit a cortex.llamacpp maintainer adds to build `libengine.so` on top of libmi355_llama.so:
// class Mi355Engine implements every slot of EngineI (base/cortex-common/enginei.h:31-73, in vtable order: dtor, Load, Unload, HandleChatCompletion,
// HandleEmbedding, LoadModel, UnloadModel, GetModelStatus, IsSupported, GetModels, SetFileLogger, SetLogLevel, StopInferencing) over the C-ABI of
// include/mi355_llama.h, and `get_engine()` is what the host dlsym()s (src/llama_engine.cc:1300-1304; examples/server/server.cc:14-24).
// Requests cross as JSON text (the ABI types of EngineI — Json::Value, trantor::Logger::LogLevel, std::filesystem::path — stay on this side).
//
// Build (in the cortex.llamacpp tree, jsoncpp and trantor on the include path as for src/llama_engine.cc):
//   g++ -std=c++17 -fPIC -shared -I base -I <repo>/include integration/mi355_engine_adapter.cc -L <repo>/cortex.llamacpp_amd/lib -lmi355_llama -ljsoncpp -o libengine.so
// tests/test_adapter_compiles.py compiles it against the reference's own enginei.h and stub json / trantor headers (tests/stubs/): the class is concrete and
// get_engine links.  That test checks the SHAPE of the boundary only.
#include <functional>
#include <memory>
#include <string>

#include "cortex-common/enginei.h"
#include "json/json.h"
#include "mi355_llama.h"

namespace {

using Cb = std::function<void(Json::Value&&, Json::Value&&)>;

// called once per response / per SSE chunk, possibly on another thread and after the request call has returned (enginei.h:37-51)
void Tramp(const char* status_json, const char* body_json, void* user) {
  auto* cb = static_cast<Cb*>(user);
  Json::Value status, body;
  Json::Reader reader;
  reader.parse(status_json, status);
  reader.parse(body_json, body);
  (*cb)(std::move(status), std::move(body));
}
// the library is through with a request (its last callback has returned - or it ends without a terminal one: a stream stopped by StopInferencing, a
// stream whose model was unloaded before its task ran): the std::function dies here, as the reference's dies with its queued task
void Release(void* user) { delete static_cast<Cb*>(user); }

std::string Dump(const std::shared_ptr<Json::Value>& j) { return j ? Json::FastWriter().write(*j) : std::string("{}"); }

// log lines of the backend re-emitted through the host's logger (the reference routes llama.cpp's log the same way, src/llama_engine.cc:313-330)
void LogBridge(int level, const char* line, void*) {
  switch (level) {
    case 0: LOG_TRACE << line; break;
    case 1: LOG_DEBUG << line; break;
    case 2: LOG_INFO << line; break;
    case 3: LOG_WARN << line; break;
    default: LOG_ERROR << line; break;
  }
}

}  // namespace

class Mi355Engine : public EngineI {
 public:
  Mi355Engine() : e_(mi355_engine_create()) { mi355_engine_set_release_callback(e_, Release); }
  ~Mi355Engine() override { mi355_engine_destroy(e_); }

  void Load(EngineLoadOption opts) final {
    mi355_engine_load(e_, opts.engine_path.string().c_str(), opts.deps_path.string().c_str(), opts.is_custom_engine_path ? 1 : 0,
                      opts.log_path.string().c_str(), opts.max_log_lines, static_cast<int>(opts.log_level));
  }
  void Unload(EngineUnloadOption) final { mi355_engine_unload(e_); }

  void HandleChatCompletion(std::shared_ptr<Json::Value> j, Cb&& cb) final {
    mi355_engine_handle_chat_completion(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb)));
  }
  void HandleEmbedding(std::shared_ptr<Json::Value> j, Cb&& cb) final {
    mi355_engine_handle_embedding(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb)));
  }
  void LoadModel(std::shared_ptr<Json::Value> j, Cb&& cb) final { mi355_engine_load_model(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb))); }
  void UnloadModel(std::shared_ptr<Json::Value> j, Cb&& cb) final { mi355_engine_unload_model(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb))); }
  void GetModelStatus(std::shared_ptr<Json::Value> j, Cb&& cb) final {
    mi355_engine_get_model_status(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb)));
  }
  bool IsSupported(const std::string& f) final { return mi355_engine_is_supported(e_, f.c_str()) != 0; }
  void GetModels(std::shared_ptr<Json::Value> j, Cb&& cb) final { mi355_engine_get_models(e_, Dump(j).c_str(), Tramp, new Cb(std::move(cb))); }
  void SetFileLogger(int max_log_lines, const std::string& log_path) final { mi355_engine_set_file_logger(e_, max_log_lines, log_path.c_str()); }
  void SetLogLevel(trantor::Logger::LogLevel log_level) final {
    trantor::Logger::setLogLevel(log_level);
    mi355_engine_set_log_level(e_, static_cast<int>(log_level));
  }
  void StopInferencing(const std::string& model_id) final { mi355_engine_stop_inferencing(e_, model_id.c_str()); }

  // optional: send the backend's log lines through the host's trantor logger instead of the backend's own file / stderr sink
  void BridgeLogs() { mi355_engine_set_log_callback(e_, LogBridge, nullptr); }

 private:
  mi355_engine* e_;
};

extern "C" {
#if defined(_WIN32)
__declspec(dllexport)
#else
__attribute__((visibility("default")))
#endif
EngineI* get_engine() { return new Mi355Engine(); }
}
