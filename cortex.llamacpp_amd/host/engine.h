// engine.h — host-side mirror of the reference's plugin surface: class EngineI (base/cortex-common/enginei.h:13-74) as
// implemented by LlamaEngine (src/llama_engine.{h,cc}).  Same method names, same request keys, same
// status object {is_done, has_error, is_stream, status_code} and response bodies (llama_engine.cc:180-270, 363-500,
// 734-1113).  The reference's ABI types (Json::Value, trantor) are unavailable offline, so bodies are carried in
// mi355::Json; `include/mi355_llama.h` additionally exposes this class through JSON strings (mi355_engine_*).
#pragma once

#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "backend_iface.h"
#include "json.h"
#include "server_context.h"

namespace mi355 {

// per-model worker pool: the reference's trantor::ConcurrentTaskQueue(n_parallel) (llama_engine.cc:720)
class TaskQueue {
  public:
    explicit TaskQueue(int n_threads);
    ~TaskQueue();
    void run(std::function<void()> f);
  private:
    std::vector<std::thread> workers_;
    std::queue<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

struct BackendInfo { uint64_t vram = 0, ram = 0, model_size = 0; };
// creates the arithmetic backend for a /loadmodel body; product: make_hip_backend (hip_backend.cc)
using BackendFactory = std::function<std::unique_ptr<IBackend>(const Json &load_body, BackendInfo &info, std::string &err)>;

class LlamaEngine {
  public:
    using Callback = std::function<void(Json &&status, Json &&body)>;
    explicit LlamaEngine(BackendFactory factory);
    ~LlamaEngine();

    void HandleChatCompletion(const Json &body, Callback cb);
    void HandleEmbedding(const Json &body, Callback cb);
    void LoadModel(const Json &body, Callback cb);
    void UnloadModel(const Json &body, Callback cb);
    void GetModelStatus(const Json &body, Callback cb);
    void GetModels(const Json &body, Callback cb);
    bool IsSupported(const std::string &f) const;
    void StopInferencing(const std::string &model_id);
    // enginei.h:14-35: EngineLoadOption {engine_path, deps_path, is_custom_engine_path, log_path, max_log_lines, log_level} / Load / Unload, and
    // SetFileLogger / SetLogLevel (:69-71), as LlamaEngine implements them (llama_engine.cc:289-303, 502-548): Load = SetFileLogger + SetLogLevel
    struct EngineLoadOption { std::string engine_path, deps_path, log_path; bool is_custom_engine_path = false; int max_log_lines = 0; int log_level = 2; };
    void Load(const EngineLoadOption &opts);
    void Unload();
    void SetFileLogger(int max_log_lines, const std::string &log_path);
    void SetLogLevel(int log_level);                       // trantor::Logger::LogLevel numbers (kTrace 0 .. kFatal 5)

    static std::string GetModelId(const Json &body);   // llama_utils.h:153-177

  private:
    struct ServerInfo {
        std::unique_ptr<IBackend> backend;
        std::unique_ptr<LlamaServerContext> ctx;
        std::unique_ptr<TaskQueue> q;
        std::string user_prompt, ai_prompt, system_prompt, pre_prompt;
        int repeat_last_n = 32;
        bool caching_enabled = true;
        Json stop_words;
        std::string model_type = "llm";
        int64_t start_time = 0;
        BackendInfo info;
        int ngl = 0;
        std::string grammar_file_content;   // load option grammar_file: its text constrains every completion of this model (:573-585, 812-814)
    };
    bool LoadModelImpl(const Json &body, std::string &err);
    bool CheckModelLoaded(const Callback &cb, const std::string &model_id);
    void HandleInferenceImpl(const Json &body, Callback cb);
    void WarmUpModel(ServerInfo &si);

    BackendFactory factory_;
    std::mutex map_mutex_;
    std::map<std::string, std::shared_ptr<ServerInfo>> server_map_;
    std::mutex stop_mutex_;
    std::set<std::string> force_stop_;
    std::atomic<int> no_of_requests_{0};
};

}  // namespace mi355
