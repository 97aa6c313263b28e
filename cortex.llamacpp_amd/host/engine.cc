// engine.cc — see engine.h.  Citations are into /root/reference/src/llama_engine.cc unless noted.
#include "engine.h"
#include "grammar.h"
#include "log.h"

#include <algorithm>
#include <chrono>
#include <ctime>
#include <random>

namespace mi355 {

namespace {
constexpr int k200OK = 200, k400BadRequest = 400, k409Conflict = 409, k500InternalServerError = 500;

Json make_status(bool is_done, bool has_error, bool is_stream, int code) {
    Json s = Json::object();
    s["is_done"] = is_done; s["has_error"] = has_error; s["is_stream"] = is_stream; s["status_code"] = code;
    return s;
}
Json message(const std::string &m) { Json j = Json::object(); j["message"] = m; return j; }

std::string rtrim(std::string s) { while (!s.empty() && (s.back() == ' ' || s.back() == '\n' || s.back() == '\t' || s.back() == '\r')) s.pop_back(); return s; }
void ltrim(std::string &s) { size_t i = 0; while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++; s.erase(0, i); }

std::string random_id(size_t n) {   // llama_utils::generate_random_string
    static const char cs[] = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789";
    static thread_local std::mt19937 gen{std::random_device{}()};
    std::uniform_int_distribution<size_t> d(0, sizeof(cs) - 2);
    std::string s;
    for (size_t i = 0; i < n; i++) s += cs[d(gen)];
    return s;
}

// chat.completion.chunk (CreateReturnJson :220-270)
std::string chunk_json(const std::string &content, const Json &finish_reason, bool include_usage, const Json *usage, const Json &logprobs) {
    Json root = Json::object();
    root["id"] = random_id(20); root["model"] = "_"; root["created"] = (int64_t)std::time(nullptr); root["object"] = "chat.completion.chunk";
    Json choices = Json::array();
    if (!usage) {
        Json choice = Json::object(), delta = Json::object();
        choice["index"] = 0;
        delta["content"] = content; delta["role"] = "assistant";
        choice["delta"] = delta;
        choice["finish_reason"] = finish_reason;
        if (!logprobs.empty()) choice["logprobs"] = logprobs;
        choices.push_back(choice);
    }
    root["choices"] = choices;
    if (include_usage) root["usage"] = usage ? *usage : Json();
    return root.dump();
}

// chat.completion (CreateFullReturnJson :180-218)
Json full_json(const std::string &content, int prompt_tokens, int completion_tokens, const Json &logprobs) {
    Json root = Json::object();
    root["id"] = random_id(20); root["model"] = "_"; root["created"] = (int64_t)std::time(nullptr); root["object"] = "chat.completion";
    root["system_fingerprint"] = "_";
    Json choice = Json::object(), msg = Json::object();
    choice["index"] = 0;
    msg["role"] = "assistant"; msg["content"] = content;
    choice["message"] = msg;
    choice["finish_reason"] = "stop";
    if (!logprobs.empty()) choice["logprobs"] = logprobs;
    Json choices = Json::array();
    choices.push_back(choice);
    root["choices"] = choices;
    Json usage = Json::object();
    usage["prompt_tokens"] = prompt_tokens; usage["completion_tokens"] = completion_tokens; usage["total_tokens"] = prompt_tokens + completion_tokens;
    root["usage"] = usage;
    return root;
}
}  // namespace

// ---------------------------------------------------------------- TaskQueue
TaskQueue::TaskQueue(int n) {
    for (int i = 0; i < std::max(n, 1); i++)
        workers_.emplace_back([this] {
            while (true) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(m_);
                    cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                    if (stop_ && q_.empty()) return;
                    f = std::move(q_.front());
                    q_.pop();
                }
                f();
            }
        });
}
TaskQueue::~TaskQueue() {
    { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
    cv_.notify_all();
    for (auto &w : workers_) if (w.joinable()) w.join();
}
void TaskQueue::run(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(m_); q_.push(std::move(f)); }
    cv_.notify_one();
}

// ---------------------------------------------------------------- engine
LlamaEngine::LlamaEngine(BackendFactory f) : factory_(std::move(f)) {}

LlamaEngine::~LlamaEngine() {
    std::lock_guard<std::mutex> lk(map_mutex_);
    for (auto &kv : server_map_) {
        kv.second->q.reset();                       // drain workers first (they hold the context)
        if (kv.second->ctx) kv.second->ctx->ReleaseResources();
    }
    server_map_.clear();
}

void LlamaEngine::Load(const EngineLoadOption &opts) {   // llama_engine.cc:289-299
    log_line(LOG_DEBUG, "Loading engine..");
    log_line(LOG_DEBUG, "Is custom engine path: %d", opts.is_custom_engine_path ? 1 : 0);
    log_line(LOG_DEBUG, "Engine path: %s", opts.engine_path.c_str());
    SetFileLogger(opts.max_log_lines, opts.log_path);
    SetLogLevel(opts.log_level);
    log_line(LOG_INFO, "Engine loaded successfully");
}
void LlamaEngine::Unload() { log_line(LOG_INFO, "Engine unloaded successfully"); }   // :301-303
void LlamaEngine::SetLogLevel(int log_level) { log_set_level(log_level); }          // :502-504
void LlamaEngine::SetFileLogger(int max_log_lines, const std::string &log_path) {   // :510-548
    if (!log_set_file(log_path, max_log_lines)) log_line(LOG_WARN, "cannot open log file %s", log_path.c_str());
}

bool LlamaEngine::IsSupported(const std::string &f) const {   // enginei.h:54-62
    return f == "HandleChatCompletion" || f == "HandleEmbedding" || f == "LoadModel" || f == "UnloadModel" || f == "GetModelStatus" ||
           f == "GetModels" || f == "SetFileLogger" || f == "SetLogLevel" || f == "StopInferencing";
}

std::string LlamaEngine::GetModelId(const Json &body) {   // llama_utils.h:153-177
    if (body["model"].is_string() && !body["model"].as_string().empty()) return body["model"].as_string();
    if (body["model_alias"].is_string() && !body["model_alias"].as_string().empty()) return body["model_alias"].as_string();
    for (const char *k : {"llama_model_path", "model_path"}) {
        if (body[k].is_string() && !body[k].as_string().empty()) {
            std::string p = body[k].as_string();
            const size_t sl = p.find_last_of("/\\");
            if (sl != std::string::npos) p = p.substr(sl + 1);
            const size_t dot = p.rfind(".gguf");
            if (dot != std::string::npos && dot + 5 == p.size()) p = p.substr(0, dot);
            return p;
        }
    }
    return "";
}

void LlamaEngine::LoadModel(const Json &body, Callback cb) {   // :363-423
    const std::string model_id = GetModelId(body);
    if (model_id.empty()) {
        log_line(LOG_INFO, "Model id is empty in request");
        cb(make_status(false, true, false, k400BadRequest), message("No model id found in request body"));
        return;
    }
    {
        std::lock_guard<std::mutex> lk(map_mutex_);
        auto it = server_map_.find(model_id);
        if (it != server_map_.end() && it->second->ctx && it->second->ctx->model_loaded_external) {
            log_line(LOG_INFO, "Model already loaded");
            cb(make_status(true, false, false, k409Conflict), message("Model already loaded"));
            return;
        }
    }
    std::string err;
    if (!LoadModelImpl(body, err)) {
        log_line(LOG_ERROR, "Failed to load model: %s", err.c_str());
        Json m = message("Failed to load model");
        if (!err.empty()) m["error"] = err;
        cb(make_status(false, true, false, k500InternalServerError), std::move(m));
    } else {
        log_line(LOG_INFO, "Model loaded successfully: %s", model_id.c_str());
        cb(make_status(true, false, false, k200OK), message("Model loaded successfully"));
    }
}

bool LlamaEngine::LoadModelImpl(const Json &body, std::string &err) {   // :547-732
    const std::string model_id = GetModelId(body);
    auto si = std::make_shared<ServerInfo>();
    const std::string path = body["llama_model_path"].is_string() ? body["llama_model_path"].as_string() : body["model_path"].str_or("");
    if (path.empty()) { err = "Missing model path in request"; return false; }
    // `mmproj` (src/llama_engine.cc:553-562) turns the context multimodal (LLaVA: src/llama_server_context.cc:184-230): the backend factory loads the projector
    // file beside the model (hip_backend.cc) and fails the load when it cannot
    if (!body["mmproj"].is_null()) {
        if (!body["mmproj"].is_string() || body["mmproj"].as_string().empty()) { err = "mmproj: expected the path of a projector file"; return false; }
        log_line(LOG_INFO, "MMPROJ FILE detected, multi-model enabled!");
    }
    if (body["grammar_file"].is_string()) {                                   // :573-585
        FILE *gf = fopen(body["grammar_file"].as_string().c_str(), "rb");
        if (!gf) { err = "Grammar file not found"; log_line(LOG_ERROR, "Grammar file not found"); return false; }
        char buf[4096];
        for (size_t n; (n = fread(buf, 1, sizeof buf, gf)) > 0;) si->grammar_file_content.append(buf, n);
        fclose(gf);
        std::string gerr;
        if (!si->grammar_file_content.empty() && !Grammar::parse(si->grammar_file_content, gerr)) { err = "grammar_file: " + gerr; return false; }
    }
    si->backend = factory_(body, si->info, err);
    if (!si->backend) return false;
    ServerParams sp;
    sp.n_parallel = std::max(1, body.value<int>("n_parallel", 1));           // :620
    sp.cont_batching = body.value<bool>("cont_batching", true);               // :625-627
    sp.n_predict = body.value<int>("n_predict", -1);
    sp.model_alias = model_id;
    si->user_prompt = body.value<std::string>("user_prompt", "USER: ");       // :662-669
    si->ai_prompt = body.value<std::string>("ai_prompt", "ASSISTANT: ");
    si->system_prompt = body.value<std::string>("system_prompt", "ASSISTANT's RULE: ");
    si->pre_prompt = body.value<std::string>("pre_prompt", "");
    si->repeat_last_n = body.value<int>("repeat_last_n", 32);                 // :670-671
    si->caching_enabled = body.value<bool>("caching_enabled", true);          // :660-661
    si->stop_words = body["stop"];                                            // :672
    si->model_type = body.value<std::string>("model_type", "llm");
    if (body.value<bool>("embedding", false)) si->model_type = "embedding";
    if (si->backend->is_encoder()) si->model_type = "embedding";             // an encoder file cannot complete text: no warm-up completion, chat requests are refused
    si->ngl = body.value<int>("ngl", 300);
    si->start_time = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    si->ctx.reset(new LlamaServerContext(si->backend.get(), sp));
    si->ctx->Initialize();
    si->q.reset(new TaskQueue(sp.n_parallel));                                // :720-721
    if (si->model_type == "llm" && si->backend->vocab().has_vocab()) WarmUpModel(*si);   // :728-730
    std::lock_guard<std::mutex> lk(map_mutex_);
    server_map_[model_id] = si;
    return true;
}

void LlamaEngine::WarmUpModel(ServerInfo &si) {   // :1247-1267
    Json d = Json::object();
    d["prompt"] = "Hello"; d["n_predict"] = 2; d["n_probs"] = 0;
    const int id = si.ctx->RequestCompletion(d, false, false, -1);
    (void)si.ctx->NextResult(id);
}

void LlamaEngine::UnloadModel(const Json &body, Callback cb) {   // :425-445
    const std::string model_id = GetModelId(body);
    if (!CheckModelLoaded(cb, model_id)) return;
    std::shared_ptr<ServerInfo> si;
    {
        std::lock_guard<std::mutex> lk(map_mutex_);
        si = server_map_[model_id];
        server_map_.erase(model_id);
    }
    si->ctx->ReleaseResources();
    si->q.reset();
    cb(make_status(true, false, false, k200OK), message("Model unloaded successfully"));
    log_line(LOG_INFO, "Model unloaded successfully");
}

void LlamaEngine::GetModelStatus(const Json &body, Callback cb) {   // :447-466
    const std::string model_id = GetModelId(body);
    if (!CheckModelLoaded(cb, model_id)) return;
    Json j = Json::object();
    j["model_loaded"] = true;
    {
        std::lock_guard<std::mutex> lk(map_mutex_);
        auto it = server_map_.find(model_id);
        j["model_data"] = it != server_map_.end() && it->second->ctx ? it->second->ctx->GetModelProps().dump() : std::string("{}");   // :457
    }
    cb(make_status(true, false, false, k200OK), std::move(j));
    log_line(LOG_INFO, "Model status responded");
}

void LlamaEngine::GetModels(const Json &, Callback cb) {   // :468-500
    Json arr = Json::array();
    {
        std::lock_guard<std::mutex> lk(map_mutex_);
        for (const auto &kv : server_map_) {
            if (!kv.second->ctx || !kv.second->ctx->model_loaded_external) continue;
            Json v = Json::object();
            v["id"] = kv.first; v["engine"] = "cortex.llamacpp"; v["start_time"] = kv.second->start_time;
            v["model_size"] = kv.second->info.model_size; v["vram"] = kv.second->info.vram; v["ram"] = kv.second->info.ram;
            v["object"] = "model";
            arr.push_back(std::move(v));
        }
    }
    Json root = Json::object();
    root["object"] = "list";
    root["data"] = arr;
    cb(make_status(true, false, false, k200OK), std::move(root));
}

bool LlamaEngine::CheckModelLoaded(const Callback &cb, const std::string &model_id) {   // :1225-1245
    std::lock_guard<std::mutex> lk(map_mutex_);
    auto it = server_map_.find(model_id);
    if (it == server_map_.end() || !it->second->ctx || !it->second->ctx->model_loaded_external) {
        Callback c = cb;
        c(make_status(false, true, false, k409Conflict), message("Model has not been loaded, please load model into cortex.llamacpp"));
        return false;
    }
    return true;
}

void LlamaEngine::StopInferencing(const std::string &model_id) {   // :502-508
    std::lock_guard<std::mutex> lk(stop_mutex_);
    force_stop_.insert(model_id);
}

// base64 of the raw little-endian f32 bytes (encoding_format = "base64"; llama_utils::base64Encode / FloatVectorToBytes)
static std::string base64_bytes(const uint8_t *p, size_t n) {
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((n + 2) / 3 * 4);
    for (size_t i = 0; i < n; i += 3) {
        const uint32_t b0 = p[i], b1 = i + 1 < n ? p[i + 1] : 0, b2 = i + 2 < n ? p[i + 2] : 0, w = b0 << 16 | b1 << 8 | b2;
        out.push_back(tbl[w >> 18 & 63]); out.push_back(tbl[w >> 12 & 63]);
        out.push_back(i + 1 < n ? tbl[w >> 6 & 63] : '='); out.push_back(i + 2 < n ? tbl[w & 63] : '=');
    }
    return out;
}
static std::string base64_floats(const std::vector<float> &v) { return base64_bytes(reinterpret_cast<const uint8_t *>(v.data()), v.size() * sizeof(float)); }

static Json embedding_payload(const Json &embedding, int index, bool is_base64) {   // CreateEmbeddingPayload :92-114
    Json item = Json::object();
    item["object"] = "embedding";
    item["index"] = index;
    if (is_base64) {
        std::vector<float> v;
        for (const Json &x : embedding.items()) v.push_back((float)x.as_double());
        item["embedding"] = base64_floats(v);
    } else {
        item["embedding"] = embedding;
    }
    return item;
}

static bool all_int32(const Json &a) {   // AreAllElementsInt32
    if (!a.is_array() || a.size() == 0) return false;
    for (const Json &x : a.items()) if (!x.is_int()) return false;
    return true;
}

void LlamaEngine::HandleEmbedding(const Json &body, Callback cb) {   // :353-361, HandleEmbeddingImpl :1115-1223
    const std::string model_id = GetModelId(body);
    if (!CheckModelLoaded(cb, model_id)) return;
    std::shared_ptr<ServerInfo> si;
    { std::lock_guard<std::mutex> lk(map_mutex_); si = server_map_[model_id]; }
    ++no_of_requests_;
    si->q->run([si, cb, body, model_id]() mutable {
        LlamaServerContext &llama = *si->ctx;
        Json data = Json::array();
        const bool is_base64 = body.value<std::string>("encoding_format", "float") == "base64";
        int prompt_tokens = 0;
        auto request = [&](const Json &elem) -> int {             // one prompt: a string, or an array of token ids
            Json d = Json::object();
            d["n_predict"] = 0;
            if (elem.is_string()) d["prompt"] = elem.as_string();
            else { d["prompt"] = "Mock prompt"; d["prompt_tokens"] = elem; }
            return llama.RequestCompletion(d, false, true, -1);
        };
        auto collect = [&](int task_id, int index) {
            TaskResult r = llama.NextResult(task_id);
            if (r.error || !r.result_json["embedding"].is_array()) { data.push_back(embedding_payload(Json::array(), index, is_base64)); return; }
            prompt_tokens += (int)r.result_json["tokens_evaluated"].as_int();
            data.push_back(embedding_payload(r.result_json["embedding"], index, is_base64));
        };
        const Json &input = body["input"];
        if (input.is_string() || all_int32(input)) {
            collect(request(input), 0);
        } else if (input.is_array()) {
            std::vector<int> ids;
            for (const Json &elem : input.items())
                if (elem.is_string() || all_int32(elem)) ids.push_back(request(elem));
            for (size_t i = 0; i < ids.size(); i++) collect(ids[i], (int)i);
        }
        Json root = Json::object(), usage = Json::object();
        root["data"] = data;
        root["model"] = model_id;
        root["object"] = "list";
        usage["prompt_tokens"] = prompt_tokens;
        usage["total_tokens"] = prompt_tokens;
        root["usage"] = usage;
        cb(make_status(true, false, false, k200OK), std::move(root));
    });
}

void LlamaEngine::HandleChatCompletion(const Json &body, Callback cb) {   // :341-351
    if (!CheckModelLoaded(cb, GetModelId(body))) return;
    HandleInferenceImpl(body, std::move(cb));
}

void LlamaEngine::HandleInferenceImpl(const Json &body, Callback cb) {   // :734-1113
    const std::string model_id = GetModelId(body);
    std::shared_ptr<ServerInfo> si;
    { std::lock_guard<std::mutex> lk(map_mutex_); si = server_map_[model_id]; }
    if (si->model_type == "embedding") {   // :739-750
        cb(make_status(false, true, false, k400BadRequest), message("Model type is wrong. Expected to be llm type"));
        return;
    }
    const int request_id = ++no_of_requests_;
    (void)request_id;
    // request defaults: chat_completion_request.h:60-92
    const bool stream = body.value<bool>("stream", false);
    const bool include_usage = body["stream_options"].value<bool>("include_usage", false);
    int n_probs = body.value<int>("n_probs", 0);
    if (body.value<bool>("logprobs", false)) n_probs = std::max(n_probs, body.value<int>("top_logprobs", 1));
    Json data = Json::object();
    data["cache_prompt"] = si->caching_enabled;
    data["n_keep"] = 0;
    data["stream"] = stream;
    data["n_predict"] = body.value<int>("max_tokens", 500);
    data["top_p"] = body.value<float>("top_p", 0.95f);
    data["temperature"] = body.value<float>("temperature", 0.8f);
    data["frequency_penalty"] = body.value<float>("frequency_penalty", 0.0f);
    data["presence_penalty"] = body.value<float>("presence_penalty", 0.0f);
    data["seed"] = body.value<int64_t>("seed", -1);
    data["dynatemp_range"] = body.value<float>("dynatemp_range", 0.0f);
    data["dynatemp_exponent"] = body.value<float>("dynatemp_exponent", 1.0f);
    data["top_k"] = body.value<int>("top_k", 40);
    data["min_p"] = body.value<float>("min_p", 0.05f);
    data["typical_p"] = body.value<float>("typ_p", 1.0f);
    data["repeat_last_n"] = body.value<int>("repeat_last_n", 64);
    data["repeat_penalty"] = body.value<float>("repeat_penalty", 1.1f);
    data["mirostat"] = body.value<int>("mirostat", 0);
    data["mirostat_tau"] = body.value<float>("mirostat_tau", 5.0f);
    data["mirostat_eta"] = body.value<float>("mirostat_eta", 0.1f);
    data["ignore_eos"] = body.value<bool>("ignore_eos", false);
    data["n_probs"] = n_probs;
    data["min_keep"] = body.value<int>("min_keep", 0);
    data["grammar"] = body.value<std::string>("grammar", "");               // :793
    if (const Json &rf = body["response_format"]; rf.is_object() && (rf["type"].str_or("") == "json_object" || rf["type"].str_or("") == "json_schema")) {   // :794-801
        std::string gbnf, gerr;
        if (!json_schema_to_gbnf(rf["json_schema"]["schema"], gbnf, gerr)) {
            cb(make_status(false, true, false, k400BadRequest), message("response_format: " + gerr));
            return;
        }
        data["grammar"] = gbnf;
    }
    if (!si->grammar_file_content.empty()) data["grammar"] = si->grammar_file_content;   // :812-814
    if (const std::string &gtext = data["grammar"].as_string(); !gtext.empty()) {
        // a grammar that does not parse is answered here, with the parser's message (from inside the slot loop a non-streaming request would only
        // learn "Internal error during inference", :1094-1096)
        std::string gerr;
        if (!Grammar::parse(gtext, gerr)) {
            cb(make_status(false, true, false, k400BadRequest), message("grammar: " + gerr));
            return;
        }
    }
    if (body["logit_bias"].is_array()) data["logit_bias"] = body["logit_bias"];
    else if (body["logit_bias"].is_object()) {    // {"token": bias} -> [[token, bias]] (chat_completion_request.h:140-160)
        Json arr = Json::array();
        for (const auto &kv : body["logit_bias"].members()) { Json e = Json::array(); e.push_back((int64_t)atoll(kv.first.c_str())); e.push_back(kv.second); arr.push_back(e); }
        data["logit_bias"] = arr;
    }
    // prompt = pre_prompt + sum(role_prefix + content) + ai_prompt (:816-852); no Jinja template in the reference
    std::string formatted = si->pre_prompt;
    auto get_message = [](const Json &c) -> std::string {
        if (c.is_array()) { for (const Json &mc : c.items()) if (mc["type"].as_string() == "text") return mc["text"].as_string(); return ""; }
        return c.as_string();
    };
    // `image_url` content pieces (src/llama_engine.cc:854-900): on a multimodal context every piece becomes a placeholder [img-N] in the prompt and an entry of
    // image_data - the base64 payload of a data: URL, or the bytes of a local file; remote URLs are not fetched (the reference does not either).  A context
    // without a projector answers such a request with an error instead of completing it as text
    const bool multimodal = si->backend && si->backend->multimodal();
    bool has_images = false;
    for (const Json &msg : body["messages"].items()) {
        if (!msg["content"].is_array()) continue;
        for (const Json &mc : msg["content"].items())
            if (mc["type"].is_string() && mc["type"].as_string() == "image_url") has_images = true;
    }
    if (has_images && !multimodal) {
        cb(make_status(false, true, false, k400BadRequest), message("image_url content is not supported: the model was loaded without a multimodal projector (mmproj)"));
        return;
    }
    Json image_data = Json::array();
    int n_images = 0;
    std::string image_error;
    // the text of a message; with images: its text pieces and a placeholder per image, in the order given
    auto get_message_mm = [&](const Json &c) -> std::string {
        if (!c.is_array()) return c.as_string();
        std::string out;
        for (const Json &mc : c.items()) {
            const std::string type = mc["type"].str_or("");
            if (type == "text") out += mc["text"].str_or("");
            else if (type == "image_url") {
                const std::string url = mc["image_url"]["url"].str_or("");
                std::string b64;
                const size_t comma = url.find("base64,");
                if (url.rfind("data:image", 0) == 0 && comma != std::string::npos) b64 = url.substr(comma + 7);
                else if (url.rfind("http", 0) == 0) image_error = "remote images are not fetched: send the image as a data: URL";
                else {
                    FILE *f = fopen(url.c_str(), "rb");
                    if (!f) image_error = "local image not found: " + url;
                    else {
                        std::vector<uint8_t> bytes;
                        uint8_t buf[65536];
                        for (size_t n; (n = fread(buf, 1, sizeof buf, f)) > 0;) bytes.insert(bytes.end(), buf, buf + n);
                        fclose(f);
                        b64 = base64_bytes(bytes.data(), bytes.size());
                    }
                }
                Json piece = Json::object();
                piece["data"] = b64; piece["id"] = n_images;
                image_data.push_back(piece);
                out += "[img-" + std::to_string(n_images) + "]";
                n_images++;
            }
        }
        return out;
    };
    if (body["prompt"].is_string() && !body["prompt"].as_string().empty()) {
        formatted = body["prompt"].as_string();
    } else {
        for (const Json &msg : body["messages"].items()) {
            const std::string in_role = msg["role"].as_string();
            const std::string role = in_role == "user" ? si->user_prompt : in_role == "assistant" ? si->ai_prompt : in_role == "system" ? si->system_prompt : in_role;
            const std::string content = has_images ? get_message_mm(msg["content"]) : get_message(msg["content"]);
            if (!content.empty()) formatted += role + content;
        }
        formatted += si->ai_prompt;
    }
    if (!image_error.empty()) { cb(make_status(false, true, false, k400BadRequest), message(image_error)); return; }
    if (n_images > 0) data["image_data"] = image_data;
    data["prompt"] = formatted;
    Json stop = Json::array();
    const Json &req_stop = (body["stop"].is_array() && body["stop"].size() > 0) ? body["stop"] : si->stop_words;
    for (const Json &w : req_stop.items()) if (w.is_string()) stop.push_back(w);
    stop.push_back("<|im_end|>");                 // :922-929
    stop.push_back(rtrim(si->user_prompt));
    data["stop"] = stop;
    const int n = std::max(1, body.value<int>("n", 1));

    if (stream) {   // :939-1043
        si->q->run([this, si, cb, data, n_probs, include_usage, model_id]() mutable {
            LlamaServerContext &llama = *si->ctx;
            const int task_id = llama.RequestCompletion(data, false, false, -1);
            bool first = true;
            bool terminal = false;                  // the stream has had its last callback (is_done, or the error chunk)
            while (llama.model_loaded_external) {
                {
                    std::lock_guard<std::mutex> lk(stop_mutex_);
                    if (force_stop_.erase(model_id)) { llama.RequestCancel(task_id); break; }
                }
                TaskResult result = llama.NextResult(task_id);
                if (!result.error) {
                    std::string to_send = result.result_json["content"].as_string();
                    if (first) { ltrim(to_send); first = false; }     // trim the leading space of the first token
                    Json logprobs = n_probs > 0 ? result.result_json["completion_probabilities"] : Json();
                    Json resp = Json::object();
                    resp["data"] = "data: " + chunk_json(to_send, Json(""), include_usage, nullptr, logprobs) + "\n\n";
                    cb(make_status(false, false, true, k200OK), std::move(resp));
                    if (result.stop) {
                        llama.RequestCancel(task_id);
                        Json usage;
                        if (include_usage) {
                            usage = Json::object();
                            const int pt = (int)result.result_json["tokens_evaluated"].as_int(), ct = (int)result.result_json["tokens_predicted"].as_int();
                            Json details = Json::object();
                            details["reasoning_tokens"] = 0;
                            usage["prompt_tokens"] = pt; usage["completion_tokens"] = ct; usage["total_tokens"] = pt + ct;
                            usage["completion_tokens_details"] = details;
                        }
                        Json last = Json::object();
                        last["data"] = "data: " + chunk_json("", Json("stop"), include_usage, include_usage ? &usage : nullptr, Json()) + "\n\n" + "data: [DONE]" + "\n\n";
                        cb(make_status(true, false, true, k200OK), std::move(last));
                        terminal = true;
                        break;
                    }
                } else {   // :1017-1024
                    llama.RequestCancel(task_id);
                    Json resp = Json::object();
                    resp["data"] = std::string();
                    cb(make_status(false, true, true, k200OK), std::move(resp));
                    terminal = true;
                    break;
                }
            }
            // :1028-1041 - the loop ended because the model was unloaded under the stream (between two results, or before this task ever ran): the
            // provider is still waiting, so it gets the error chunk {data: "", has_error, is_stream} as its last callback.  (A stream that already had its
            // terminal callback gets nothing more: the reference would call back once more behind is_done when an unload follows a finished stream at once.)
            if (!terminal && !llama.model_loaded_external) {
                log_line(LOG_WARN, "Model unloaded during inference");
                Json resp = Json::object();
                resp["data"] = std::string();
                cb(make_status(false, true, true, k200OK), std::move(resp));
            }
        });
    } else {   // :1044-1112
        si->q->run([si, cb, data, n, n_probs]() mutable {
            LlamaServerContext &llama = *si->ctx;
            std::vector<int> ids;
            for (int i = 0; i < n; i++) ids.push_back(llama.RequestCompletion(data, false, false, -1));
            Json resp;
            bool has_error = false;
            int prompt_tokens = 0, predicted = 0, index = 0;
            for (int id : ids) {
                TaskResult r = llama.NextResult(id);
                if (!r.error && r.stop) {
                    prompt_tokens += (int)r.result_json["tokens_evaluated"].as_int();
                    predicted += (int)r.result_json["tokens_predicted"].as_int();
                    std::string to_send = r.result_json["content"].as_string();
                    ltrim(to_send);
                    Json logprobs = n_probs > 0 ? r.result_json["completion_probabilities"] : Json();
                    Json one = full_json(to_send, prompt_tokens, predicted, logprobs);
                    if (resp.is_null()) resp = one;
                    else { Json choice = one["choices"].at(0); choice["index"] = index; resp["choices"].push_back(choice); resp["usage"] = one["usage"]; }
                    index++;
                } else {
                    has_error = true;
                    resp = message("Internal error during inference");
                    break;
                }
            }
            cb(make_status(true, has_error, false, k200OK), std::move(resp));
        });
    }
}

}  // namespace mi355
