// server_context.cc — see server_context.h.  Behavioural mirror of the reference loop; citations are into
// /root/reference/src/llama_server_context.cc unless noted.
#include "server_context.h"
#include "grammar.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>

namespace mi355 {

// Minimal fork-join pool: run(n, f) calls f(0..n-1) on the workers plus the caller and returns when all are done.
struct LlamaServerContext::SamplePool {
    explicit SamplePool(int n_threads) {
        for (int i = 0; i < n_threads; i++) workers.emplace_back([this] { loop(); });
    }
    ~SamplePool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
    void run(int n, const std::function<void(int)> &f) {
        if (n <= 1 || workers.empty()) { for (int i = 0; i < n; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> lk(m);
            fn = &f; total = n; next = 0; done = 0; gen++;
        }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return done == total; });
        fn = nullptr;
    }
    void work() {
        for (;;) {
            int i;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> lk(m);
                if (!fn || next >= total) return;
                i = next++; f = fn;
            }
            (*f)(i);
            {
                std::lock_guard<std::mutex> lk(m);
                if (++done == total) cv_done.notify_all();
            }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            work();
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv, cv_done;
    const std::function<void(int)> *fn = nullptr;
    int total = 0, next = 0, done = 0;
    uint64_t gen = 0;
    bool stop = false;
};

int64_t time_us() {
    using namespace std::chrono;
    return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- slot (llama_client_slot.cc)
void LlamaClientSlot::Reset() {           // :3-27
    num_prompt_tokens = 0;
    generated_text.clear();
    truncated = false; stopped_eos = false; stopped_word = false; stopped_limit = false;
    stopping_word.clear();
    n_past = 0;
    sent_count = 0;
    sent_token_probs_index = 0;
    generated_token_probs.clear();
    prompt_ready = false;
    i_batch = -1;
    images.clear(); input_suffix.clear(); next_image = 0;
}

bool LlamaClientSlot::HasBudget(const ServerParams &global) {   // :29-37
    n_remaining = -1;
    if (params.n_predict != -1) n_remaining = params.n_predict - n_decoded;
    else if (global.n_predict != -1) n_remaining = global.n_predict - n_decoded;
    return n_remaining > 0 || n_remaining == -1;
}

void LlamaClientSlot::Release() {          // :55-60
    if (state == SlotState::kIdle || state == SlotState::kProcessing) {
        t_token_generation = (double)(time_us() - t_start_genereration) / 1e3;
        command = SlotCommand::kRelease;
    }
}

Json LlamaClientSlot::GetFormatedTimings() const {   // :62-76
    Json t = Json::object();
    t["prompt_n"] = num_prompt_tokens_processed;
    t["prompt_ms"] = t_prompt_processing;
    t["prompt_per_token_ms"] = t_prompt_processing / std::max(num_prompt_tokens_processed, 1);
    t["prompt_per_second"] = 1e3 / std::max(t_prompt_processing, 1e-9) * num_prompt_tokens_processed;
    t["predicted_n"] = n_decoded;
    t["predicted_ms"] = t_token_generation;
    t["predicted_per_token_ms"] = t_token_generation / std::max(n_decoded, 1);
    t["predicted_per_second"] = 1e3 / std::max(t_token_generation, 1e-9) * n_decoded;
    return t;
}

// ---------------------------------------------------------------- context
LlamaServerContext::LlamaServerContext(IBackend *be, const ServerParams &p) : params(p), be_(be) {}

LlamaServerContext::~LlamaServerContext() { ReleaseResources(); }

void LlamaServerContext::Initialize() {    // :244-282
    id_gen_ = 0;
    n_ctx = be_->n_ctx();
    const int n_ctx_slot = n_ctx / std::max(params.n_parallel, 1);
    slots.clear();
    slots.resize((size_t)std::max(params.n_parallel, 1));
    for (int i = 0; i < (int)slots.size(); i++) {
        slots[(size_t)i].id = i;
        slots[(size_t)i].n_ctx = n_ctx_slot;
        slots[(size_t)i].Reset();
    }
    b_token_.assign((size_t)n_ctx, 0); b_pos_.assign((size_t)n_ctx, 0); b_seq_.assign((size_t)n_ctx, 0); b_logits_.assign((size_t)n_ctx, 0);
    model_loaded_external = true;
    bgr_thread_ = std::thread(&LlamaServerContext::DoBackgroundTasks, this);
}

void LlamaServerContext::ReleaseResources() {   // :366-380
    if (getenv("MI355_LOOP_TIMING") && n_ticks_ > 0)
        fprintf(stderr, "[loop] %ld decode calls, %.1f tokens each: decode+logits %.3f ms, sampling %.3f ms, post-processing %.3f ms per call\n",
                n_ticks_, (double)n_tick_tokens_ / (double)n_ticks_, t_decode_us_ / 1e3 / (double)n_ticks_, t_sample_us_ / 1e3 / (double)n_ticks_,
                t_post_us_ / 1e3 / (double)n_ticks_);
    // The flag is part of both waits' predicates: it changes under the waiters' mutexes.  Flipped outside them, the loop thread could read it (true) in its
    // predicate, lose the notification sent before it blocks, and sleep for ever - the join below with it (seen as a rare hang of the host-logic program
    // under load); the same for a caller inside NextResult.
    bool was;
    { std::lock_guard<std::mutex> lk(mutex_tasks_); was = model_loaded_external.exchange(false); }
    if (was) {
        condition_tasks_.notify_all();
        if (bgr_thread_.joinable()) bgr_thread_.join();
        { std::lock_guard<std::mutex> lk(mutex_results_); }
        condition_results_.notify_all();
    }
}

void LlamaServerContext::DoBackgroundTasks() {  // :1239-1246
    while (model_loaded_external) UpdateSlots();
    KvCacheClear();
}

void LlamaServerContext::KvCacheClear() { be_->kv_clear(); clean_kv_cache = false; }

int LlamaServerContext::RequestCompletion(Json data, bool, bool embedding, int) {   // :295-323
    std::unique_lock<std::mutex> lock(mutex_tasks_);
    Task t{id_gen_++, -1, false, std::move(data), embedding};
    const int id = t.id;
    queue_tasks_.push_back(std::move(t));
    condition_tasks_.notify_one();
    return id;
}

TaskResult LlamaServerContext::NextResult(int task_id) {   // :325-352
    std::unique_lock<std::mutex> lock(mutex_results_);
    auto mine = [&]() -> long {
        for (size_t i = 0; i < queue_results_.size(); i++) if (queue_results_[i].id == task_id) return (long)i;
        return -1;
    };
    condition_results_.wait(lock, [&] { return mine() >= 0 || !model_loaded_external; });
    const long i = mine();
    if (i >= 0) {
        TaskResult r = std::move(queue_results_[(size_t)i]);
        queue_results_.erase(queue_results_.begin() + i);
        return r;
    }
    TaskResult r;
    r.id = task_id; r.error = true; r.stop = true;
    r.result_json = Json::object();
    r.result_json["content"] = "model unloaded";
    return r;
}

void LlamaServerContext::RequestCancel(int task_id) {   // :354-364
    std::unique_lock<std::mutex> lock(mutex_tasks_);
    queue_tasks_.push_back(Task{id_gen_++, task_id, true, Json()});
    condition_tasks_.notify_one();
}

std::vector<int32_t> LlamaServerContext::Tokenize(const Json &json_prompt, bool add_bos, bool parse_special) const {   // :382-414
    std::vector<int32_t> out;
    const Vocab &v = be_->vocab();
    if (json_prompt.is_array()) {
        bool first = true;
        for (const Json &p : json_prompt.items()) {
            if (p.is_string()) {
                auto t = v.tokenize(p.as_string(), first && add_bos, parse_special);
                out.insert(out.end(), t.begin(), t.end());
                first = false;
            } else {
                if (first) first = false;
                out.push_back((int32_t)p.as_int());
            }
        }
    } else {
        out = v.tokenize(json_prompt.as_string(), add_bos, parse_special);
    }
    return out;
}

LlamaClientSlot *LlamaServerContext::GetSlot(int id) {   // :416-432 (LRU)
    int64_t t_last = time_us();
    LlamaClientSlot *last_used = nullptr;
    for (auto &slot : slots) {
        if (slot.id == id && slot.Available()) return &slot;
        if (slot.Available() && slot.t_last_used < t_last) { last_used = &slot; t_last = slot.t_last_used; }
    }
    return last_used;
}

// standard base64 (RFC 4648; '=' padding optional, whitespace skipped); false on any other character
static bool base64_decode(const std::string &in, std::vector<uint8_t> &out) {
    out.clear();
    out.reserve(in.size() * 3 / 4);
    uint32_t acc = 0;
    int bits = 0;
    for (const char ch : in) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+' || ch == '-') v = 62;
        else if (ch == '/' || ch == '_') v = 63;
        else if (ch == '=') break;
        else if (ch == '\n' || ch == '\r' || ch == ' ' || ch == '\t') continue;
        else return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}

bool LlamaServerContext::LaunchSlotWithData(LlamaClientSlot *&slot, const Json &data) {   // :434-641
    SlotParams dp;
    const SamplingParams &ds = params.sampling;
    if (data.contains("__oaicompat")) { slot->oaicompat = true; slot->oaicompat_model = data.value<std::string>("model", "gpt-3.5-turbo-0613"); }
    else { slot->oaicompat = false; slot->oaicompat_model.clear(); }
    slot->params.stream = data.value<bool>("stream", false);
    slot->params.cache_prompt = data.value<bool>("cache_prompt", false);
    slot->params.n_predict = data.value<int>("n_predict", dp.n_predict);
    SamplingParams &sp = slot->sparams;
    sp = ds;
    sp.top_k = data.value<int>("top_k", ds.top_k);
    sp.top_p = data.value<float>("top_p", ds.top_p);
    sp.min_p = data.value<float>("min_p", ds.min_p);
    sp.typ_p = data.value<float>("typical_p", ds.typ_p);
    sp.temp = data.value<float>("temperature", ds.temp);
    sp.penalty_last_n = data.value<int>("repeat_last_n", ds.penalty_last_n);
    sp.penalty_n_ctx = slot->n_ctx;                            // what repeat_last_n = -1 means (upstream: the context size)
    sp.penalty_repeat = data.value<float>("repeat_penalty", ds.penalty_repeat);
    sp.penalty_freq = data.value<float>("frequency_penalty", ds.penalty_freq);
    sp.penalty_present = data.value<float>("presence_penalty", ds.penalty_present);
    sp.mirostat = data.value<int>("mirostat", ds.mirostat);
    sp.mirostat_tau = data.value<float>("mirostat_tau", ds.mirostat_tau);
    sp.mirostat_eta = data.value<float>("mirostat_eta", ds.mirostat_eta);
    slot->params.n_keep = data.value<int>("n_keep", slot->params.n_keep);
    sp.seed = (uint32_t)data.value<int64_t>("seed", (int64_t)(int32_t)ds.seed);
    slot->params.seed = sp.seed;
    sp.n_probs = data.value<int>("n_probs", ds.n_probs);
    sp.min_keep = data.value<int>("min_keep", ds.min_keep);
    sp.dynatemp_range = data.value<float>("dynatemp_range", ds.dynatemp_range);
    sp.dynatemp_exponent = data.value<float>("dynatemp_exponent", ds.dynatemp_exponent);
    sp.ignore_eos = data.value<bool>("ignore_eos", ds.ignore_eos);

    slot->prompt_tokens.clear();
    if (const Json *pt = data.find("prompt_tokens"); pt && pt->is_array())
        for (const Json &t : pt->items()) slot->prompt_tokens.push_back((int32_t)t.as_int());
    slot->num_prompt_tokens = (int32_t)slot->prompt_tokens.size();
    slot->prompt = data.contains("prompt") ? data["prompt"] : Json("");
    {   // token ids straight from the request: one bad id must fail THIS request, not the decode of every slot in the tick
        const int n_vocab = be_->n_vocab();
        auto bad_id = [&](int64_t t) { return t < 0 || t >= n_vocab; };
        auto refuse = [&](const std::string &why) { launch_error_ = why; return false; };
        for (int32_t t : slot->prompt_tokens) if (bad_id(t)) return refuse("token id " + std::to_string(t) + " out of range for this model's vocabulary");
        if (slot->prompt.is_array())
            for (const Json &p : slot->prompt.items()) {
                if (p.is_string()) continue;
                if (!p.is_int()) return refuse("prompt array elements must be strings or token ids");
                if (bad_id(p.as_int())) return refuse("token id " + std::to_string(p.as_int()) + " out of range for this model's vocabulary");
            }
    }

    // image_data of a multimodal request (:557-623): [{"data": base64, "id": n}], each named in the prompt by a placeholder [img-n].  The prompt is cut at the
    // placeholders into the text in front of every image and the text behind the last one; an image that cannot be decoded, or a placeholder without an
    // image, fails the request here
    slot->images.clear(); slot->input_suffix.clear(); slot->next_image = 0;
    if (be_->multimodal()) {
        if (const Json *imgs = data.find("image_data"); imgs && imgs->is_array() && imgs->size() > 0) {
            for (const Json &ij : imgs->items()) {
                SlotImage im;
                im.id = ij.contains("id") ? (int)ij["id"].as_int() : (int)slot->images.size();
                if (!base64_decode(ij["data"].str_or(""), im.bytes) || im.bytes.empty()) { launch_error_ = "image [id: " + std::to_string(im.id) + "]: not base64 data"; return false; }
                std::string why;
                if (!be_->image_check(im.bytes.data(), im.bytes.size(), why)) { launch_error_ = "failed to load image [id: " + std::to_string(im.id) + "]: " + why; return false; }
                slot->images.push_back(std::move(im));
            }
            if (!slot->prompt.is_string()) { launch_error_ = "a request with images takes its prompt as one string"; return false; }
            const std::string text = slot->prompt.as_string();
            std::vector<SlotImage> ordered;
            size_t from = 0, at = 0;
            while ((at = text.find("[img-", at)) != std::string::npos) {
                const size_t close = text.find(']', at + 5);
                if (close == std::string::npos) break;
                const std::string num = text.substr(at + 5, close - at - 5);
                char *endp = nullptr;
                const long want = strtol(num.c_str(), &endp, 10);
                if (num.empty() || *endp) { launch_error_ = "invalid image number id in prompt: " + num; return false; }
                auto it = std::find_if(slot->images.begin(), slot->images.end(), [&](const SlotImage &m) { return m.id == (int)want && m.prefix_prompt.empty() && !m.bytes.empty(); });
                if (it == slot->images.end()) { launch_error_ = "image with id " + num + " not found"; return false; }
                SlotImage taken = std::move(*it);
                it->bytes.clear();                                   // (an id names ONE image: a second placeholder with it finds nothing)
                taken.prefix_prompt = text.substr(from, at - from);
                ordered.push_back(std::move(taken));
                from = close + 1;
                at = close + 1;
            }
            if (ordered.empty()) { launch_error_ = "the prompt names none of the request's images ([img-N])"; return false; }
            slot->images = std::move(ordered);
            slot->input_suffix = text.substr(from);
            slot->prompt = Json("");
            slot->params.cache_prompt = false;                       // (:620: no prompt cache for multimodal requests)
        }
    }

    sp.logit_bias.clear();
    const Vocab &vocab = be_->vocab();
    if (sp.ignore_eos && vocab.eos() >= 0) sp.logit_bias.push_back({vocab.eos(), -INFINITY});
    if (const Json *lb = data.find("logit_bias"); lb && lb->is_array()) {
        const int n_vocab = be_->n_vocab();
        for (const Json &el : lb->items()) {
            if (!el.is_array() || el.size() != 2) continue;
            float bias;
            if (el.at(1).is_number()) bias = (float)el.at(1).as_double();
            else if (el.at(1).is_bool() && !el.at(1).as_bool()) bias = -INFINITY;
            else continue;
            if (el.at(0).is_int()) {
                const int tok = (int)el.at(0).as_int();
                if (tok >= 0 && tok < n_vocab) sp.logit_bias.push_back({tok, bias});
            } else if (el.at(0).is_string()) {
                for (int32_t tok : vocab.tokenize(el.at(0).as_string(), false)) sp.logit_bias.push_back({tok, bias});
            }
        }
    }
    slot->params.antiprompt.clear();
    if (const Json *stop = data.find("stop"); stop && stop->is_array())
        for (const Json &w : stop->items()) if (w.is_string() && !w.as_string().empty()) slot->params.antiprompt.push_back(w.as_string());

    slot->smpl.reset(new Sampler(sp));
    // `grammar` (src/llama_server_context.cc:473): GBNF text; a text that does not parse fails this request with the parser's message
    if (const std::string gtext = data.value<std::string>("grammar", ""); !gtext.empty()) {
        std::string gerr;
        std::shared_ptr<const Grammar> g = Grammar::parse(gtext, gerr);
        if (!g) { launch_error_ = "grammar: " + gerr; return false; }
        if (grammar_pieces_.empty()) {
            const int n_vocab = be_->n_vocab();
            grammar_pieces_.resize((size_t)n_vocab);
            grammar_eog_.assign((size_t)n_vocab, 0);
            for (int t = 0; t < n_vocab; t++) {
                if (vocab.is_eog(t)) grammar_eog_[(size_t)t] = 1;
                else if (!vocab.is_control(t)) grammar_pieces_[(size_t)t] = vocab.token_to_piece(t, false);
            }
        }
        slot->smpl->set_grammar(std::move(g), &grammar_pieces_, &grammar_eog_);
    }
    slot->command = SlotCommand::kLoadPrompt;
    all_slots_are_idle = false;
    return true;
}

void LlamaServerContext::SendEmbedding(LlamaClientSlot &slot, int batch_index) {   // :1026-1070
    TaskResult res;
    res.id = slot.task_id; res.error = false; res.stop = true;
    const int n_embd = be_->n_embd();
    std::vector<float> embd_res((size_t)n_embd, 0.0f);
    // llama_get_embeddings_seq where the model pools (mean over the prompt's tokens / its first token; "last" is the flagged row itself), else
    // llama_get_embeddings_ith of the last token (:1041-1044).  A pooled prompt had every token flagged when it was queued (UpdateSlots).
    const int pool = be_->pooling_type();
    const int n_rows = (pool == 1 || pool == 2) ? slot.num_prompt_tokens_processed : 1;
    std::vector<float> pooled;
    const float *embd = nullptr;
    if ((pool == 1 || pool == 2) && (n_rows <= 0 || batch_index - n_rows + 1 < 0)) {
        // the prompt's rows are not all in THIS batch (it was ingested over several scheduler ticks): pooling over the rows at hand would silently
        // answer with another vector - refuse instead
        SendError(slot, "embedding: a pooled prompt must fit one batch (raise n_batch or shorten the input)");
        return;
    }
    if (pool == 1 && n_rows > 0 && batch_index - n_rows + 1 >= 0) {
        pooled.assign((size_t)n_embd, 0.0f);
        bool ok = true;
        for (int r = 0; r < n_rows && ok; r++) {
            const float *e = be_->embeddings_ith(batch_index - n_rows + 1 + r);
            if (!e) { ok = false; break; }
            for (int i = 0; i < n_embd; i++) pooled[(size_t)i] += e[i];      // token order, f32: the order of ggml's mean pooling mat-mul over the batch rows
        }
        if (ok) {
            const float inv = 1.0f / (float)n_rows;
            for (int i = 0; i < n_embd; i++) pooled[(size_t)i] *= inv;
            embd = pooled.data();
        }
    } else if (pool == 2 && batch_index - n_rows + 1 >= 0) {
        embd = be_->embeddings_ith(batch_index - n_rows + 1);
    } else {
        embd = be_->embeddings_ith(batch_index);
    }
    if (embd) {                                          // common_embd_normalize(embd, out, n, 2): Euclidean norm in double
        double sum = 0.0;
        for (int i = 0; i < n_embd; i++) sum += (double)embd[i] * (double)embd[i];
        sum = std::sqrt(sum);
        const float norm = sum > 0.0 ? (float)(1.0 / sum) : 0.0f;
        for (int i = 0; i < n_embd; i++) embd_res[(size_t)i] = embd[i] * norm;
    }
    Json arr = Json::array();
    for (float v : embd_res) arr.push_back((double)v);
    res.result_json = Json::object();
    res.result_json["tokens_evaluated"] = slot.num_prompt_tokens;
    res.result_json["embedding"] = arr;
    {
        std::lock_guard<std::mutex> lock(mutex_results_);
        queue_results_.push_back(std::move(res));
    }
    condition_results_.notify_all();
}

void LlamaServerContext::ProcessTasks() {   // :1152-1237
    std::unique_lock<std::mutex> lock(mutex_tasks_);
    std::deque<Task> deferred;
    while (!queue_tasks_.empty()) {
        Task task = std::move(queue_tasks_.front());
        queue_tasks_.pop_front();
        if (task.cancel) {
            for (auto &slot : slots) if (slot.task_id == task.target_id) { slot.Release(); break; }
            // a cancelled task that never got a slot is dropped from the queue
            deferred.erase(std::remove_if(deferred.begin(), deferred.end(), [&](const Task &t) { return t.id == task.target_id; }), deferred.end());
            continue;
        }
        LlamaClientSlot *slot = GetSlot((int)task.data.value<int>("slot_id", -1));
        if (!slot) { deferred.push_back(std::move(task)); continue; }     // no free slot: retry on the next tick
        slot->Reset();
        slot->task_id = task.id;
        slot->embedding = task.embedding_mode;       // :1194
        if (!LaunchSlotWithData(slot, task.data)) {
            slot->Release();
            SendError(*slot, "Invalid request: " + (launch_error_.empty() ? std::string("the slot could not take it") : launch_error_));
            launch_error_.clear();
        }
    }
    for (auto &t : deferred) queue_tasks_.push_back(std::move(t));
}

// ---- text leaving a slot.  Behaviour of the reference's process_token / find_stopping_strings (:682-813) - a stop string never leaves, text that may still
// become one is held back, a character cut by a token boundary is held back whole - written as three questions about the UNSENT tail of the generated text.
namespace {

// bytes at the end of `s` that open a UTF-8 sequence whose continuation bytes have not all arrived (0: the text ends on a character boundary)
size_t utf8_pending_bytes(const std::string &s) {
    const size_t n = s.size();
    for (size_t back = 1; back <= 4 && back <= n; back++) {
        const unsigned char c = (unsigned char)s[n - back];
        if ((c & 0xC0) == 0x80) continue;                      // a continuation byte: its lead byte is further back
        const size_t need = (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : (c & 0xF8) == 0xF0 ? 4 : 1;
        return need > back ? back : 0;
    }
    return 0;
}

struct StopHit { size_t at = std::string::npos; const std::string *word = nullptr; };

// earliest start of a COMPLETE stop string in `tail`, looking only where the newest piece (its last `fresh` bytes) can have completed one; the first
// word of the list wins a tie
StopHit first_complete_stop(const std::string &tail, size_t fresh, const std::vector<std::string> &stops) {
    StopHit hit;
    for (const std::string &w : stops) {
        const size_t window = w.size() + fresh;
        const size_t at = tail.find(w, tail.size() > window ? tail.size() - window : 0);
        if (at != std::string::npos && at < hit.at) { hit.at = at; hit.word = &w; }
    }
    return hit;
}

// earliest position from which the END of `tail` is the beginning of some stop string (the longest such beginning per word); npos: none
size_t first_open_stop(const std::string &tail, const std::vector<std::string> &stops) {
    size_t best = std::string::npos;
    for (const std::string &w : stops)
        for (size_t len = std::min(w.size(), tail.size()); len >= 1; len--)
            if (tail.compare(tail.size() - len, len, w, 0, len) == 0) { best = std::min(best, tail.size() - len); break; }
    return best;
}

}  // namespace

size_t LlamaServerContext::FindStoppingStrings(const std::string &text, size_t last_token_size, bool full, LlamaClientSlot &slot) {   // (kept for callers of the reference's name)
    if (!full) return first_open_stop(text, slot.params.antiprompt);
    const StopHit hit = first_complete_stop(text, last_token_size, slot.params.antiprompt);
    if (hit.word) { slot.stopped_word = true; slot.stopping_word = *hit.word; slot.has_next_token = false; }
    return hit.at;
}

bool LlamaServerContext::ProcessToken(CompletionTokenOutput &result, LlamaClientSlot &slot) {   // :716-813
    const std::string piece = be_->vocab().token_to_piece(result.tok, true);
    slot.sampled = result.tok;
    slot.generated_text += piece;
    slot.has_next_token = true;
    if (utf8_pending_bytes(slot.generated_text) == 0) {         // (else: nothing is searched, sent or counted until the character is whole)
        const size_t sent = std::min(slot.sent_count, slot.generated_text.size());
        const std::string tail = slot.generated_text.substr(sent);
        const StopHit hit = first_complete_stop(tail, piece.size(), slot.params.antiprompt);
        if (hit.word) {
            // the stop string and everything behind it are dropped; what stood in front of it is not sent as a partial any more (the final response of a
            // non-streaming request carries the whole text)
            slot.stopped_word = true; slot.stopping_word = *hit.word; slot.has_next_token = false;
            slot.generated_text.resize(sent + hit.at);
        } else if (first_open_stop(tail, slot.params.antiprompt) == std::string::npos) {
            result.text_to_send = tail;
            slot.sent_count += tail.size();
        }
        slot.AddTokenString(result);
        if (slot.params.stream) SendPartialResponse(slot, result);
        else slot.sent_token_probs_index++;
    }
    if (slot.n_decoded > 2 && slot.has_next_token && !slot.HasBudget(params)) { slot.stopped_limit = true; slot.has_next_token = false; }
    if (be_->vocab().is_eog(result.tok)) { slot.stopped_eos = true; slot.has_next_token = false; }
    return slot.has_next_token;
}

Json LlamaServerContext::ProbsToJson(const std::vector<CompletionTokenOutput> &probs) const {
    Json out = Json::array();
    for (const auto &p : probs) {
        Json pj = Json::array();
        for (const auto &tp : p.probs) {
            Json e = Json::object();
            e["tok_str"] = be_->vocab().token_to_piece(tp.tok, true);
            e["prob"] = tp.p;
            pj.push_back(std::move(e));
        }
        Json e = Json::object();
        e["content"] = be_->vocab().token_to_piece(p.tok, true);
        e["probs"] = std::move(pj);
        out.push_back(std::move(e));
    }
    return out;
}

Json LlamaServerContext::GetFormatedGeneration(const LlamaClientSlot &slot) const {   // :878-918
    Json g = Json::object();
    g["n_ctx"] = slot.n_ctx; g["model"] = params.model_alias; g["seed"] = (int64_t)slot.sparams.seed;
    g["temperature"] = slot.sparams.temp; g["dynatemp_range"] = slot.sparams.dynatemp_range; g["dynatemp_exponent"] = slot.sparams.dynatemp_exponent;
    g["top_k"] = slot.sparams.top_k; g["top_p"] = slot.sparams.top_p; g["min_p"] = slot.sparams.min_p; g["typical_p"] = slot.sparams.typ_p;
    g["repeat_last_n"] = slot.sparams.penalty_last_n; g["repeat_penalty"] = slot.sparams.penalty_repeat;
    g["presence_penalty"] = slot.sparams.penalty_present; g["frequency_penalty"] = slot.sparams.penalty_freq;
    g["mirostat"] = slot.sparams.mirostat; g["mirostat_tau"] = slot.sparams.mirostat_tau; g["mirostat_eta"] = slot.sparams.mirostat_eta;
    g["stop"] = Json::array_of(slot.params.antiprompt);
    g["n_predict"] = slot.params.n_predict; g["n_keep"] = params.n_keep; g["ignore_eos"] = slot.sparams.ignore_eos;
    g["stream"] = slot.params.stream; g["n_probs"] = slot.sparams.n_probs; g["min_keep"] = slot.sparams.min_keep;
    return g;
}

// a result leaves for the waiting caller (NextResult)
void LlamaServerContext::PostResult(TaskResult &&res) {
    { std::lock_guard<std::mutex> lock(mutex_results_); queue_results_.push_back(std::move(res)); }
    condition_results_.notify_all();
}

// the token probabilities [from, to) of a slot as the response carries them
Json LlamaServerContext::ProbsSlice(const LlamaClientSlot &slot, size_t from, size_t to) const {
    const size_t n = slot.generated_token_probs.size();
    from = std::min(from, n); to = std::min(to, n);
    std::vector<CompletionTokenOutput> part;
    if (from < to) part.assign(slot.generated_token_probs.begin() + (long)from, slot.generated_token_probs.begin() + (long)to);
    return ProbsToJson(part);
}

void LlamaServerContext::SendPartialResponse(LlamaClientSlot &slot, const CompletionTokenOutput &tkn) {   // :920-962
    TaskResult res;
    res.id = slot.task_id; res.error = false; res.stop = false;
    Json &j = res.result_json;
    j = Json::object();
    j["content"] = tkn.text_to_send; j["stop"] = false; j["slot_id"] = slot.id; j["multimodal"] = false;
    if (slot.sparams.n_probs > 0) {                         // the probabilities of as many tokens as the text just released re-tokenises to
        const size_t from = std::min(slot.sent_token_probs_index, slot.generated_token_probs.size());
        const size_t to = std::min(slot.sent_token_probs_index + be_->vocab().tokenize(tkn.text_to_send, false).size(), slot.generated_token_probs.size());
        j["completion_probabilities"] = ProbsSlice(slot, from, to);
        slot.sent_token_probs_index = to;
    }
    if (slot.oaicompat) { j["oaicompat_token_ctr"] = slot.n_decoded; j["model"] = slot.oaicompat_model; }
    PostResult(std::move(res));
}

void LlamaServerContext::SendFinalResponse(LlamaClientSlot &slot) {   // :964-1024
    TaskResult res;
    res.id = slot.task_id; res.error = false; res.stop = true;
    Json &j = res.result_json;
    j = Json::object();
    const bool whole = !slot.params.stream;                  // a streaming request has had its text already
    j["content"] = whole ? slot.generated_text : "";
    j["slot_id"] = slot.id; j["stop"] = true; j["model"] = params.model_alias;
    j["tokens_predicted"] = slot.n_decoded; j["tokens_evaluated"] = slot.num_prompt_tokens;
    j["generation_settings"] = GetFormatedGeneration(slot);
    j["prompt"] = slot.prompt; j["truncated"] = slot.truncated;
    j["stopped_eos"] = slot.stopped_eos; j["stopped_word"] = slot.stopped_word; j["stopped_limit"] = slot.stopped_limit;
    j["stopping_word"] = slot.stopping_word; j["tokens_cached"] = slot.n_past;
    j["timings"] = slot.GetFormatedTimings();
    if (slot.sparams.n_probs > 0) {
        // all tokens but the one that completed a stop string (non-streaming), else the ones whose text has been sent
        const size_t n = slot.generated_token_probs.size();
        const size_t to = (whole && slot.stopped_word && n > 0) ? n - 1 : std::min(slot.sent_token_probs_index, n);
        j["completion_probabilities"] = whole ? ProbsSlice(slot, 0, to) : Json();
    }
    if (slot.oaicompat) { j["oaicompat_token_ctr"] = slot.n_decoded; j["model"] = slot.oaicompat_model; }
    PostResult(std::move(res));
}

void LlamaServerContext::SendError(LlamaClientSlot &slot, const std::string &err) {   // :840-853
    TaskResult res;
    res.id = slot.task_id; res.stop = false; res.error = true;
    res.result_json = Json::object();
    res.result_json["content"] = err;
    PostResult(std::move(res));
}

static size_t common_part(const std::vector<int32_t> &a, const std::vector<int32_t> &b) {
    size_t i = 0;
    while (i < a.size() && i < b.size() && a[i] == b[i]) i++;
    return i;
}

bool LlamaServerContext::UpdateSlots() {   // :1248-1710
    ProcessTasks();
    int n_tokens = 0;       // batch.n_tokens
    if (all_slots_are_idle) {
        if (clean_kv_cache) KvCacheClear();
        std::unique_lock<std::mutex> lock(mutex_tasks_);
        condition_tasks_.wait(lock, [&] { return (!queue_tasks_.empty() && model_loaded_external) || !model_loaded_external; });
        if (!model_loaded_external) return true;
        lock.unlock();
        ProcessTasks();
    }
    // context shift (:1274-1306)
    for (auto &slot : slots) {
        if (slot.IsProcessing() && slot.n_past >= slot.n_ctx) {
            const int n_left = slot.n_past - slot.params.n_keep - 1;
            const int n_discard = n_left / 2;
            be_->kv_seq_rm(slot.id, slot.params.n_keep + 1, slot.params.n_keep + n_discard + 1);
            be_->kv_seq_add(slot.id, slot.params.n_keep + 1 + n_discard, slot.n_past, -n_discard);
            if (slot.params.cache_prompt) {
                for (size_t i = (size_t)(slot.params.n_keep + 1 + n_discard); i < slot.cache_tokens.size(); i++) slot.cache_tokens[i - (size_t)n_discard] = slot.cache_tokens[i];
                slot.cache_tokens.resize(slot.cache_tokens.size() - (size_t)n_discard);
            }
            slot.n_past -= n_discard;
            slot.truncated = true;
        }
    }
    auto batch_add = [&](int32_t tok, int32_t pos, int32_t seq, bool logits) {
        b_token_[(size_t)n_tokens] = tok; b_pos_[(size_t)n_tokens] = pos; b_seq_[(size_t)n_tokens] = seq; b_logits_[(size_t)n_tokens] = logits ? 1 : 0;
        n_tokens++;
    };
    // decode tokens of the ongoing sequences (:1309-1348)
    for (auto &slot : slots) {
        if (slot.command == SlotCommand::kRelease) {
            slot.state = SlotState::kIdle; slot.command = SlotCommand::kNone; slot.t_last_used = time_us();
            continue;
        }
        if (slot.state == SlotState::kIdle) continue;
        slot.i_batch = n_tokens;
        batch_add(slot.sampled, slot.n_past, slot.id, true);
        slot.n_decoded += 1;
        slot.n_past += 1;
        if (slot.params.cache_prompt) slot.cache_tokens.push_back(slot.sampled);
    }
    int32_t n_batch = be_->n_batch();
    bool embd_tick = false, tick_mode_set = n_tokens > 0;      // a tick that already carries generated tokens is a generation tick
    // prompt ingestion (:1355-1621)
    if (params.cont_batching || n_tokens == 0) {
        for (auto &slot : slots) {
            const bool has_prompt = slot.prompt.is_array() || (slot.prompt.is_string() && !slot.prompt.as_string().empty()) || !slot.prompt_tokens.empty() || !slot.images.empty();
            if (slot.state == SlotState::kIdle && slot.command == SlotCommand::kLoadPrompt && !has_prompt) {
                slot.Release();
                SendFinalResponse(slot);
                continue;
            }
            if (!(slot.state == SlotState::kIdle && slot.command == SlotCommand::kLoadPrompt)) continue;
            // an embedding prompt never shares a tick with generation (see the note at the decode call): it waits while
            // any slot generates, and a generation prompt waits for a tick that already took an embedding prompt
            if (!tick_mode_set) { embd_tick = slot.embedding && n_tokens == 0; tick_mode_set = true; }
            if (slot.embedding != embd_tick) continue;
            auto &prompt_tokens = slot.prompt_tokens;
            if (!slot.prompt_ready) {   // first visit of this prompt: tokenise, truncate, find the cached prefix
                slot.t_start_process_prompt = time_us();
                slot.t_start_genereration = 0;
                if (!slot.images.empty()) {
                    // a multimodal prompt (ProcessImages + the layout IngestImages walks, :814-831, 1073-1129): text, an image's embedding rows, text, ..., the
                    // text behind the last image.  The rows are computed here; their positions are held by placeholder ids in prompt_tokens
                    prompt_tokens.clear();
                    bool failed = false;
                    for (size_t k = 0; k < slot.images.size() && !failed; k++) {
                        SlotImage &im = slot.images[k];
                        const std::vector<int32_t> pre = Tokenize(Json(im.prefix_prompt), k == 0 && be_->vocab().add_bos(), false);
                        prompt_tokens.insert(prompt_tokens.end(), pre.begin(), pre.end());
                        std::string why;
                        im.n_rows = be_->image_embed(im.bytes.data(), im.bytes.size(), im.rows, why);
                        if (im.n_rows <= 0) { failed = true; break; }
                        im.pos0 = (int)prompt_tokens.size();
                        prompt_tokens.insert(prompt_tokens.end(), (size_t)im.n_rows, 0);
                        im.bytes.clear(); im.bytes.shrink_to_fit();
                    }
                    const std::vector<int32_t> suf = Tokenize(Json(slot.input_suffix), false, false);
                    prompt_tokens.insert(prompt_tokens.end(), suf.begin(), suf.end());
                    // the rows cannot be cut the way a long text prompt is: the request must fit, and must end in text (the logits come from its last token)
                    if (failed || suf.empty() || (int)prompt_tokens.size() >= slot.n_ctx) {
                        slot.state = SlotState::kProcessing; slot.command = SlotCommand::kNone;
                        slot.Release();
                        SendError(slot, failed ? "Failed processing images" : suf.empty() ? "a prompt with images must end in text" : "the prompt and its images do not fit the context");
                        continue;
                    }
                    slot.next_image = 0;
                }
                if (prompt_tokens.empty()) prompt_tokens = Tokenize(slot.prompt, be_->vocab().add_bos(), true);
                slot.n_past = 0;
                slot.num_prompt_tokens = (int32_t)prompt_tokens.size();
                if (slot.params.n_keep < 0) slot.params.n_keep = slot.num_prompt_tokens;
                slot.params.n_keep = std::min(slot.n_ctx - 4, slot.params.n_keep);
                if (slot.num_prompt_tokens >= slot.n_ctx) {   // truncate by half-blocks (:1452-1485)
                    const int n_left = slot.n_ctx - slot.params.n_keep;
                    const int n_block_size = std::max(n_left / 2, 1);
                    const int erased_blocks = (slot.num_prompt_tokens - slot.params.n_keep - n_block_size) / n_block_size;
                    std::vector<int32_t> nt(prompt_tokens.begin(), prompt_tokens.begin() + slot.params.n_keep);
                    nt.insert(nt.end(), prompt_tokens.begin() + slot.params.n_keep + erased_blocks * n_block_size, prompt_tokens.end());
                    slot.truncated = true;
                    prompt_tokens = nt;
                    slot.num_prompt_tokens = (int32_t)prompt_tokens.size();
                }
                if (slot.embedding && slot.num_prompt_tokens > be_->n_ubatch()) {     // this prompt is too large to process (:1431-1447)
                    slot.state = SlotState::kProcessing; slot.command = SlotCommand::kNone;
                    slot.Release();
                    SendError(slot, "input is too large to process. increase the physical batch size");
                    continue;
                }
                slot.smpl->reset();
                if (!slot.params.cache_prompt || (slot.embedding && (be_->pooling_type() == 1 || be_->pooling_type() == 2))) {
                    slot.n_past = 0;                       // (a pooled embedding needs every token of the prompt in this batch)
                } else {
                    for (int32_t t : prompt_tokens) slot.smpl->accept(t, false);
                    slot.n_past = (int32_t)common_part(slot.cache_tokens, prompt_tokens);
                }
                if (slot.n_past == slot.num_prompt_tokens && slot.n_past > 0) slot.n_past--;   // evaluate at least one token
                slot.num_prompt_tokens_processed = 0;
                slot.prompt_ready = true;
            }
            if (slot.embedding && n_tokens + slot.num_prompt_tokens > n_batch) continue;   // does not fit this tick (:1519-1524)
            // keep only the common part of the cache (:1536-1558)
            if (!be_->kv_seq_rm(slot.id, slot.n_past, -1)) {
                be_->kv_seq_rm(slot.id, -1, -1);
                slot.n_past = 0;
                slot.smpl->reset();
            }
            slot.cache_tokens.resize((size_t)slot.n_past);
            // an image whose turn it is: its rows go to the model now, as batches of embeddings (a llama_batch carries token ids OR rows, :1093-1107) - the text
            // in front of it was decoded with the previous tick's batch
            bool image_failed = false;
            while (slot.next_image < slot.images.size() && slot.images[slot.next_image].pos0 == slot.n_past) {
                SlotImage &im = slot.images[slot.next_image];
                const int E = be_->n_embd();
                for (int r0 = 0; r0 < im.n_rows && !image_failed; r0 += n_batch) {
                    const int nr = std::min(n_batch, im.n_rows - r0);
                    if (be_->decode_embd(im.rows.data() + (size_t)r0 * E, nr, slot.n_past + r0, slot.id) != 0) image_failed = true;
                }
                if (image_failed) break;
                slot.n_past += im.n_rows;
                slot.num_prompt_tokens_processed += im.n_rows;
                im.rows.clear(); im.rows.shrink_to_fit();
                slot.next_image++;
            }
            if (image_failed) {
                slot.state = SlotState::kProcessing; slot.command = SlotCommand::kNone;
                slot.Release();
                SendError(slot, "Failed processing images");
                continue;
            }
            const int text_end = slot.next_image < slot.images.size() ? slot.images[slot.next_image].pos0 : (int)prompt_tokens.size();   // (the next image's rows wait for the tick after this text)
            for (; slot.n_past < text_end && n_tokens < (int)b_token_.size(); ++slot.n_past) {
                // (an embedding prompt of a model that pools over the sequence needs every token's hidden state: all rows flagged)
                batch_add(prompt_tokens[(size_t)slot.n_past], slot.n_past, slot.id, slot.embedding && (be_->pooling_type() == 1 || be_->pooling_type() == 2));
                if (slot.params.cache_prompt) slot.cache_tokens.push_back(prompt_tokens[(size_t)slot.n_past]);
                slot.num_prompt_tokens_processed++;
            }
            if (slot.n_past == slot.num_prompt_tokens) {   // whole prompt queued: start decoding
                slot.state = SlotState::kProcessing;
                slot.command = SlotCommand::kNone;
                if (n_tokens > 0) b_logits_[(size_t)n_tokens - 1] = 1;
                slot.n_decoded = 0;
                slot.i_batch = n_tokens - 1;
            }
        }
    }
    if (n_tokens == 0) { all_slots_are_idle = true; return true; }

    // llama_set_embeddings is a context-wide switch that the reference flips per request (:299); here a tick is either an
    // embedding tick or a generation tick (decided at prompt ingestion below), so the switch follows the batch
    be_->set_embeddings(embd_tick);
    for (int32_t i = 0; i < n_tokens; i += n_batch) {   // :1628-1707
        const int32_t nt = std::min(n_batch, n_tokens - i);
        BatchView bv{nt, b_token_.data() + i, b_pos_.data() + i, b_seq_.data() + i, b_logits_.data() + i};
        const int64_t t_d0 = time_us();
        const int ret = be_->decode(bv);
        if (ret != 0) {
            if (n_batch == 1 || ret < 0) {
                // ret > 0 down to a batch of one: no KV cell left (the reference's message); ret < 0: a backend failure,
                // reported as what it is.  Slots whose prompt is only partly ingested are part of the failed batch too.
                const char *why = be_->last_error();
                const std::string msg = ret < 0 ? std::string("Decode failed: ") + (why && *why ? why : "backend error")
                                                : std::string("Input prompt is too big compared to KV size. Please try increasing KV size.");
                for (auto &slot : slots) {
                    const bool ingesting = slot.state == SlotState::kIdle && slot.command == SlotCommand::kLoadPrompt && slot.n_past > 0;
                    if (!slot.IsProcessing() && !ingesting) continue;
                    slot.state = SlotState::kProcessing; slot.command = SlotCommand::kNone;
                    slot.Release();
                    SendError(slot, msg);
                }
                break;
            }
            n_batch /= 2;       // retry with half the batch to find a free KV slot
            i -= n_batch;
            continue;
        }
        // phase 1: token ids of every generating slot of this chunk (host sampling in parallel, device argmax for plain greedy)
        const int64_t t_s0 = time_us();
        std::vector<LlamaClientSlot *> gen;
        for (auto &slot : slots) {
            if (slot.i_batch < i || slot.i_batch >= i + nt) continue;
            if (slot.embedding) {                        // prompt evaluated for embedding (:1670-1676)
                SendEmbedding(slot, slot.i_batch - i);
                slot.Release();
                slot.i_batch = -1;
                continue;
            }
            gen.push_back(&slot);
        }
        std::vector<int32_t> ids(gen.size(), -1);
        std::vector<const float *> rows(gen.size(), nullptr);
        const int n_vocab_now = be_->n_vocab();
        // device-side head of the chain (logit_bias -> penalties -> top_k): k candidates per slot cross instead of the row, all the
        // tick's sampling slots in one request to the backend
        std::vector<IBackend::TopkRequest> front;
        std::vector<size_t> front_slot;
        for (size_t gi = 0; gi < gen.size(); gi++) {
            LlamaClientSlot &slot = *gen[gi];
            if (slot.smpl->is_plain_greedy()) {             // device-side greedy front end: no pass over the vocabulary on the host
                ids[gi] = be_->argmax_ith(slot.i_batch - i);
                if (ids[gi] >= 0) { slot.smpl->set_greedy_result(ids[gi]); continue; }
            }
            Sampler::FrontPlan fp;
            if (be_->topk_max_k() > 0 && slot.smpl->plan_front(n_vocab_now, be_->topk_max_k(), be_->topk_max_adj(), fp)) {
                IBackend::TopkRequest rq;
                const SamplingParams &sp = slot.smpl->params();
                rq.i = slot.i_batch - i; rq.k = fp.k; rq.tok = std::move(fp.tok); rq.bias = std::move(fp.bias); rq.cnt = std::move(fp.cnt);
                rq.repeat = sp.penalty_repeat; rq.freq = sp.penalty_freq; rq.present = sp.penalty_present;
                front.push_back(std::move(rq));
                front_slot.push_back(gi);
            }
        }
        if (!front.empty()) be_->topk_batch(front);
        for (size_t fi = 0; fi < front.size(); fi++) {
            const size_t gi = front_slot[fi];
            LlamaClientSlot &slot = *gen[gi];
            const IBackend::TopkRequest &rq = front[fi];
            if (!rq.ok) continue;
            std::vector<TokenProb> c((size_t)rq.k);
            for (int j = 0; j < rq.k; j++) c[(size_t)j] = TokenProb{rq.out_tok[(size_t)j], rq.out_logit[(size_t)j]};
            ids[gi] = slot.smpl->finish(c);
            if (ids[gi] >= 0 && !slot.smpl->grammar_admits(ids[gi])) {   // refused by the request's grammar: the whole row, masked to what the grammar admits
                const float *row = be_->logits_ith(slot.i_batch - i);
                ids[gi] = row ? slot.smpl->resample_with_grammar(row, n_vocab_now) : -1;
            }
        }
        for (size_t gi = 0; gi < gen.size(); gi++)
            if (ids[gi] < 0) rows[gi] = be_->logits_ith(gen[gi]->i_batch - i);
        const int n_vocab = be_->n_vocab();
        const std::function<void(int)> sample_one = [&](int gi) {
            if (ids[(size_t)gi] < 0 && rows[(size_t)gi]) ids[(size_t)gi] = gen[(size_t)gi]->smpl->sample(rows[(size_t)gi], n_vocab);
        };
        if (!sample_pool_) sample_pool_.reset(new SamplePool((int)std::min<unsigned>(7u, std::max(1u, std::thread::hardware_concurrency()) - 1u)));
        sample_pool_->run((int)gen.size(), sample_one);
        const int64_t t_p0 = time_us();
        // phase 2: in slot order, exactly as the reference's loop (:1665-1704)
        for (size_t gi = 0; gi < gen.size(); gi++) {
            LlamaClientSlot &slot = *gen[gi];
            const int32_t id = ids[gi];
            if (id < 0) { slot.Release(); SendError(slot, "no logits"); slot.i_batch = -1; continue; }
            CompletionTokenOutput result;
            slot.smpl->accept(id);
            if (slot.n_decoded == 1) {
                slot.t_start_genereration = time_us();
                slot.t_prompt_processing = (double)(slot.t_start_genereration - slot.t_start_process_prompt) / 1e3;
            }
            result.tok = id;
            const auto &cand = slot.smpl->candidates();
            for (size_t k = 0; k < (size_t)slot.sparams.n_probs && k < cand.size(); k++) result.probs.push_back(cand[k]);
            if (!ProcessToken(result, slot)) {
                slot.Release();
                SendFinalResponse(slot);
            }
            slot.i_batch = -1;
        }
        const int64_t t_e = time_us();
        t_decode_us_ += (double)(t_s0 - t_d0); t_sample_us_ += (double)(t_p0 - t_s0); t_post_us_ += (double)(t_e - t_p0);
        n_ticks_++; n_tick_tokens_ += nt;
    }
    return true;
}

}  // namespace mi355
