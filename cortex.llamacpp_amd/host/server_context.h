// server_context.h — host-side mirror of the reference's slot scheduler: LlamaClientSlot (src/llama_client_slot.{h,cc}) and
// LlamaServerContext (src/llama_server_context.{h,cc}): task / result queues, LRU slot assignment, the continuous-batching
// UpdateSlots loop (:1248-1710), context shift (:1274-1306), prompt-prefix reuse (:1489-1558), stop strings and partial
// UTF-8 hold-back (ProcessToken :716-813), timings (llama_client_slot.cc:55-94).  Same names, same JSON keys.
// Embedding requests (SendEmbedding :1026-1070) and LLaVA image requests (image_data + [img-N] placeholders, :557-623, 814-831, 1073-1129) included.
// Not mirrored: infill, system-prompt broadcast.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "backend_iface.h"
#include "json.h"
#include "sampling.h"

namespace mi355 {

enum class SlotState : uint8_t { kIdle, kProcessing };
enum class SlotCommand : uint8_t { kNone, kLoadPrompt, kRelease };

struct SlotParams {
    bool stream = true;
    bool cache_prompt = false;
    uint32_t seed = 0xFFFFFFFFu;
    int32_t n_keep = 0;
    int32_t n_predict = -1;
    std::vector<std::string> antiprompt;
};

struct CompletionTokenOutput {
    std::vector<TokenProb> probs;
    int32_t tok = 0;
    std::string text_to_send;
};

struct ServerParams {            // the subset of common_params the loop reads
    int32_t n_predict = -1;
    int32_t n_keep = 0;
    int32_t n_parallel = 1;
    bool cont_batching = true;
    std::string model_alias;
    SamplingParams sampling;
};

// one image of a multimodal request (llama_client_slot.h SlotImage): its bytes as sent, the text in front of its [img-N] placeholder, and - once the prompt
// has been laid out - the position its embedding rows start at
struct SlotImage {
    int id = 0;
    std::string prefix_prompt;
    std::vector<uint8_t> bytes;
    std::vector<float> rows;          // [n_rows][n_embd]
    int n_rows = 0, pos0 = 0;
};

struct LlamaClientSlot {
    int id = 0;
    int task_id = -1;
    SlotParams params;
    SlotState state = SlotState::kIdle;
    SlotCommand command = SlotCommand::kNone;
    int64_t t_last_used = -1;
    int32_t n_ctx = 0, n_past = 0, n_decoded = 0, n_remaining = -1, i_batch = -1;
    int32_t num_prompt_tokens = 0, num_prompt_tokens_processed = 0;
    Json prompt;
    std::vector<int32_t> prompt_tokens;
    std::string generated_text;
    int32_t sampled = 0;
    std::vector<int32_t> cache_tokens;
    std::vector<CompletionTokenOutput> generated_token_probs;
    std::vector<SlotImage> images;  // multimodal request: in prompt order
    std::string input_suffix;       // the text behind the last image
    size_t next_image = 0;          // first image whose rows have not been decoded yet
    bool prompt_ready = false;      // prompt tokenised / truncated / matched against the cache (first visit done)
    bool has_next_token = true, truncated = false, stopped_eos = false, stopped_word = false, stopped_limit = false;
    bool oaicompat = false;
    bool embedding = false;         // task.embedding_mode (:1194): the prompt is evaluated for its embedding, nothing is sampled
    std::string oaicompat_model, stopping_word;
    SamplingParams sparams;
    std::unique_ptr<Sampler> smpl;
    size_t sent_count = 0, sent_token_probs_index = 0;
    int64_t t_start_process_prompt = 0, t_start_genereration = 0;
    double t_prompt_processing = 0, t_token_generation = 0;   // ms

    void Reset();
    bool HasBudget(const ServerParams &global);
    bool Available() const { return state == SlotState::kIdle && command == SlotCommand::kNone; }
    bool IsProcessing() const { return (state == SlotState::kIdle && command == SlotCommand::kLoadPrompt) || state == SlotState::kProcessing; }
    void AddTokenString(const CompletionTokenOutput &t) { if (command != SlotCommand::kRelease) generated_token_probs.push_back(t); }
    void Release();
    Json GetFormatedTimings() const;
};

struct TaskResult {
    int id = -1;
    bool stop = false;
    bool error = false;
    Json result_json;
};

class LlamaServerContext {
  public:
    LlamaServerContext(IBackend *be, const ServerParams &p);
    ~LlamaServerContext();
    void Initialize();                 // slots + background thread (llama_server_context.cc:244-282)
    void ReleaseResources();           // stop the loop (:366-380)

    int RequestCompletion(Json data, bool infill, bool embedding, int multitask_id);   // (:295-323)
    TaskResult NextResult(int task_id);                                                // (:325-352), blocks
    void RequestCancel(int task_id);                                                   // (:354-364)
    void KvCacheClear();

    // one scheduler tick, exposed for tests (the background thread calls it forever)
    bool UpdateSlots();
    std::vector<int32_t> Tokenize(const Json &json_prompt, bool add_bos, bool parse_special) const;

    ServerParams params;
    std::vector<LlamaClientSlot> slots;
    std::atomic<bool> model_loaded_external{false};
    bool all_slots_are_idle = false;
    bool clean_kv_cache = true;
    int n_ctx = 0;

  private:
    struct Task { int id; int target_id; bool cancel; Json data; bool embedding_mode = false; };
    void SendEmbedding(LlamaClientSlot &slot, int batch_index);
    // sampling of the slots of one decoded batch, spread over a few host threads (the chains are independent per slot;
    // at a 128 K vocabulary one chain costs ~0.1 ms, and the reference's loop runs them back to back)
    // MI355_LOOP_TIMING=1: where a scheduler tick spends its time (printed when the context is released)
    double t_decode_us_ = 0, t_sample_us_ = 0, t_post_us_ = 0, t_tick_us_ = 0;
    long n_ticks_ = 0, n_tick_tokens_ = 0;
    struct SamplePool;
    std::unique_ptr<SamplePool> sample_pool_;
    LlamaClientSlot *GetSlot(int id);
    bool LaunchSlotWithData(LlamaClientSlot *&slot, const Json &data);
    std::string launch_error_;         // why the last LaunchSlotWithData refused its request (sent to the client with the error)
    // grammar-constrained requests: every token's text once per model (what a candidate is checked with), and which tokens end the generation
    std::vector<std::string> grammar_pieces_;
    std::vector<uint8_t> grammar_eog_;
    void ProcessTasks();
    bool ProcessToken(CompletionTokenOutput &result, LlamaClientSlot &slot);
    void SendPartialResponse(LlamaClientSlot &slot, const CompletionTokenOutput &tkn);
    void SendFinalResponse(LlamaClientSlot &slot);
    void PostResult(TaskResult &&res);
    Json ProbsSlice(const LlamaClientSlot &slot, size_t from, size_t to) const;
    void SendError(LlamaClientSlot &slot, const std::string &err);
    Json GetFormatedGeneration(const LlamaClientSlot &slot) const;
  public:
    // (public for the host tests: the stop-string questions ProcessToken asks, under the reference's name)
    size_t FindStoppingStrings(const std::string &text, size_t last_token_size, bool full, LlamaClientSlot &slot);
    Json GetModelProps() const { return slots.empty() ? Json::object() : GetFormatedGeneration(slots[0]); }   // llama_server_context.cc:291-293
  private:
    Json ProbsToJson(const std::vector<CompletionTokenOutput> &probs) const;
    void DoBackgroundTasks();

    IBackend *be_;
    int id_gen_ = 0;
    std::mutex mutex_tasks_, mutex_results_;
    std::condition_variable condition_tasks_, condition_results_;
    std::deque<Task> queue_tasks_;
    std::deque<TaskResult> queue_results_;
    std::thread bgr_thread_;
    // the llama_batch the loop fills (:265)
    std::vector<int32_t> b_token_, b_pos_, b_seq_;
    std::vector<int8_t> b_logits_;
};

int64_t time_us();

}  // namespace mi355
