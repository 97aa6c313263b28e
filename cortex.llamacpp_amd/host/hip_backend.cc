// hip_backend.cc — see hip_backend.h.  Key mapping follows src/llama_engine.cc:587-660:
//   llama_model_path | model_path, ngl (300), ctx_len (2048), n_batch (2048), n_ubatch (= n_batch), n_parallel (1),
//   cache_type f16|q8_0|q4_0 (invalid -> f16), flash_attn (true; forced on by a quantised cache), embedding.
#include "hip_backend.h"

#include <exception>
#include <mutex>

#include "clip.h"
#include "log.h"

#include <sys/stat.h>

#include <algorithm>

#include "../csrc/dev_common.h"
#include "gguf.h"
#include "runtime.h"
#include "tp_split.h"

namespace mi355 {
namespace {

class HipBackend : public IBackend {
  public:
    HipBackend(std::unique_ptr<Model> m, std::unique_ptr<Context> c, Vocab v, bool device_sampling)
        : model_(std::move(m)), ctx_(std::move(c)), vocab_(std::move(v)), device_sampling_(device_sampling) {}
    ~HipBackend() override { ctx_.reset(); model_.reset(); }

    int n_ctx() const override { return (int)ctx_->cp.n_ctx; }
    int n_batch() const override { return (int)ctx_->cp.n_batch; }
    int n_ubatch() const override { return (int)ctx_->cp.n_ubatch; }
    int n_vocab() const override { return model_->hp.n_vocab; }
    int n_embd() const override { return model_->hp.n_embd; }
    const Vocab &vocab() const override { return vocab_; }

    int decode(const BatchView &b) override {
        n_seq_id_.assign((size_t)b.n_tokens, 1);
        seq_store_.assign(b.seq_id, b.seq_id + b.n_tokens);
        seq_ptr_.resize((size_t)b.n_tokens);
        for (int i = 0; i < b.n_tokens; i++) seq_ptr_[(size_t)i] = &seq_store_[(size_t)i];
        return ctx_->decode(b.n_tokens, b.token, b.pos, n_seq_id_.data(), seq_ptr_.data(), b.logits);
    }
    void set_clip(std::unique_ptr<ClipModel> c) { clip_ = std::move(c); }
    bool multimodal() const override { return clip_ != nullptr; }
    // (the bytes are a request's: whatever goes wrong with them - a picture too large to hold included - is that request's error, not the loop thread's end)
    // (the picture decoded here is kept for image_embed: a request's image is checked when the request arrives and embedded when its slot first visits the
    // prompt - the same bytes, and a large JPEG costs tens of milliseconds to decode.  A handful of pictures, at most 64 MB of pixels)
    bool image_check(const uint8_t *bytes, size_t n, std::string &err) override {
        try {
            ClipImageU8 img;
            err = clip_image_load_from_bytes(bytes, n, img);
            if (err.empty()) keep_decoded(bytes, n, std::move(img));
        } catch (const std::exception &e) { err = std::string("image: ") + e.what(); }
        return err.empty();
    }
    int image_embed(const uint8_t *bytes, size_t n, std::vector<float> &rows, std::string &err) override {
        if (!clip_) { err = "no multimodal projector loaded"; return -1; }
        try {
            ClipImageU8 img;
            if (!take_decoded(bytes, n, img)) err = clip_image_load_from_bytes(bytes, n, img);
            if (!err.empty()) return -1;
            int n_rows = 0;
            err = clip_->embed(img, rows, n_rows);
            return err.empty() ? n_rows : -1;
        } catch (const std::exception &e) { err = std::string("image: ") + e.what(); return -1; }
    }
    int decode_embd(const float *rows, int n, int pos0, int seq) override {
        pos_store_.resize((size_t)n);
        for (int i = 0; i < n; i++) pos_store_[(size_t)i] = pos0 + i;
        n_seq_id_.assign((size_t)n, 1);
        seq_store_.assign((size_t)n, seq);
        seq_ptr_.resize((size_t)n);
        for (int i = 0; i < n; i++) seq_ptr_[(size_t)i] = &seq_store_[(size_t)i];
        no_logits_.assign((size_t)n, 0);
        return ctx_->decode(n, nullptr, pos_store_.data(), n_seq_id_.data(), seq_ptr_.data(), no_logits_.data(), rows);
    }
    const char *last_error() const override { return ctx_->last_error.c_str(); }
    const float *logits_ith(int i) override { return ctx_->logits_ith(i); }
    int argmax_ith(int i) override { return ctx_->argmax_ith(i); }
    int topk_ith(int i, int k, const std::vector<int32_t> &adj_tok, const std::vector<float> &adj_bias, const std::vector<int32_t> &adj_cnt, float repeat, float freq,
                 float present, int32_t *toks, float *logits) override {
        if (adj_tok.size() > (size_t)TOPK_MAX_ADJ) return -1;
        TopkAdj a{};
        a.n = (int)adj_tok.size(); a.repeat = repeat; a.freq = freq; a.present = present;
        for (int j = 0; j < a.n; j++) { a.tok[j] = adj_tok[(size_t)j]; a.bias[j] = adj_bias[(size_t)j]; a.cnt[j] = adj_cnt[(size_t)j]; }
        return ctx_->topk_ith(i, k, a, toks, logits);
    }
    void topk_batch(std::vector<TopkRequest> &reqs) override {
        const int n = (int)reqs.size();
        if (n == 0) return;
        adjs_.resize((size_t)n); is_.resize((size_t)n); ks_.resize((size_t)n);
        out_t_.resize((size_t)n * TOPK_MAX_K); out_l_.resize((size_t)n * TOPK_MAX_K);
        bool fits = true;
        for (int r = 0; r < n; r++) {
            const TopkRequest &q = reqs[(size_t)r];
            TopkAdj &a = adjs_[(size_t)r];
            if (q.tok.size() > (size_t)TOPK_MAX_ADJ || q.k < 1 || q.k > TOPK_MAX_K) { fits = false; break; }
            a.n = (int)q.tok.size(); a.repeat = q.repeat; a.freq = q.freq; a.present = q.present;
            for (int j = 0; j < a.n; j++) { a.tok[j] = q.tok[(size_t)j]; a.bias[j] = q.bias[(size_t)j]; a.cnt[j] = q.cnt[(size_t)j]; }
            is_[(size_t)r] = q.i; ks_[(size_t)r] = q.k;
        }
        const bool ok = fits && ctx_->topk_rows(n, is_.data(), ks_.data(), adjs_.data(), out_t_.data(), out_l_.data()) == n;
        for (int r = 0; r < n; r++) {
            TopkRequest &q = reqs[(size_t)r];
            q.ok = ok;
            if (!ok) continue;
            q.out_tok.assign(out_t_.begin() + (long)r * TOPK_MAX_K, out_t_.begin() + (long)r * TOPK_MAX_K + q.k);
            q.out_logit.assign(out_l_.begin() + (long)r * TOPK_MAX_K, out_l_.begin() + (long)r * TOPK_MAX_K + q.k);
        }
    }
    int topk_max_k() const override { return device_sampling_ ? TOPK_MAX_K : 0; }
    int topk_max_adj() const override { return TOPK_MAX_ADJ; }
    int pooling_type() const override { return model_->hp.pooling_type; }
    bool is_encoder() const override { return model_->hp.encoder; }
    void set_embeddings(bool on) override { ctx_->embeddings_enabled = on || ctx_->model->hp.encoder; }
    const float *embeddings_ith(int i) override { return ctx_->embeddings_ith(i); }
    void kv_clear() override { ctx_->kv_clear(); }
    bool kv_seq_rm(int seq, int p0, int p1) override { return ctx_->kv_seq_rm(seq, p0, p1); }
    void kv_seq_add(int seq, int p0, int p1, int delta) override { ctx_->kv_seq_add(seq, p0, p1, delta); }
    void kv_seq_cp(int src, int dst, int p0, int p1) override { ctx_->kv_seq_cp(src, dst, p0, p1); }

  private:
    static uint64_t bytes_key(const uint8_t *b, size_t n) {           // FNV-1a over the bytes
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
        return h ^ (uint64_t)n;
    }
    void keep_decoded(const uint8_t *b, size_t n, ClipImageU8 &&img) {
        const size_t px = img.rgb.size();
        if (px > ((size_t)64 << 20)) return;
        std::lock_guard<std::mutex> lk(decoded_mu_);
        while (!decoded_.empty() && (decoded_.size() >= 8 || decoded_bytes_ + px > ((size_t)64 << 20))) { decoded_bytes_ -= decoded_.front().second.rgb.size(); decoded_.erase(decoded_.begin()); }
        decoded_bytes_ += px;
        decoded_.emplace_back(bytes_key(b, n), std::move(img));
    }
    bool take_decoded(const uint8_t *b, size_t n, ClipImageU8 &img) {
        const uint64_t k = bytes_key(b, n);
        std::lock_guard<std::mutex> lk(decoded_mu_);
        for (size_t i = 0; i < decoded_.size(); i++)
            if (decoded_[i].first == k) { img = std::move(decoded_[i].second); decoded_bytes_ -= img.rgb.size(); decoded_.erase(decoded_.begin() + (long)i); return true; }
        return false;
    }
    std::mutex decoded_mu_;
    std::vector<std::pair<uint64_t, ClipImageU8>> decoded_;
    size_t decoded_bytes_ = 0;
    std::unique_ptr<Model> model_;
    std::unique_ptr<Context> ctx_;
    Vocab vocab_;
    bool device_sampling_ = true;
    std::unique_ptr<ClipModel> clip_;
    std::vector<int32_t> n_seq_id_, seq_store_, pos_store_;
    std::vector<int8_t> no_logits_;
    std::vector<TopkAdj> adjs_;
    std::vector<int> is_, ks_;
    std::vector<int32_t> out_t_;
    std::vector<float> out_l_;
    std::vector<int32_t *> seq_ptr_;
};

int cache_type_from_str(const std::string &s) {   // llama_engine.cc:29-55 (IsValidCacheType / kv_cache_type_from_str)
    if (s == "q8_0") return T_Q8_0;
    if (s == "q4_0") return T_Q4_0;
    return T_F16;
}

}  // namespace

std::unique_ptr<IBackend> make_hip_backend(const Json &body, BackendInfo &info, std::string &err) {
    // "split_mode": "row" - this process becomes rank 0 of a row split it forms itself (tp_split.cc); the shard it loads comes back through here without the key
    if (tp_split_requested(body)) return make_split_backend(body, info, err);
    std::string path = body.value<std::string>("llama_model_path", "");
    if (path.empty()) path = body.value<std::string>("model_path", "");
    if (path.empty()) { err = "Missing model path in request"; return nullptr; }
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { err = "Could not find model in path " + path; return nullptr; }
    // ngl (src/llama_engine.cc:609-611) is the reference's count of layers placed on the GPU; here every layer lives in HBM whatever it says
    // (there is no host compute path to leave layers on): 0 - the reference's CPU configuration - and partial counts are accepted and run
    // fully on the device; results do not depend on placement.  The load logs what was asked for.
    const int ngl_asked = body.value<int>("ngl", 300);

    int status = 0;
    std::unique_ptr<Model> model(model_load(path, body.value<int>("main_gpu", 0), err, status, body.value<int>("prefill_planes", -1),
                                            body.value<int>("tp_rank", 0), body.value<int>("tp_size", 1)));   // row split: the process's group is formed first (mi355_tp_init)
    if (!model) return nullptr;
    if (ngl_asked <= (int)model->hp.n_layer)
        log_line(LOG_WARN, "ngl=%d asked for (%s): all %d layers and the output head are resident on the MI355X - this engine has no host compute path",
                 ngl_asked, ngl_asked <= 0 ? "CPU only" : "partial offload", (int)model->hp.n_layer);

    Vocab vocab;
    std::string verr;
    if (!vocab.load(*model->file, verr)) { err = "tokenizer: " + verr; return nullptr; }
    if (vocab.n_tokens() != model->hp.n_vocab) { err = "tokenizer size does not match the embedding table"; return nullptr; }

    ContextParams cp;
    const int n_parallel = std::max(1, body.value<int>("n_parallel", 1));
    cp.n_ctx = (uint32_t)std::max(8, body.value<int>("ctx_len", 2048));
    // a projector file makes the context multimodal; the reference then asks for at least 2048 cells, room for an image's embedding rows (llama_server_context.cc:194-205)
    const std::string mmproj = body["mmproj"].is_string() ? body["mmproj"].as_string() : std::string();
    std::unique_ptr<ClipModel> clip;
    if (!mmproj.empty()) {
        if (model->hp.encoder) { err = "mmproj: an embedding model cannot take a multimodal projector"; return nullptr; }
        // clip_model_load + the width check of llama_server_context.cc:216-229
        clip.reset(new ClipModel);
        const std::string cerr = clip->load(mmproj, body.value<int>("main_gpu", 0));
        if (!cerr.empty()) { err = "unable to load clip model: " + cerr; return nullptr; }
        if (clip->proj_dim != model->hp.n_embd) {
            err = "embedding dim of the multimodal projector (" + std::to_string(clip->proj_dim) + ") is not equal to that of the model (" + std::to_string(model->hp.n_embd) +
                  "). Make sure that you use the correct mmproj file.";
            return nullptr;
        }
        // LLaVA-1.6 (an image grid: up to five encoded images a picture) needs more room than LLaVA-1.5; the reference tells the two apart by the model's
        // file name (IsLlava_1_6, :170-175, :195) - here the projector file's own image grid counts as well
        const bool v16 = path.find("llava-v1.6") != std::string::npos || clip->max_image_rows() > clip->n_patches();
        if (v16 && cp.n_ctx < 4096) { cp.n_ctx = 4096; log_line(LOG_INFO, "Request %d for context length for llava-1.6", 4096); }
        else if (cp.n_ctx < 2048) { cp.n_ctx = 2048; log_line(LOG_INFO, "Request %d for context length for the image embedding", 2048); }
    }
    cp.n_batch = (uint32_t)std::max(1, body.value<int>("n_batch", 2048));
    cp.n_ubatch = (uint32_t)std::max(1, body.value<int>("n_ubatch", (int)cp.n_batch));
    cp.n_batch = std::min(cp.n_batch, cp.n_ctx);
    cp.n_ubatch = std::min(cp.n_ubatch, cp.n_batch);
    cp.n_seq_max = (uint32_t)n_parallel;
    cp.type_k = cp.type_v = cache_type_from_str(body.value<std::string>("cache_type", "f16"));
    cp.flash_attn = body.value<bool>("flash_attn", true) || cp.type_k != T_F16;
    cp.embeddings = false;
    cp.use_graphs = body.value<bool>("use_graphs", true);
    // a flagged row is not copied to the host with every step: plain greedy takes the device arg-max, the usual chains take the device top-k
    // (Sampler::plan_front), and a chain that needs the whole row (mirostat, top_k off, long penalty windows) fetches it on demand
    cp.logits_to_host = body.value<bool>("logits_to_host", false);
    std::unique_ptr<Context> ctx(new Context(model.get(), cp));
    if (!ctx->init(err)) return nullptr;

    info.vram = model->device_bytes + ctx->device_bytes;
    info.ram = model->host_bytes;
    info.model_size = model->file_tensor_bytes;
    if (clip) info.vram += clip->device_bytes;
    std::unique_ptr<HipBackend> be(new HipBackend(std::move(model), std::move(ctx), std::move(vocab), body.value<bool>("device_sampling", true)));
    be->set_clip(std::move(clip));
    return be;
}

}  // namespace mi355
