// backend_iface.h — what the continuous-batching loop needs from the arithmetic backend: exactly the llama.h subset the
// reference's LlamaServerContext consumes (SURVEY.md §8b "inner boundary").  The product implementation is HipBackend
// (hip_backend.cc, over the C-ABI objects); host-logic unit tests plug in a deterministic fake.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "vocab.h"

namespace mi355 {

struct BatchView {           // llama_batch as filled at llama_server_context.cc:1630-1635
    int32_t n_tokens = 0;
    const int32_t *token = nullptr;
    const int32_t *pos = nullptr;
    const int32_t *seq_id = nullptr;   // one sequence id per token (the loop never uses more)
    const int8_t *logits = nullptr;
};

class IBackend {
  public:
    virtual ~IBackend() = default;
    virtual int n_ctx() const = 0;
    virtual int n_batch() const = 0;
    virtual int n_ubatch() const = 0;
    virtual int n_vocab() const = 0;
    virtual int n_embd() const = 0;
    virtual const Vocab &vocab() const = 0;
    // llama_decode: 0 ok, 1 no KV slot, < 0 error
    virtual int decode(const BatchView &b) = 0;
    // what went wrong in the last decode that returned < 0 (empty when the backend keeps no message)
    virtual const char *last_error() const { return ""; }
    // logits row of batch index i of the last decode (llama_get_logits_ith)
    virtual const float *logits_ith(int i) = 0;
    // index of the largest logit of batch row i, computed on the device (first maximum wins), or -1 when the backend has
    // no such front end: lets a purely greedy request skip the host pass over the vocabulary
    virtual int argmax_ith(int i) { (void)i; return -1; }
    // device-side head of the sampler chain (Sampler::plan_front / finish): the k best (token, adjusted logit) candidates of batch row i, best first;
    // returns k, or -1 when the backend has no such front end (the caller then samples from the whole row)
    virtual int topk_ith(int i, int k, const std::vector<int32_t> &adj_tok, const std::vector<float> &adj_bias, const std::vector<int32_t> &adj_cnt,
                         float repeat, float freq, float present, int32_t *toks, float *logits) {
        (void)i; (void)k; (void)adj_tok; (void)adj_bias; (void)adj_cnt; (void)repeat; (void)freq; (void)present; (void)toks; (void)logits;
        return -1;
    }
    // the same for every sampling slot of a scheduler tick at once (one set of launches, one synchronisation); default: row by row
    struct TopkRequest {
        int i = 0, k = 0;
        std::vector<int32_t> tok, cnt;
        std::vector<float> bias;
        float repeat = 1.0f, freq = 0.0f, present = 0.0f;
        std::vector<int32_t> out_tok;
        std::vector<float> out_logit;
        bool ok = false;
    };
    virtual void topk_batch(std::vector<TopkRequest> &reqs) {
        for (auto &r : reqs) {
            r.out_tok.assign((size_t)r.k, 0); r.out_logit.assign((size_t)r.k, 0.0f);
            r.ok = topk_ith(r.i, r.k, r.tok, r.bias, r.cnt, r.repeat, r.freq, r.present, r.out_tok.data(), r.out_logit.data()) == r.k;
        }
    }
    virtual int topk_max_k() const { return 0; }      // 0: no device front end
    virtual int topk_max_adj() const { return 0; }
    // llama_set_embeddings / llama_get_embeddings_ith (llama_server_context.cc:299, 1042-1044)
    // {arch}.pooling_type of the model: 0 none (an embedding request answers with the hidden state of its last token, llama_get_embeddings_ith), 1 mean,
    // 2 cls, 3 last (llama_get_embeddings_seq: pooled over the sequence's tokens; src/llama_server_context.cc:1041-1044)
    virtual int pooling_type() const { return 0; }
    // a bidirectional encoder (general.architecture nomic-bert): embeddings only, no next-token head - the engine treats it as model_type "embedding" whatever the load request said
    virtual bool is_encoder() const { return false; }
    virtual void set_embeddings(bool on) = 0;
    virtual const float *embeddings_ith(int i) = 0;
    // ---- LLaVA: a projector file ("mmproj") was loaded beside the model (llama_server_context.cc:184-230)
    virtual bool multimodal() const { return false; }
    // clip_image_load_from_bytes (:568): can these bytes be decoded as an image - if not, why
    virtual bool image_check(const uint8_t *bytes, size_t n, std::string &err) { (void)bytes; (void)n; err = "no multimodal projector loaded"; return false; }
    // llava_image_embed_make_with_clip_img (:820): the image's embedding rows [n][n_embd]; returns n, or < 0 with err set
    virtual int image_embed(const uint8_t *bytes, size_t n, std::vector<float> &rows, std::string &err) { (void)bytes; (void)n; (void)rows; err = "no multimodal projector loaded"; return -1; }
    // llama_decode on a llava_embd_batch (:1093-1107): n embedding rows at positions pos0 .. of sequence seq, no logits.  0 ok, 1 no KV slot, < 0 error
    virtual int decode_embd(const float *rows, int n, int pos0, int seq) { (void)rows; (void)n; (void)pos0; (void)seq; return -1; }
    virtual void kv_clear() = 0;
    virtual bool kv_seq_rm(int seq, int p0, int p1) = 0;
    virtual void kv_seq_add(int seq, int p0, int p1, int delta) = 0;
    virtual void kv_seq_cp(int src, int dst, int p0, int p1) = 0;
};

}  // namespace mi355
