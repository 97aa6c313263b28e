// sampling.h — host sampler chain the slot loop applies to one logits row.  Stands in for common_sampler_{init,reset,
// accept,sample,get_candidates} (reference call sites src/llama_server_context.cc:626-628,886,1487,1495,1553,1680-1690);
// chain order per SURVEY.md §A.6: logit_bias -> penalties(last_n, repeat, freq, presence) -> top_k -> typical_p -> top_p
// -> min_p -> temperature (entropy-driven when dynatemp_range > 0; <= 0 => greedy) -> seeded draw.  mirostat 1 / 2 replace
// everything after the penalties by temperature -> the mirostat truncation -> draw, as common_sampler_init chains them
// (upstream common/sampling.cpp, llama-sampling.cpp: llama_sampler_temp_ext / _mirostat / _mirostat_v2).  A grammar (grammar.h) constrains the
// draw the way common_sampler_sample does with grammar_first = false: the chain runs unconstrained, the drawn token is checked against the grammar,
// and only if it is refused the row is masked to the admissible tokens and the chain runs again.  Not carried over: DRY, XTC.
#pragma once

#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "grammar.h"

namespace mi355 {

struct SamplingParams {            // defaults of chat_completion_request.h:60-92 / common_params_sampling
    uint32_t seed = 0xFFFFFFFFu;   // -1 => random
    int32_t n_probs = 0;
    int32_t min_keep = 0;
    int32_t top_k = 40;
    float top_p = 0.95f;
    float min_p = 0.05f;
    float typ_p = 1.0f;
    float temp = 0.8f;
    float dynatemp_range = 0.0f, dynatemp_exponent = 1.0f;
    int32_t penalty_last_n = 64;   // 0 = penalties off, -1 = the whole context (penalty_n_ctx tokens)
    int32_t penalty_n_ctx = 0;     // what -1 stands for: the slot's context size
    float penalty_repeat = 1.0f, penalty_freq = 0.0f, penalty_present = 0.0f;
    int32_t mirostat = 0;
    float mirostat_tau = 5.0f, mirostat_eta = 0.1f;
    bool ignore_eos = false;
    std::vector<std::pair<int32_t, float>> logit_bias;
};

struct TokenProb { int32_t tok; float p; };

class Sampler {
  public:
    explicit Sampler(const SamplingParams &p = SamplingParams());
    void reset();
    void accept(int32_t token, bool advance_grammar = true);     // (prompt tokens are accepted without the grammar: src/llama_server_context.cc:1493-1495)
    int32_t sample(const float *logits, int n_vocab);
    // grammar-constrained sampling: `pieces[token]` = the bytes of the token's text ("" for tokens without text), eog[token] != 0 for the end tokens;
    // both belong to the caller and outlive the sampler
    void set_grammar(std::shared_ptr<const Grammar> g, const std::vector<std::string> *pieces, const std::vector<uint8_t> *eog);
    bool has_grammar() const { return gm_ != nullptr; }
    bool grammar_admits(int32_t token) const;
    // the token finish() / the device arg-max produced was refused by the grammar: mask the whole row and run the chain again
    int32_t resample_with_grammar(const float *logits, int n_vocab);
    // sample() = head (logit_bias -> penalties -> top_k over the whole row) + finish().  The head can run on the device instead (SURVEY.md §8f.1,
    // mi355_get_topk_ith): plan_front says whether this sampler's state allows it (false: take sample()) and what to send; finish() then runs the rest
    // of the chain on the (token, adjusted logit) candidates, best first - the same token and the same candidates() as sample() on the whole row
    struct FrontPlan { int k = 0; std::vector<int32_t> tok; std::vector<float> bias; std::vector<int32_t> cnt; };
    bool plan_front(int n_vocab, int max_k, int max_adj, FrontPlan &pl) const;
    int32_t finish(std::vector<TokenProb> &c);
    // true when sample() would return the plain argmax of the raw logits (greedy, nothing modifies them, no probabilities
    // asked for): the caller may then take the device-side argmax instead of reading the row
    bool is_plain_greedy() const;
    void set_greedy_result(int32_t tok) { cand_.assign(1, TokenProb{tok, 1.0f}); }
    // candidates of the last sample(), sorted by probability (descending), probabilities after the chain
    const std::vector<TokenProb> &candidates() const { return cand_; }
    const SamplingParams &params() const { return p_; }

  private:
    size_t front_k(int n_vocab) const;
    int32_t sample_chain(const float *logits, int n_vocab);
    SamplingParams p_;
    std::unique_ptr<GrammarMatcher> gm_;
    const std::vector<std::string> *pieces_ = nullptr;
    const std::vector<uint8_t> *eog_ = nullptr;
    std::vector<int32_t> prev_;     // ring of accepted tokens (penalty window)
    std::vector<TokenProb> cand_;
    std::mt19937 rng_;
    std::vector<float> probs_;       // (scratch of the draw)
    float mu_ = 0.0f;               // mirostat: running surprise target (starts at 2 tau)
};

}  // namespace mi355
