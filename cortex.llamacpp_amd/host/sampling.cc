#include "sampling.h"

#include <algorithm>
#include <cmath>
#include <unordered_map>

namespace mi355 {

Sampler::Sampler(const SamplingParams &p) : p_(p) { reset(); }

void Sampler::reset() {
    prev_.clear();
    cand_.clear();
    uint32_t seed = p_.seed;
    if (seed == 0xFFFFFFFFu) seed = std::random_device{}();
    rng_.seed(seed);
    mu_ = 2.0f * p_.mirostat_tau;
    if (gm_) gm_->reset();
}

void Sampler::set_grammar(std::shared_ptr<const Grammar> g, const std::vector<std::string> *pieces, const std::vector<uint8_t> *eog) {
    gm_.reset(g ? new GrammarMatcher(std::move(g)) : nullptr);
    pieces_ = pieces; eog_ = eog;
}

bool Sampler::grammar_admits(int32_t token) const {
    if (!gm_) return true;
    if (token < 0 || (size_t)token >= pieces_->size()) return false;
    if ((*eog_)[(size_t)token]) return gm_->can_end();
    return gm_->admits((*pieces_)[(size_t)token]);
}

int32_t Sampler::resample_with_grammar(const float *logits, int n_vocab) {
    std::vector<float> masked(logits, logits + n_vocab);
    int32_t n_ok = 0, first_eog = -1;
    for (int t = 0; t < n_vocab; t++) {
        if (first_eog < 0 && (size_t)t < eog_->size() && (*eog_)[(size_t)t]) first_eog = t;
        if (grammar_admits(t)) n_ok++; else masked[(size_t)t] = -INFINITY;
    }
    if (n_ok == 0) {            // nothing in this vocabulary can continue the sentence: end the generation rather than leave the grammar
        cand_.assign(1, TokenProb{first_eog, 1.0f});
        return first_eog;
    }
    return sample_chain(masked.data(), n_vocab);
}

int32_t Sampler::sample(const float *logits, int n_vocab) {
    const int32_t t = sample_chain(logits, n_vocab);
    if (!gm_ || grammar_admits(t)) return t;
    return resample_with_grammar(logits, n_vocab);
}

void Sampler::accept(int32_t token, bool advance_grammar) {
    if (gm_ && advance_grammar && token >= 0 && (size_t)token < pieces_->size() && !(*eog_)[(size_t)token]) gm_->accept((*pieces_)[(size_t)token]);
    prev_.push_back(token);
    // -1 = the whole context (upstream: penalty_last_n < 0 -> n_ctx), 0 = no window at all
    const size_t keep = p_.penalty_last_n < 0 ? (size_t)std::max(p_.penalty_n_ctx, 64) : (size_t)p_.penalty_last_n;
    if (keep == 0) prev_.clear();
    else if (prev_.size() > keep) prev_.erase(prev_.begin(), prev_.begin() + (long)(prev_.size() - keep));
}

bool Sampler::is_plain_greedy() const {
    if (gm_) return false;
    if (p_.temp > 0.0f || p_.n_probs > 0 || !p_.logit_bias.empty()) return false;   // (temp <= 0 is greedy in the mirostat chains too: temperature comes first there)
    const bool penalties = p_.penalty_repeat != 1.0f || p_.penalty_freq != 0.0f || p_.penalty_present != 0.0f;
    return !penalties || prev_.empty();
}

static void softmax_sorted(std::vector<TokenProb> &c) {   // c sorted by logit desc; p holds logits on entry
    if (c.empty()) return;
    const float mx = c[0].p;
    double sum = 0.0;
    for (auto &e : c) { e.p = expf(e.p - mx); sum += e.p; }
    for (auto &e : c) e.p = (float)(e.p / sum);
}

// The logits row stays where it is (513 KB for a 128 K vocabulary): logit_bias and the penalties touch a handful of
// tokens, so they are kept as a small sorted override list and every pass over the row merges it on the fly; the top-k
// selection is one pass with a k-element heap.  (The first version copied the row into (token, logit) pairs and
// partial-sorted them: 230-530 us per token, a quarter of a decode step.)
int32_t Sampler::sample_chain(const float *logits, int n_vocab) {
    // ---- sparse overrides: token -> modified logit, ascending token order
    std::vector<std::pair<int32_t, float>> ov;
    auto ov_find = [&](int32_t t) -> float * {
        for (auto &e : ov) if (e.first == t) return &e.second;
        return nullptr;
    };
    for (const auto &lb : p_.logit_bias) {
        if (lb.first < 0 || lb.first >= n_vocab) continue;
        if (float *v = ov_find(lb.first)) *v += lb.second; else ov.emplace_back(lb.first, logits[lb.first] + lb.second);
    }
    if (!prev_.empty() && (p_.penalty_repeat != 1.0f || p_.penalty_freq != 0.0f || p_.penalty_present != 0.0f)) {
        std::unordered_map<int32_t, int> cnt;
        for (int32_t t : prev_) cnt[t]++;
        for (const auto &kv : cnt) {
            if (kv.first < 0 || kv.first >= n_vocab) continue;
            float *v = ov_find(kv.first);
            if (!v) { ov.emplace_back(kv.first, logits[kv.first]); v = &ov.back().second; }
            float &l = *v;
            if (l <= 0) l *= p_.penalty_repeat; else l /= p_.penalty_repeat;
            l -= (float)kv.second * p_.penalty_freq + (kv.second > 0 ? p_.penalty_present : 0.0f);
        }
    }
    std::sort(ov.begin(), ov.end(), [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.first < b.first; });
    // better(a, b): a ranks before b (higher logit; lower token id on ties)
    auto better = [](const TokenProb &a, const TokenProb &b) { return a.p > b.p || (a.p == b.p && a.tok < b.tok); };
    // one pass over the row keeping the k best in a heap whose top is the WORST kept element.  The row is walked in
    // chunks of 64 whose raw maximum (a vectorisable loop) is compared with the current threshold first: once the heap is
    // warm almost every chunk is rejected without a per-element branch.  Overridden tokens are skipped in the walk and
    // offered afterwards with their modified value.
    auto top_k_pass = [&](size_t k, std::vector<TokenProb> &out) {
        out.clear();
        out.reserve(k + 1);
        auto offer = [&](int32_t tok, float v) {
            const TokenProb e{tok, v};
            if (out.size() < k) {
                out.push_back(e);
                if (out.size() == k) std::make_heap(out.begin(), out.end(), better);   // max-heap under "better" = worst on top
            } else if (better(e, out.front())) {
                std::pop_heap(out.begin(), out.end(), better);
                out.back() = e;
                std::push_heap(out.begin(), out.end(), better);
            }
        };
        size_t oi = 0;
        const size_t no = ov.size();
        constexpr int CH = 64;
        for (int i0 = 0; i0 < n_vocab; i0 += CH) {
            const int i1 = std::min(n_vocab, i0 + CH);
            if (out.size() == k) {
                float m0 = logits[i0], m1 = m0, m2 = m0, m3 = m0;
                int i = i0;
                for (; i + 4 <= i1; i += 4) {
                    m0 = logits[i] > m0 ? logits[i] : m0; m1 = logits[i + 1] > m1 ? logits[i + 1] : m1;
                    m2 = logits[i + 2] > m2 ? logits[i + 2] : m2; m3 = logits[i + 3] > m3 ? logits[i + 3] : m3;
                }
                for (; i < i1; i++) m0 = logits[i] > m0 ? logits[i] : m0;
                const float mx = std::max(std::max(m0, m1), std::max(m2, m3));
                if (mx < out.front().p) {                  // nothing here can enter (ties lose: ids only grow)
                    while (oi < no && ov[oi].first < i1) oi++;
                    continue;
                }
            }
            for (int i = i0; i < i1; i++) {
                if (oi < no && ov[oi].first == i) { oi++; continue; }
                offer(i, logits[i]);
            }
        }
        for (const auto &e : ov) offer(e.first, e.second);
        if (out.size() < k) std::make_heap(out.begin(), out.end(), better);
        std::sort(out.begin(), out.end(), better);
    };
    if (p_.temp <= 0.0f) {   // greedy: first maximum wins
        std::vector<TokenProb> c;
        top_k_pass(front_k(n_vocab), c);
        return finish(c);
    }
    // the draw itself is the standard library's std::discrete_distribution over the final candidates' probabilities on a std::mt19937 seeded with the request's
    // seed - the class and generator llama_sampler_dist uses upstream (a draw consumes two 32-bit outputs: generate_canonical<double, 53>)
    auto draw = [&](const std::vector<TokenProb> &cc) -> size_t {
        probs_.resize(cc.size());
        for (size_t i = 0; i < cc.size(); i++) probs_[i] = cc[i].p;
        std::discrete_distribution<int> dist(probs_.begin(), probs_.end());
        return (size_t)dist(rng_);
    };
    if (p_.mirostat == 1 || p_.mirostat == 2) {
        // temperature over the whole vocabulary, then the mirostat truncation (no top_k / top_p / min_p in this chain)
        std::vector<TokenProb> c;
        top_k_pass((size_t)n_vocab, c);
        if (p_.temp != 1.0f) for (auto &e : c) e.p /= p_.temp;
        softmax_sorted(c);
        if (p_.mirostat == 1) {
            // Zipf exponent from the top m = 100 probabilities, then the k that gives a surprise of mu
            const size_t m = 100;
            double sum_ti_bi = 0.0, sum_ti_sq = 0.0;
            for (size_t i = 0; i + 1 < std::min(m, c.size()); i++) {
                const double t_i = log((double)(i + 2) / (double)(i + 1)), b_i = log((double)c[i].p / (double)std::max(c[i + 1].p, 1e-37f));
                sum_ti_bi += t_i * b_i; sum_ti_sq += t_i * t_i;
            }
            const double s_hat = sum_ti_sq > 0 ? sum_ti_bi / sum_ti_sq : 1.0;
            const double eps_hat = s_hat - 1.0;
            double kf = pow((eps_hat * pow(2.0, (double)mu_)) / (1.0 - pow((double)n_vocab, -eps_hat)), 1.0 / s_hat);
            if (!(kf >= 1.0)) kf = 1.0;                            // (also catches NaN)
            const size_t kk = (size_t)std::min<double>(kf, (double)c.size());
            c.resize(std::max<size_t>(kk, 1));
        } else {
            size_t keep = c.size();
            for (size_t i = 0; i < c.size(); i++) if (-log2((double)std::max(c[i].p, 1e-37f)) > (double)mu_) { keep = i; break; }
            c.resize(std::max<size_t>(keep, 1));
        }
        { double s2 = 0; for (auto &e : c) s2 += e.p; for (auto &e : c) e.p = (float)(e.p / s2); }
        const size_t idx = draw(c);
        const float observed = -log2f(std::max(c[idx].p, 1e-37f));
        mu_ -= p_.mirostat_eta * (observed - p_.mirostat_tau);
        cand_ = c;
        return c[idx].tok;
    }
    // top_k (also establishes descending order)
    std::vector<TokenProb> c;
    top_k_pass(front_k(n_vocab), c);
    return finish(c);
}

// how many candidates the head of the chain (logit_bias -> penalties -> top_k) hands to the rest: the greedy chains keep max(n_probs, 1)
size_t Sampler::front_k(int n_vocab) const {
    if (p_.temp <= 0.0f) return std::max<size_t>((size_t)std::max(p_.n_probs, 0), 1);
    const size_t min_keep = (size_t)std::max(p_.min_keep, 1);
    size_t k = p_.top_k <= 0 ? (size_t)n_vocab : std::min<size_t>((size_t)p_.top_k, (size_t)n_vocab);
    k = std::max(k, min_keep);
    return std::min(k, (size_t)n_vocab);
}

// The head of the chain can run on the device (mi355_get_topk_ith: the same f32 operations on the adjusted tokens, the same order): this says whether
// this sampler's state allows it and what to send - at most max_k candidates, at most max_adj adjusted tokens, no mirostat (it truncates the whole
// vocabulary), one bias per token (two biases on one token are added one after the other on the host: not the same float as their sum).
bool Sampler::plan_front(int n_vocab, int max_k, int max_adj, FrontPlan &pl) const {
    if (p_.mirostat == 1 || p_.mirostat == 2) return false;
    const size_t k = front_k(n_vocab);
    if (k > (size_t)max_k) return false;
    pl.k = (int)k;
    pl.tok.clear(); pl.bias.clear(); pl.cnt.clear();
    auto slot = [&](int32_t t) -> int {
        for (size_t i = 0; i < pl.tok.size(); i++) if (pl.tok[i] == t) return (int)i;
        pl.tok.push_back(t); pl.bias.push_back(0.0f); pl.cnt.push_back(0);
        return (int)pl.tok.size() - 1;
    };
    std::vector<int32_t> biased;
    for (const auto &lb : p_.logit_bias) {
        if (lb.first < 0 || lb.first >= n_vocab) continue;
        if (std::find(biased.begin(), biased.end(), lb.first) != biased.end()) return false;
        biased.push_back(lb.first);
        pl.bias[(size_t)slot(lb.first)] = lb.second;
    }
    if (!prev_.empty() && (p_.penalty_repeat != 1.0f || p_.penalty_freq != 0.0f || p_.penalty_present != 0.0f))
        for (int32_t t : prev_) if (t >= 0 && t < n_vocab) pl.cnt[(size_t)slot(t)]++;
    return pl.tok.size() <= (size_t)max_adj;
}

// the rest of the chain on the candidates the head produced ((token, adjusted logit), best first)
int32_t Sampler::finish(std::vector<TokenProb> &c) {
    const size_t min_keep = (size_t)std::max(p_.min_keep, 1);
    if (c.empty()) { cand_.clear(); return -1; }
    if (p_.temp <= 0.0f) {
        const int best = c[0].tok;
        cand_.clear();
        if (p_.n_probs > 0) { softmax_sorted(c); cand_ = c; }
        else cand_.push_back({best, 1.0f});
        return best;
    }
    // the draw itself is the standard library's std::discrete_distribution over the final candidates' probabilities on a std::mt19937 seeded with the request's
    // seed - the class and generator llama_sampler_dist uses upstream (a draw consumes two 32-bit outputs: generate_canonical<double, 53>)
    auto draw = [&](const std::vector<TokenProb> &cc) -> size_t {
        probs_.resize(cc.size());
        for (size_t i = 0; i < cc.size(); i++) probs_[i] = cc[i].p;
        std::discrete_distribution<int> dist(probs_.begin(), probs_.end());
        return (size_t)dist(rng_);
    };
    softmax_sorted(c);
    // typical_p
    if (p_.typ_p < 1.0f && c.size() > 1) {
        double ent = 0.0;
        for (const auto &e : c) if (e.p > 0) ent -= (double)e.p * log((double)e.p);
        std::vector<std::pair<float, size_t>> dev;
        for (size_t i = 0; i < c.size(); i++) dev.emplace_back((float)fabs(-log((double)std::max(c[i].p, 1e-30f)) - ent), i);
        std::sort(dev.begin(), dev.end());
        double cum = 0.0;
        size_t last = dev.size();
        for (size_t i = 0; i < dev.size(); i++) { cum += c[dev[i].second].p; if (cum > p_.typ_p && i + 1 >= min_keep) { last = i + 1; break; } }
        std::vector<TokenProb> nc;
        for (size_t i = 0; i < last; i++) nc.push_back(c[dev[i].second]);
        std::sort(nc.begin(), nc.end(), [](const TokenProb &a, const TokenProb &b) { return a.p > b.p; });
        double s = 0; for (auto &e : nc) s += e.p; for (auto &e : nc) e.p = (float)(e.p / s);
        c.swap(nc);
    }
    // top_p
    if (p_.top_p < 1.0f) {
        double cum = 0.0;
        size_t last = c.size();
        for (size_t i = 0; i < c.size(); i++) { cum += c[i].p; if (cum >= p_.top_p && i + 1 >= min_keep) { last = i + 1; break; } }
        c.resize(last);
        double s = 0; for (auto &e : c) s += e.p; for (auto &e : c) e.p = (float)(e.p / s);
    }
    // min_p
    if (p_.min_p > 0.0f && !c.empty()) {
        const float thr = c[0].p * p_.min_p;
        size_t last = c.size();
        for (size_t i = 0; i < c.size(); i++) if (c[i].p < thr && i >= min_keep) { last = i; break; }
        c.resize(std::max<size_t>(last, 1));
        double s = 0; for (auto &e : c) s += e.p; for (auto &e : c) e.p = (float)(e.p / s);
    }
    // temperature on the surviving candidates: p_i ^ (1/T), renormalised (equivalent to scaling logits).  With
    // dynatemp_range > 0 the temperature follows the normalised entropy of the candidates:
    // T = T_min + (T_max - T_min) * (H / ln n) ^ exponent, T_min = max(0, temp - range), T_max = temp + range
    float temp = p_.temp;
    if (p_.dynatemp_range > 0.0f && c.size() > 1) {
        const double t_min = std::max(0.0, (double)p_.temp - (double)p_.dynatemp_range), t_max = (double)p_.temp + (double)p_.dynatemp_range;
        double ent = 0.0;
        for (const auto &e : c) if (e.p > 0) ent -= (double)e.p * log((double)e.p);
        const double norm = ent / log((double)c.size());
        temp = (float)(t_min + (t_max - t_min) * pow(std::max(norm, 0.0), (double)p_.dynatemp_exponent));
    }
    if (temp != 1.0f && temp > 0.0f) {
        double s = 0;
        for (auto &e : c) { e.p = (float)pow((double)e.p, 1.0 / (double)temp); s += e.p; }
        for (auto &e : c) e.p = (float)(e.p / s);
    } else if (temp <= 0.0f) {                                   // entropy drove the temperature to zero: the most likely candidate
        c.resize(1); c[0].p = 1.0f;
    }
    cand_ = c;
    return c[draw(c)].tok;
}

}  // namespace mi355
