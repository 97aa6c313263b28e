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
}

void Sampler::accept(int32_t token) {
    prev_.push_back(token);
    const size_t keep = (size_t)std::max(p_.penalty_last_n, 0);
    if (keep == 0) prev_.clear();
    else if (prev_.size() > keep) prev_.erase(prev_.begin(), prev_.begin() + (long)(prev_.size() - keep));
}

static void softmax_sorted(std::vector<TokenProb> &c) {   // c sorted by logit desc; p holds logits on entry
    if (c.empty()) return;
    const float mx = c[0].p;
    double sum = 0.0;
    for (auto &e : c) { e.p = expf(e.p - mx); sum += e.p; }
    for (auto &e : c) e.p = (float)(e.p / sum);
}

int32_t Sampler::sample(const float *logits, int n_vocab) {
    // working copy as (token, logit)
    std::vector<TokenProb> c((size_t)n_vocab);
    for (int i = 0; i < n_vocab; i++) c[(size_t)i] = {i, logits[i]};
    for (const auto &lb : p_.logit_bias)
        if (lb.first >= 0 && lb.first < n_vocab) c[(size_t)lb.first].p += lb.second;
    // penalties over the last penalty_last_n accepted tokens
    if (!prev_.empty() && (p_.penalty_repeat != 1.0f || p_.penalty_freq != 0.0f || p_.penalty_present != 0.0f)) {
        std::unordered_map<int32_t, int> cnt;
        for (int32_t t : prev_) cnt[t]++;
        for (const auto &kv : cnt) {
            if (kv.first < 0 || kv.first >= n_vocab) continue;
            float &l = c[(size_t)kv.first].p;
            if (l <= 0) l *= p_.penalty_repeat; else l /= p_.penalty_repeat;
            l -= (float)kv.second * p_.penalty_freq + (kv.second > 0 ? p_.penalty_present : 0.0f);
        }
    }
    const size_t min_keep = (size_t)std::max(p_.min_keep, 1);
    if (p_.temp <= 0.0f) {   // greedy: first maximum wins
        int best = 0;
        for (int i = 1; i < n_vocab; i++) if (c[(size_t)i].p > c[(size_t)best].p) best = i;
        const size_t np = (size_t)std::max(p_.n_probs, 0);
        cand_.clear();
        if (np > 0) {
            std::partial_sort(c.begin(), c.begin() + (long)std::min(np, c.size()), c.end(), [](const TokenProb &a, const TokenProb &b) { return a.p > b.p || (a.p == b.p && a.tok < b.tok); });
            c.resize(std::min(np, c.size()));
            softmax_sorted(c);
            cand_ = c;
        } else {
            cand_.push_back({best, 1.0f});
        }
        return best;
    }
    // top_k (also establishes descending order)
    size_t k = p_.top_k <= 0 ? c.size() : std::min<size_t>((size_t)p_.top_k, c.size());
    k = std::max(k, min_keep);
    k = std::min(k, c.size());
    auto by_logit = [](const TokenProb &a, const TokenProb &b) { return a.p > b.p || (a.p == b.p && a.tok < b.tok); };
    std::partial_sort(c.begin(), c.begin() + (long)k, c.end(), by_logit);
    c.resize(k);
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
    // temperature on the surviving candidates: p_i ^ (1/T), renormalised (equivalent to scaling logits)
    if (p_.temp != 1.0f) {
        double s = 0;
        for (auto &e : c) { e.p = (float)pow((double)e.p, 1.0 / (double)p_.temp); s += e.p; }
        for (auto &e : c) e.p = (float)(e.p / s);
    }
    cand_ = c;
    std::uniform_real_distribution<double> u(0.0, 1.0);
    const double r = u(rng_);
    double cum = 0.0;
    for (const auto &e : c) { cum += e.p; if (r < cum) return e.tok; }
    return c.back().tok;
}

}  // namespace mi355
