// tests/host/host_tests.cc — CPU unit tests of the host-side mirror (JSON, tokenizer, sampler, slot loop, engine façade)
// against a deterministic fake arithmetic backend.  Built and run by tests/test_host_logic.py with g++ (no HIP, no GPU).
#include <cassert>
#include <csignal>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
#include <map>
#include <random>
#include <set>
#include <unistd.h>
#include <cstdlib>
#include <string>

#include "../../cortex.llamacpp_amd/host/engine.h"
#include "../../cortex.llamacpp_amd/host/gguf.h"
#include "../../cortex.llamacpp_amd/host/json.h"
#include "../../cortex.llamacpp_amd/host/sampling.h"
#include "../../cortex.llamacpp_amd/host/grammar.h"
#include "../../cortex.llamacpp_amd/host/server_context.h"
#include "../../cortex.llamacpp_amd/host/vocab.h"

using namespace mi355;

static int g_fail = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
    } while (0)

// ---------------------------------------------------------------- fake backend
static Vocab make_vocab() {
    std::vector<std::string> toks = {"<unk>", "<s>", "</s>"};
    std::vector<float> sc = {0, 0, 0};
    std::vector<int> ty = {TT_UNKNOWN, TT_CONTROL, TT_CONTROL};
    for (int b = 0; b < 256; b++) { char buf[8]; snprintf(buf, sizeof buf, "<0x%02X>", b); toks.push_back(buf); sc.push_back(0); ty.push_back(TT_BYTE); }
    const char *pieces[] = {"\xE2\x96\x81", "h", "e", "l", "o", "w", "r", "d", "s", "t", "a", "he", "ll", "hell", "hello", "\xE2\x96\x81hello", "wo", "wor", "worl",
                            "world", "\xE2\x96\x81world", "\xE2\x96\x81s", "\xE2\x96\x81st", "\xE2\x96\x81sto", "p", "\xE2\x96\x81stop", "U", "S", "E", "R", ":", "A", "I", "N", "T", "'", "!", ".", ","};
    float s = -1.0f;
    for (const char *p : pieces) { toks.push_back(p); sc.push_back(strlen(p) > 3 ? -s : s); ty.push_back(TT_NORMAL); s -= 1.0f; }
    // longer pieces get higher scores so merges prefer them
    for (size_t i = 259; i < toks.size(); i++) sc[i] = (float)toks[i].size();
    Vocab v;
    v.init_spm(toks, sc, ty, 1, 2, 0, true);
    return v;
}

struct FakeBackend : IBackend {
    Vocab voc = make_vocab();
    int ctx = 256, nbatch = 64;
    int eos_after = -1;           // emit EOS as the preferred token once a sequence reaches this position
    int decode_sleep_us = 0;
    int fail_decode_over = -1;    // decode of more than this many tokens returns 1 (no KV slot)
    std::vector<std::vector<int32_t>> calls_tokens, calls_seq, calls_pos;
    std::vector<std::string> kv_ops;
    std::map<int, std::map<int, int>> kv;   // seq -> pos -> token
    std::vector<std::vector<float>> last_logits;
    std::vector<int> last_flag_index;
    int first_normal = 259;

    int n_ctx() const override { return ctx; }
    int n_batch() const override { return nbatch; }
    int n_ubatch() const override { return nbatch; }
    int n_vocab() const override { return voc.n_tokens(); }
    int n_embd() const override { return 8; }
    const Vocab &vocab() const override { return voc; }
    int next_token(int tok, int pos) const {
        if (eos_after >= 0 && pos >= eos_after) return voc.eos();
        const int nn = voc.n_tokens() - first_normal;
        return first_normal + (int)(((unsigned)tok * 31u + (unsigned)pos * 7u + 11u) % (unsigned)nn);
    }
    int hard_fail = 0;                 // decode returns -1 with this message (a backend failure, not a full cache)
    const char *last_error() const override { return hard_fail ? "device lost (test)" : ""; }
    int decode(const BatchView &b) override {
        if (hard_fail) return -1;
        if (fail_decode_over >= 0 && b.n_tokens > fail_decode_over) return 1;
        if (decode_sleep_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(decode_sleep_us));
        order.push_back("t" + std::to_string(b.n_tokens));
        calls_tokens.emplace_back(b.token, b.token + b.n_tokens);
        calls_seq.emplace_back(b.seq_id, b.seq_id + b.n_tokens);
        calls_pos.emplace_back(b.pos, b.pos + b.n_tokens);
        last_logits.clear(); last_flag_index.assign((size_t)b.n_tokens, -1);
        for (int i = 0; i < b.n_tokens; i++) {
            kv[b.seq_id[i]][b.pos[i]] = b.token[i];
            if ((int)kv[b.seq_id[i]].size() > ctx) return 1;
            if (b.logits[i]) {
                std::vector<float> lg((size_t)n_vocab());
                for (int v = 0; v < n_vocab(); v++) lg[(size_t)v] = (float)((v * 2654435761u >> 20) & 1023) / 1024.0f;   // background in [0,1)
                lg[(size_t)next_token(b.token[i], b.pos[i])] = 12.0f;
                last_flag_index[(size_t)i] = (int)last_logits.size();
                last_logits.push_back(std::move(lg));
            }
        }
        return 0;
    }
    bool embd_on = false;
    int pooling = 0;
    int pooling_type() const override { return pooling; }
    std::vector<std::vector<float>> embd_rows;
    std::vector<float> embd_row;
    void set_embeddings(bool on) override { embd_on = on; embd_flags.push_back(on); }
    std::vector<bool> embd_flags;
    const float *embeddings_ith(int i) override {
        if (!embd_on || i < 0 || i >= (int)last_flag_index.size() || last_flag_index[(size_t)i] < 0) return nullptr;
        embd_row.assign((size_t)n_embd(), 0.0f);
        const auto &tk = calls_tokens.back();
        for (int k = 0; k < n_embd(); k++) embd_row[(size_t)k] = (float)((tk[(size_t)i] + 3 * k) % 7) - 3.0f;   // depends on the row's token
        return embd_row.data();
    }
    const float *logits_ith(int i) override { return (i >= 0 && i < (int)last_flag_index.size() && last_flag_index[(size_t)i] >= 0) ? last_logits[(size_t)last_flag_index[(size_t)i]].data() : nullptr; }
    // (the op log is written by the loop thread - an idle loop clears the cache on its own - and read / cleared by the test's thread: under a mutex)
    std::mutex ops_mu;
    void log_op(std::string op) { std::lock_guard<std::mutex> lk(ops_mu); kv_ops.push_back(std::move(op)); }
    void clear_ops() { std::lock_guard<std::mutex> lk(ops_mu); kv_ops.clear(); }
    std::vector<std::string> ops_snapshot() { std::lock_guard<std::mutex> lk(ops_mu); return kv_ops; }
    void kv_clear() override { kv.clear(); log_op("clear"); }
    bool kv_seq_rm(int seq, int p0, int p1) override {
        log_op("rm " + std::to_string(seq) + " " + std::to_string(p0) + " " + std::to_string(p1));
        auto &m = kv[seq];
        for (auto it = m.begin(); it != m.end();) { if (it->first >= (p0 < 0 ? 0 : p0) && (p1 < 0 || it->first < p1)) it = m.erase(it); else ++it; }
        return true;
    }
    void kv_seq_add(int seq, int p0, int p1, int d) override {
        log_op("add " + std::to_string(seq) + " " + std::to_string(p0) + " " + std::to_string(p1) + " " + std::to_string(d));
        std::map<int, int> nm;
        for (auto &e : kv[seq]) nm[(e.first >= p0 && e.first < p1) ? e.first + d : e.first] = e.second;
        kv[seq] = nm;
    }
    void kv_seq_cp(int, int, int, int) override {}
    // ---- images (test_image_requests): "IMG" + a byte n = an image of n embedding rows; anything else does not decode
    bool mm = false;
    bool multimodal() const override { return mm; }
    bool image_check(const uint8_t *b, size_t n, std::string &err) override { if (n == 4 && !memcmp(b, "IMG", 3) && b[3] > 0) return true; err = "not an image"; return false; }
    int embed_fail_for = -1;
    int image_embed(const uint8_t *b, size_t n, std::vector<float> &rows, std::string &err) override {
        if (!image_check(b, n, err)) return -1;
        if ((int)b[3] == embed_fail_for) { err = "encoder failed (test)"; return -1; }
        rows.assign((size_t)b[3] * (size_t)n_embd(), (float)b[3]);
        return (int)b[3];
    }
    struct EmbdCall { int n, pos0, seq; float first; };
    std::vector<EmbdCall> embd_calls;
    std::vector<std::string> order;       // "t<n>" a token batch of n, "e<n>" an embedding batch of n: what reached the model, in order
    int decode_embd(const float *rows, int n, int pos0, int seq) override {
        embd_calls.push_back({n, pos0, seq, rows[0]});
        order.push_back("e" + std::to_string(n));
        for (int i = 0; i < n; i++) kv[seq][pos0 + i] = -1;
        return 0;
    }
};

// ---------------------------------------------------------------- tests
static void test_json() {
    Json j;
    std::string err;
    CHECK(Json::parse("{\"a\": [1, 2.5, \"x\\n\\u00e9\", true, null], \"b\": {\"c\": -3}}", j, &err));
    CHECK(j["a"].size() == 5 && j["a"].at(0).as_int() == 1 && j["a"].at(1).as_double() == 2.5);
    CHECK(j["a"].at(2).as_string() == "x\n\xC3\xA9");
    CHECK(j["b"].value<int>("c", 0) == -3 && j.value<int>("zz", 7) == 7);
    Json k;
    CHECK(Json::parse(j.dump(), k) && k.dump() == j.dump());
    CHECK(!Json::parse("{\"a\": }", k));
    CHECK(Json(std::string("a\xC1z\xE2\x96")).dump() == "\"a\xEF\xBF\xBDz\xEF\xBF\xBD\xEF\xBF\xBD\"");   // ill-formed bytes -> U+FFFD
    CHECK(Json(std::string("\xE2\x96\x81ok")).dump() == "\"\xE2\x96\x81ok\"");
}

static void test_vocab() {
    Vocab v = make_vocab();
    auto ids = v.tokenize("hello world", true);
    CHECK(ids.size() == 3 && ids[0] == v.bos());
    CHECK(v.token_to_piece(ids[1]) == " hello" && v.token_to_piece(ids[2]) == " world");
    CHECK(v.detokenize({ids[1], ids[2]}) == " hello world");
    auto z = v.tokenize("h\xC3\xA9", false);               // 'é' is not in the vocab -> two byte tokens
    CHECK(z.size() == 4);                                   // "▁" "h" <0xC3> <0xA9>
    CHECK(v.detokenize(z) == " h\xC3\xA9");
    auto sp = v.tokenize("hello</s>", false, true);         // parse_special splits on the control token
    CHECK(!sp.empty() && sp.back() == v.eos());
    CHECK(v.is_eog(v.eos()) && !v.is_eog(ids[1]));
}

// WordPiece as llama.cpp's WPM tokenizer runs it (vocab.h): the reference's embedding smoke model carries a "bert" vocabulary.  Known answers worked out by hand
// from the algorithm: lower-casing, accent stripping, punctuation and CJK as words of their own, greedy longest match with the U+2581 word prefix, whole-word
// fallback to [UNK], [CLS] ... [SEP] around the text.
static void test_vocab_wordpiece() {
#define W "\xE2\x96\x81"
    const std::vector<std::string> toks = {"[PAD]", "[UNK]", "[CLS]", "[SEP]", W "hello", W "world", W "un", "aff", "able", W ",", W "!", W "cafe", W "a", "b", "##x",
                                           W "\xE4\xB8\xAD", W "\xE6\x96\x87", W "h", "ello", W "wor", "ld", W "naive"};
#undef W
    Vocab v;
    v.init_wpm(toks, 2, 3, 1);
    typedef std::vector<int32_t> I;
    CHECK((v.tokenize("Hello, WORLD!", true) == I{2, 4, 9, 5, 10, 3}));                 // lower-cased; the comma and the bang are words
    CHECK((v.tokenize("hello", false) == I{4}));                                        // longest match first: not "h" + "ello"
    CHECK((v.tokenize("unaffable", false) == I{6, 7, 8}));                              // word-initial piece, then bare continuations
    CHECK((v.tokenize("unaffablex", false) == I{1}));                                   // an uncovered rest turns the WHOLE word into [UNK]
    CHECK((v.tokenize("  hello\t\nworld  ", false) == I{4, 5}));                        // any whitespace separates, none survives
    CHECK((v.tokenize("Caf\xC3\xA9 na\xC3\xAFve", false) == I{11, 21}));                 // accents stripped (e-acute, i-diaeresis), lower-cased
    CHECK((v.tokenize("cafe\xCC\x81", false) == I{11}));                                 // a combining acute accent after the letter is dropped too
    CHECK((v.tokenize("\xE4\xB8\xAD\xE6\x96\x87", false) == I{15, 16}));                 // CJK ideographs are words of their own
    CHECK((v.tokenize("ab", false) == I{12, 13}) && (v.tokenize("ba", false) == I{1}));  // "b" exists only as a continuation
    CHECK((v.tokenize("", true) == I{2, 3}));
    CHECK(v.bos() == 2 && v.eos() == 3 && v.add_bos());
}

static void test_sampler() {
    std::vector<float> lg(100, 0.0f);
    lg[17] = 5.0f; lg[42] = 4.9f; lg[3] = 4.0f;
    SamplingParams p;
    p.temp = 0.0f;
    Sampler g(p);
    CHECK(g.sample(lg.data(), 100) == 17);
    p.temp = 0.8f; p.top_k = 1;
    Sampler k1(p);
    CHECK(k1.sample(lg.data(), 100) == 17);
    p.top_k = 40; p.seed = 123;
    Sampler a(p), b(p);
    for (int i = 0; i < 20; i++) { const int x = a.sample(lg.data(), 100), y = b.sample(lg.data(), 100); CHECK(x == y); a.accept(x); b.accept(y); }
    p.temp = 0.0f; p.penalty_repeat = 2.0f; p.penalty_last_n = 8;
    Sampler pen(p);
    pen.accept(17);
    CHECK(pen.sample(lg.data(), 100) == 42);                // 5.0 / 2 < 4.9
    p.penalty_repeat = 1.0f; p.logit_bias = {{17, -INFINITY}};
    Sampler lb(p);
    CHECK(lb.sample(lg.data(), 100) == 42);
    p.logit_bias.clear(); p.temp = 1.0f; p.top_k = 0; p.top_p = 0.5f; p.min_p = 0.0f; p.seed = 7;
    Sampler tp(p);
    std::set<int> seen;
    for (int i = 0; i < 200; i++) seen.insert(tp.sample(lg.data(), 100));
    CHECK(seen.size() <= 2 && seen.count(17));              // nucleus of 0.5 keeps the two dominant tokens at most
}

// mirostat 1 / 2, dynamic temperature and the -1 penalty window (upstream common/sampling.cpp chains)
static void test_sampler_mirostat_dynatemp() {
    const int V = 2000;
    std::vector<float> lg((size_t)V);
    for (int i = 0; i < V; i++) lg[(size_t)i] = -0.01f * (float)((i * 37) % V);          // a long smooth tail
    lg[5] = 3.0f; lg[9] = 2.5f;
    for (int mode = 1; mode <= 2; mode++) {
        SamplingParams p;
        p.mirostat = mode; p.mirostat_tau = 3.0f; p.mirostat_eta = 0.2f; p.temp = 1.0f; p.seed = 11;
        p.top_k = 1;                                           // ignored by the mirostat chains: with it the draw would always be 5
        Sampler a(p), b(p);
        std::set<int> seen;
        double surprise = 0.0;
        const int N = 400;
        for (int i = 0; i < N; i++) {
            const int x = a.sample(lg.data(), V), y = b.sample(lg.data(), V);
            CHECK(x == y);                                    // seeded: reproducible
            CHECK(x >= 0 && x < V);
            seen.insert(x);
            const auto &c = a.candidates();
            CHECK(!c.empty());
            double ps = 0; for (const auto &e : c) ps += e.p;
            CHECK(fabs(ps - 1.0) < 1e-3);
            for (const auto &e : c) if (e.tok == x) surprise += -log2((double)e.p) / N;
            a.accept(x); b.accept(y);
        }
        CHECK(seen.size() > 3);                                // not collapsed to top_k = 1
        (void)surprise;
    }
    {   // the truncation follows mu: a tiny target surprise keeps only the head of the distribution
        SamplingParams p; p.mirostat = 2; p.mirostat_tau = 0.05f; p.mirostat_eta = 0.0f; p.temp = 1.0f; p.seed = 3;
        Sampler s(p);                                          // mu = 2 tau = 0.1 bit: only tokens with p > 0.93 pass, else the single best
        for (int i = 0; i < 50; i++) CHECK(s.sample(lg.data(), V) == 5);
    }
    {   // dynamic temperature: range 0 is the plain chain (same draws); a flat pair of candidates gets T_max, a peaked one less
        std::vector<float> l2(50, -20.0f); l2[1] = 2.0f; l2[2] = 2.0f; l2[3] = 1.9f;
        SamplingParams p; p.temp = 0.7f; p.top_k = 3; p.top_p = 1.0f; p.min_p = 0.0f; p.seed = 5;
        Sampler plain(p);
        p.dynatemp_range = 0.0f;
        Sampler zero(p);
        for (int i = 0; i < 30; i++) CHECK(plain.sample(l2.data(), 50) == zero.sample(l2.data(), 50));
        p.dynatemp_range = 0.6f; p.dynatemp_exponent = 1.0f;
        Sampler dyn(p);
        (void)dyn.sample(l2.data(), 50);
        const auto c_dyn = dyn.candidates();
        (void)plain.sample(l2.data(), 50);
        const auto c_pl = plain.candidates();
        CHECK(c_dyn.size() == 3 && c_pl.size() == 3);
        // near-uniform candidates: entropy ~ max -> temperature ~ temp + range = 1.3 > 0.7: flatter than the plain chain
        CHECK(c_dyn[2].p > c_pl[2].p);
        p.temp = 0.3f; p.dynatemp_range = 0.3f;                // T_min = 0: a one-sided distribution drives it towards greedy
        std::vector<float> l3(50, -30.0f); l3[7] = 10.0f; l3[8] = 0.0f;
        Sampler d2(p);
        for (int i = 0; i < 20; i++) CHECK(d2.sample(l3.data(), 50) == 7);
    }
    {   // the seeded draw IS std::discrete_distribution over the candidates' probabilities on std::mt19937(seed) - what llama_sampler_dist does: an
        // independent generator fed the sampler's own candidate lists must name the same tokens, draw after draw
        std::vector<float> l4(64);
        for (int i = 0; i < 64; i++) l4[i] = 0.05f * (float)((i * 37) % 64);
        SamplingParams p; p.temp = 0.9f; p.top_k = 12; p.top_p = 0.95f; p.min_p = 0.01f; p.seed = 4242;
        Sampler s(p);
        std::mt19937 ref(4242);
        for (int i = 0; i < 64; i++) {
            const int tok = s.sample(l4.data(), 64);
            const auto c = s.candidates();
            std::vector<float> pr;
            for (const auto &e : c) pr.push_back(e.p);
            std::discrete_distribution<int> dist(pr.begin(), pr.end());
            CHECK(c[(size_t)dist(ref)].tok == tok);
        }
    }
    {   // repeat_last_n = -1: the window is the context, not "off"
        std::vector<float> l4(100, 0.0f); l4[17] = 5.0f; l4[42] = 4.9f;
        SamplingParams p; p.temp = 0.0f; p.penalty_repeat = 2.0f; p.penalty_last_n = -1; p.penalty_n_ctx = 256;
        Sampler s(p);
        s.accept(17);
        for (int i = 0; i < 100; i++) s.accept(60 + (i % 30));  // 17 is 100 tokens back: outside a 64 window, inside the context
        CHECK(s.sample(l4.data(), 100) == 42);
        p.penalty_last_n = 0;                                  // 0 = no penalties at all
        Sampler off(p);
        off.accept(17);
        CHECK(off.sample(l4.data(), 100) == 17);
    }
}

// The reference counts n_decoded when a sampled token is fed back (llama_server_context.cc:1335), so the budget check
// (:787) fires on the (n_predict+1)-th sampled token: n_predict = n yields n + 1 pieces of text and tokens_predicted = n.
static void test_sampler_topk_matches_full_sort() {
    // the one-pass heap selection with sparse overrides against a full sort of the modified row, with ties and
    // overrides that move tokens both ways across the cut
    std::mt19937 rng(7);
    const int V = 5000;
    for (int trial = 0; trial < 20; trial++) {
        std::vector<float> lg((size_t)V);
        for (auto &x : lg) x = (float)((int)(rng() % 2001) - 1000) / 64.0f;      // coarse grid: many exact ties
        SamplingParams p;
        p.temp = 1.0f; p.top_k = 1 + (int)(rng() % 60); p.top_p = 1.0f; p.min_p = 0.0f; p.seed = 5;
        p.penalty_repeat = 1.3f; p.penalty_freq = 0.1f; p.penalty_present = 0.2f; p.penalty_last_n = 16;
        for (int b = 0; b < 5; b++) p.logit_bias.push_back({(int)(rng() % V), (float)((int)(rng() % 41) - 20)});
        Sampler s(p);
        std::vector<int> prev;
        for (int i = 0; i < 20; i++) { const int t = (int)(rng() % V); s.accept(t); prev.push_back(t); }
        if (prev.size() > 16) prev.erase(prev.begin(), prev.end() - 16);
        std::vector<float> mod = lg;
        for (const auto &lb : p.logit_bias) mod[(size_t)lb.first] += lb.second;
        std::map<int, int> cnt;
        for (int t : prev) cnt[t]++;
        for (const auto &kv : cnt) {
            float &l = mod[(size_t)kv.first];
            if (l <= 0) l *= p.penalty_repeat; else l /= p.penalty_repeat;
            l -= (float)kv.second * p.penalty_freq + p.penalty_present;
        }
        std::vector<int> order((size_t)V);
        for (int i = 0; i < V; i++) order[(size_t)i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return mod[(size_t)a] > mod[(size_t)b] || (mod[(size_t)a] == mod[(size_t)b] && a < b); });
        s.sample(lg.data(), V);
        const auto &c = s.candidates();
        CHECK((int)c.size() == p.top_k);
        for (size_t i = 0; i < c.size(); i++) CHECK(c[i].tok == order[i]);
        SamplingParams g = p; g.temp = 0.0f;
        Sampler sg(g);
        for (int t : prev) sg.accept(t);
        CHECK(sg.sample(lg.data(), V) == order[0]);
    }
}

static std::string expected_text(const FakeBackend &be, const std::vector<int32_t> &prompt, int n_predict) {
    const int n = n_predict + 1;
    std::string s;
    int tok = prompt.back(), pos = (int)prompt.size() - 1;
    for (int i = 0; i < n; i++) { tok = be.next_token(tok, pos); pos++; s += be.voc.token_to_piece(tok); }
    return s;
}

// The head of the chain (logit_bias -> penalties -> top_k) computed apart - what mi355_get_topk_ith does on the device, restated here with the same f32
// operations - followed by Sampler::finish must give the token and the candidates of Sampler::sample on the whole row: same parameters, same seed, same
// history; ties in the logits included.
static void test_sampler_front_plan_matches_full_chain() {
    const int V = 3000;
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.0f, 3.0f);
    auto better = [](const TokenProb &a, const TokenProb &b) { return a.p > b.p || (a.p == b.p && a.tok < b.tok); };
    int planned = 0;
    for (int trial = 0; trial < 60; trial++) {
        SamplingParams sp;
        sp.seed = 1234u + (uint32_t)trial;
        sp.top_k = trial % 7 == 0 ? 1 : 5 + trial % 60;
        sp.top_p = trial % 3 ? 0.9f : 1.0f; sp.min_p = trial % 4 ? 0.05f : 0.0f; sp.typ_p = trial % 5 == 0 ? 0.8f : 1.0f;
        sp.temp = trial % 6 == 0 ? 0.0f : 0.7f + 0.01f * (float)trial;
        sp.n_probs = trial % 4 == 0 ? 5 : 0;
        if (trial % 2) { sp.penalty_repeat = 1.15f; sp.penalty_freq = 0.1f; sp.penalty_present = 0.2f; sp.penalty_last_n = 32; }
        if (trial % 3 == 0) { sp.logit_bias.push_back({7, 2.5f}); sp.logit_bias.push_back({11, -INFINITY}); sp.logit_bias.push_back({V + 5, 1.0f}); }
        std::vector<float> lg((size_t)V);
        for (auto &x : lg) x = nd(rng);
        lg[100] = lg[200] = lg[50] = 9.5f;                       // a three-way tie near the top: the lowest id ranks first
        Sampler a(sp), b(sp);
        for (int h = 0; h < 40; h++) { const int32_t t = (int32_t)(rng() % 300); a.accept(t); b.accept(t); }
        const int32_t id_full = a.sample(lg.data(), V);
        Sampler::FrontPlan fp;
        CHECK(b.plan_front(V, 128, 192, fp));
        planned++;
        // the device's part: adjustments on the listed tokens, then the k best of the row
        std::vector<float> adj = lg;
        for (size_t j = 0; j < fp.tok.size(); j++) {
            float l = adj[(size_t)fp.tok[j]] + fp.bias[j];
            if (fp.cnt[j] > 0) {
                if (l <= 0.0f) l *= sp.penalty_repeat; else l /= sp.penalty_repeat;
                l -= (float)fp.cnt[j] * sp.penalty_freq + sp.penalty_present;
            }
            adj[(size_t)fp.tok[j]] = l;
        }
        std::vector<TokenProb> all((size_t)V);
        for (int v = 0; v < V; v++) all[(size_t)v] = TokenProb{v, adj[(size_t)v]};
        std::partial_sort(all.begin(), all.begin() + fp.k, all.end(), better);
        std::vector<TokenProb> c(all.begin(), all.begin() + fp.k);
        const int32_t id_front = b.finish(c);
        CHECK(id_front == id_full);
        CHECK(a.candidates().size() == b.candidates().size());
        for (size_t j = 0; j < a.candidates().size() && j < b.candidates().size(); j++)
            CHECK(a.candidates()[j].tok == b.candidates()[j].tok && a.candidates()[j].p == b.candidates()[j].p);
    }
    CHECK(planned == 60);
    // what keeps the whole row on the host: mirostat, top_k off, two biases on one token, a penalty window with more distinct tokens than the device takes
    SamplingParams m; m.mirostat = 2;
    Sampler::FrontPlan fp;
    CHECK(!Sampler(m).plan_front(V, 128, 192, fp));
    SamplingParams nk; nk.top_k = 0;
    CHECK(!Sampler(nk).plan_front(V, 128, 192, fp));
    SamplingParams two; two.logit_bias.push_back({3, 1.0f}); two.logit_bias.push_back({3, 2.0f});
    CHECK(!Sampler(two).plan_front(V, 128, 192, fp));
    SamplingParams win; win.penalty_repeat = 1.1f; win.penalty_last_n = 400;
    Sampler w(win);
    for (int h = 0; h < 400; h++) w.accept(h);
    CHECK(!w.plan_front(V, 128, 192, fp));
}

// ---------------------------------------------------------------- grammars (host/grammar.h)
static bool g_accepts(const std::shared_ptr<const Grammar> &g, const std::string &text, bool *could_end = nullptr) {
    GrammarMatcher m(g);
    for (const char c : text) if (!m.accept(std::string(1, c))) return false;      // byte by byte: multi-byte characters cross calls
    if (could_end) *could_end = m.can_end();
    return true;
}
static bool g_sentence(const std::shared_ptr<const Grammar> &g, const std::string &text) { bool e = false; return g_accepts(g, text, &e) && e; }

static void test_grammar_parse_and_match() {
    std::string err;
    auto arith = Grammar::parse(
        "# sums and products\n"
        "root ::= expr\n"
        "expr ::= term ([-+*/] term)*\n"
        "term ::= num | \"(\" expr \")\"\n"
        "num  ::= [0-9]+\n", err);
    CHECK(arith && err.empty());
    CHECK(g_sentence(arith, "1+(23*4)") && g_sentence(arith, "7") && g_sentence(arith, "((1))"));
    bool e = true;
    CHECK(g_accepts(arith, "1+", &e) && !e);                      // a prefix, not a sentence
    CHECK(g_accepts(arith, "(1", &e) && !e);
    CHECK(!g_accepts(arith, "1+)") && !g_accepts(arith, ")") && !g_accepts(arith, "1 + 2"));
    { GrammarMatcher m(arith); CHECK(m.admits("12") && m.admits("(") && !m.admits("+") && !m.admits("") && !m.can_end()); CHECK(m.accept("12+3") && m.can_end() && m.admits("*(") && !m.admits(")")); }
    // repetition forms, bounded and not, on literals, classes and groups
    auto rep = Grammar::parse("root ::= \"ab\"{2,3} [x-z]? (\"-\" [0-9]{2}){0,2} \"!\"+\n", err);
    CHECK(rep);
    CHECK(g_sentence(rep, "abab!") && g_sentence(rep, "abababz-12-34!!!") && g_sentence(rep, "ababx!"));
    CHECK(!g_sentence(rep, "ab!") && !g_accepts(rep, "abababab") && !g_accepts(rep, "abab-1!") && !g_accepts(rep, "abab-12-34-56") && !g_sentence(rep, "abab"));
    // any character, negated classes, escapes, characters outside ASCII (in literals, in classes, split across pieces)
    auto uni = Grammar::parse("root ::= \"caf\\u00e9\" . [^a-z\\n] [\\x41-\\x43] \"\xE2\x86\x92\" [\xCE\xB1-\xCF\x89]+\n", err);
    CHECK(uni);
    CHECK(g_sentence(uni, "caf\xC3\xA9\xF0\x9F\x99\x82" "7B\xE2\x86\x92\xCE\xB2\xCE\xB3"));
    CHECK(!g_accepts(uni, "cafe") && !g_accepts(uni, "caf\xC3\xA9xq") && !g_accepts(uni, "caf\xC3\xA9x7D"));
    {
        GrammarMatcher m(uni);
        CHECK(m.accept("caf") && m.admits("\xC3") && !m.admits("\xC4") && m.admits("\xC3\xA9") && !m.admits("\xC3\xA8"));
        CHECK(m.accept("\xC3") && !m.can_end() && m.admits("\xA9") && !m.admits("x") && m.accept("\xA9"));
        CHECK(m.admits("\xF0\x9F") && m.accept("\xF0\x9F") && m.accept("\x99\x82") && m.accept("7B") && m.admits("\xE2") && !m.admits("\xE3"));
        GrammarMatcher bad(uni);
        CHECK(bad.accept("caf") && !bad.accept("\xC3(") && bad.dead());          // a continuation byte was due
        // overlong forms of an admissible character are not that character: "f" as C1 A6 / E0 81 A6 / F0 80 81 A6
        GrammarMatcher ov(uni);
        CHECK(ov.accept("ca") && ov.admits("f") && !ov.admits("\xC1") && !ov.admits("\xC1\xA6") && !ov.admits("\xE0\x81") && !ov.admits("\xF0\x80\x81\xA6") && !ov.admits("\xF0\x80"));
    }
    // the usual layout of a JSON grammar: groups over several lines, alternatives starting a line inside them, comments
    auto js = Grammar::parse(
        "root   ::= object\n"
        "value  ::= object | array | string | number | (\"true\" | \"false\" | \"null\") ws\n"
        "object ::=\n"
        "  \"{\" ws (\n"
        "            string \":\" ws value\n"
        "    (\",\" ws string \":\" ws value)*\n"
        "  )? \"}\" ws\n"
        "array  ::=\n"
        "  \"[\" ws (\n"
        "            value\n"
        "    (\",\" ws value)*\n"
        "  )? \"]\" ws\n"
        "string ::=\n"
        "  \"\\\"\" (\n"
        "    [^\"\\\\\\x7F\\x00-\\x1F] |\n"
        "    \"\\\\\" ([\"\\\\bfnrt] | \"u\" [0-9a-fA-F]{4}) # escapes\n"
        "  )* \"\\\"\" ws\n"
        "number ::= (\"-\"? ([0-9] | [1-9] [0-9]{0,15})) (\".\" [0-9]+)? ([eE] [-+]? [0-9] [1-9]{0,15})? ws\n"
        "# Optional space: by convention, applied in this grammar after literal chars when allowed\n"
        "ws ::= | \" \" | \"\\n\" [ \\t]{0,20}\n", err);
    CHECK(js);
    if (!js) printf("  %s\n", err.c_str());
    CHECK(g_sentence(js, "{\"a\": [1, 2.5e3, \"x\\n\\u00e9\", true, null], \"b\": {\"c\": -3}}") && g_sentence(js, "{}"));
    CHECK(!g_accepts(js, "[1]") && !g_accepts(js, "{\"a\" 1}") && !g_accepts(js, "{'a': 1}") && !g_sentence(js, "{\"a\": 1"));
    // what cannot be a grammar
    CHECK(!Grammar::parse("root ::= root \"a\" | \"a\"\n", err) && err.find("left recursion") != std::string::npos);
    CHECK(!Grammar::parse("root ::= a\na ::= b \"x\"\nb ::= | a\n", err) && err.find("left recursion") != std::string::npos);   // through an empty alternative
    CHECK(!Grammar::parse("root ::= thing\n", err) && err.find("undefined rule thing") != std::string::npos);
    CHECK(!Grammar::parse("start ::= \"a\"\n", err) && err.find("root") != std::string::npos);
    CHECK(!Grammar::parse("root ::= \"abc\n", err) && !Grammar::parse("root ::= [a-\n", err) && !Grammar::parse("root ::= (\"a\"\n", err));
    CHECK(!Grammar::parse("root ::= *\n", err) && !Grammar::parse("root ::= \"a\"{3,2}\n", err) && !Grammar::parse("root = \"a\"\n", err) && !Grammar::parse("", err));
    CHECK(!Grammar::parse("root ::= \"a\"\nroot ::= \"b\"\n", err) && err.find("twice") != std::string::npos);
    CHECK(Grammar::parse("root ::= \"a\"", err));                                                                               // no final newline
}

static void test_json_schema_grammar() {
    auto conv = [&](const std::string &schema_text, std::string *gbnf_out = nullptr) -> std::shared_ptr<const Grammar> {
        Json schema;
        if (!schema_text.empty() && !Json::parse(schema_text, schema)) { printf("  bad test schema\n"); return nullptr; }
        std::string gbnf, err;
        if (!json_schema_to_gbnf(schema, gbnf, err)) { printf("  schema: %s\n", err.c_str()); return nullptr; }
        if (gbnf_out) *gbnf_out = gbnf;
        auto g = Grammar::parse(gbnf, err);
        if (!g) printf("  gbnf: %s\n%s\n", err.c_str(), gbnf.c_str());
        return g;
    };
    // json_object mode: no schema -> any JSON object, compact or spaced
    auto any = conv("");
    CHECK(any);
    CHECK(g_sentence(any, "{\"a\":[1,2,{\"b\":null}],\"c\":\"x\\\"y\",\"d\":-1.5e-3}") && g_sentence(any, "{ \"k\" : [ true , false ] }\n") == false);   // (nothing after the closing brace)
    CHECK(g_sentence(any, "{ \"k\" : [ true , false ] }") && g_sentence(any, "{}") && !g_accepts(any, "[1]") && !g_accepts(any, "\"s\"") && !g_sentence(any, "{\"a\":1"));
    CHECK(!g_accepts(any, "{\"a\":01}") && !g_accepts(any, "{\"a\":\"\t\"}") && !g_accepts(any, "{a:1}"));
    // an object with required and optional members, a bounded array, numbers
    auto person = conv("{\"type\":\"object\",\"properties\":{\"name\":{\"type\":\"string\"},\"age\":{\"type\":\"integer\"},\"tags\":{\"type\":\"array\",\"items\":{\"type\":\"string\"},\"maxItems\":2},"
                       "\"score\":{\"type\":\"number\"}},\"required\":[\"name\",\"age\"]}");
    CHECK(person);
    CHECK(g_sentence(person, "{\"name\":\"x\",\"age\":3}") && g_sentence(person, "{ \"name\": \"Ann\", \"age\": 41, \"tags\": [\"a\", \"b\"], \"score\": 2.5 }"));
    CHECK(g_sentence(person, "{\"name\":\"x\",\"age\":3,\"score\":1}") && g_sentence(person, "{\"name\":\"x\",\"age\":3,\"tags\":[]}"));
    CHECK(!g_sentence(person, "{\"name\":\"x\"}") && !g_accepts(person, "{\"age\":3,") && !g_accepts(person, "{\"name\":\"x\",\"age\":3.5") &&
          !g_accepts(person, "{\"name\":\"x\",\"age\":3,\"tags\":[\"a\",\"b\",") && !g_accepts(person, "{\"name\":\"x\",\"age\":3,\"other\"") &&
          !g_accepts(person, "{\"name\":\"x\",\"age\":3,\"score\":1,\"tags\""));                  // optional members keep the schema's order
    // nothing required: any subset, in order, commas only between members; additionalProperties: false with no members -> {}
    auto opt = conv("{\"type\":\"object\",\"properties\":{\"a\":{\"type\":\"boolean\"},\"b\":{\"type\":\"null\"},\"c\":{\"const\":\"k\"}}}");
    CHECK(opt);
    for (const char *ok : {"{}", "{\"a\":true}", "{\"b\":null}", "{\"c\":\"k\"}", "{\"a\":false,\"c\":\"k\"}", "{\"a\":true,\"b\":null,\"c\":\"k\"}", "{ \"b\": null , \"c\": \"k\" }"}) CHECK(g_sentence(opt, ok));
    for (const char *no : {"{,", "{\"a\":true,}", "{\"b\":null,\"a\"", "{\"c\":\"x\"", "{\"a\":1"}) CHECK(!g_sentence(opt, no) && !(g_accepts(opt, no) && std::string(no).back() != ','));
    auto closed = conv("{\"type\":\"object\",\"additionalProperties\":false}");
    CHECK(closed && g_sentence(closed, "{ }") && !g_accepts(closed, "{\"a\""));
    auto typed = conv("{\"type\":\"object\",\"additionalProperties\":{\"type\":\"integer\"}}");
    CHECK(typed && g_sentence(typed, "{\"a\":1,\"b\":-2}") && !g_accepts(typed, "{\"a\":\"x\"") );
    // enum / const / anyOf / type lists / string lengths / tuples
    auto en = conv("{\"enum\":[\"red\",\"green\",3,null,{\"a\":1}]}");
    CHECK(en && g_sentence(en, "\"red\"") && g_sentence(en, "3") && g_sentence(en, "null") && g_sentence(en, "{\"a\":1}") && !g_accepts(en, "\"blue\"") && !g_accepts(en, "4"));
    auto alt = conv("{\"anyOf\":[{\"type\":\"integer\"},{\"type\":\"string\",\"minLength\":2,\"maxLength\":3},{\"type\":[\"boolean\",\"null\"]}]}");
    CHECK(alt && g_sentence(alt, "-12") && g_sentence(alt, "\"ab\"") && g_sentence(alt, "\"abc\"") && g_sentence(alt, "true") && g_sentence(alt, "null"));
    CHECK(!g_sentence(alt, "\"a\"") && !g_accepts(alt, "\"abcd") && !g_accepts(alt, "1.5") && !g_accepts(alt, "["));
    auto tup = conv("{\"type\":\"array\",\"prefixItems\":[{\"type\":\"integer\"},{\"type\":\"string\"}]}");
    CHECK(tup && g_sentence(tup, "[1, \"a\"]") && !g_sentence(tup, "[1]") && !g_accepts(tup, "[\"a\""));
    auto arr = conv("{\"type\":\"array\",\"items\":{\"type\":\"integer\"},\"minItems\":2,\"maxItems\":3}");
    CHECK(arr && g_sentence(arr, "[1,2]") && g_sentence(arr, "[1, 2, 3]") && !g_sentence(arr, "[1]") && !g_accepts(arr, "[1,2,3,") && !g_sentence(arr, "[]"));
    // $ref, also recursive; allOf
    auto tree = conv("{\"$ref\":\"#/$defs/node\",\"$defs\":{\"node\":{\"type\":\"object\",\"properties\":{\"v\":{\"type\":\"integer\"},\"kids\":{\"type\":\"array\",\"items\":{\"$ref\":\"#/$defs/node\"}}},"
                     "\"required\":[\"v\"]}}}");
    CHECK(tree && g_sentence(tree, "{\"v\":1,\"kids\":[{\"v\":2},{\"v\":3,\"kids\":[]}]}") && !g_accepts(tree, "{\"v\":1,\"kids\":[{\"kids\""));
    auto all = conv("{\"allOf\":[{\"properties\":{\"a\":{\"type\":\"integer\"}},\"required\":[\"a\"]},{\"properties\":{\"b\":{\"type\":\"string\"}}}]}");
    CHECK(all && g_sentence(all, "{\"a\":1,\"b\":\"x\"}") && g_sentence(all, "{\"a\":1}") && !g_sentence(all, "{\"b\":\"x\"}"));
    // what it refuses
    Json bad; std::string gbnf, err;
    CHECK(Json::parse("{\"$ref\":\"#/$defs/missing\"}", bad) && !json_schema_to_gbnf(bad, gbnf, err) && err.find("$ref") != std::string::npos);
    CHECK(Json::parse("{\"type\":\"frob\"}", bad) && !json_schema_to_gbnf(bad, gbnf, err));
    CHECK(Json::parse("{\"enum\":[]}", bad) && !json_schema_to_gbnf(bad, gbnf, err));
}

static void test_sampler_with_grammar() {
    // a vocabulary of pieces; the logits prefer tokens the grammar refuses
    const std::vector<std::string> pieces = {"", "yes", "no", "y", "es", " ", "!", "maybe", "\xC3", "\xA9", "n"};
    std::vector<uint8_t> eog(pieces.size(), 0);
    eog[0] = 1;                                                    // token 0 ends the generation
    std::string err;
    auto g = Grammar::parse("root ::= (\"yes\" | \"no\" | \"n\\u00e9\") \"!\"\n", err);
    CHECK(g);
    SamplingParams p;
    p.temp = 0.0f; p.penalty_repeat = 1.0f;
    Sampler s(p);
    s.set_grammar(g, &pieces, &eog);
    CHECK(!s.is_plain_greedy() && s.has_grammar());
    std::vector<float> lg = {9.0f, 1.0f, 0.5f, 2.0f, 3.0f, 5.0f, 6.0f, 8.0f, 0.1f, 0.2f, 0.3f};
    CHECK(!s.grammar_admits(0) && !s.grammar_admits(7) && s.grammar_admits(1) && s.grammar_admits(3) && !s.grammar_admits(4) && s.grammar_admits(10));
    int t = s.sample(lg.data(), (int)lg.size());
    CHECK(t == 3);                                                 // "y": the best of {yes, no, y, n}; "maybe", the end token, "!" and " " are refused
    s.accept(t);
    t = s.sample(lg.data(), (int)lg.size());
    CHECK(t == 4);                                                 // only "es" continues "y"
    s.accept(t);
    CHECK(!s.grammar_admits(0) && s.grammar_admits(6));
    t = s.sample(lg.data(), (int)lg.size());
    CHECK(t == 6);
    s.accept(t);
    CHECK(s.grammar_admits(0) && !s.grammar_admits(6) && !s.grammar_admits(1));
    t = s.sample(lg.data(), (int)lg.size());
    CHECK(t == 0);                                                 // the sentence is complete: nothing but the end token
    // a character split over two tokens; tokens accepted as PROMPT do not move the grammar; reset() starts the sentence again
    s.reset();
    s.accept(7, false);
    CHECK(s.grammar_admits(10));
    s.accept(10);
    CHECK(s.grammar_admits(8) && !s.grammar_admits(9) && !s.grammar_admits(6) && !s.grammar_admits(0));
    s.accept(8);
    CHECK(s.grammar_admits(9) && !s.grammar_admits(8) && !s.grammar_admits(6));
    s.accept(9);
    CHECK(s.grammar_admits(6) && !s.grammar_admits(0));
    s.reset();
    CHECK(s.grammar_admits(1) && !s.grammar_admits(6));
    // the device front end hands over k candidates: a refused draw is re-drawn from the whole row
    std::vector<TokenProb> c = {{0, 9.0f}, {7, 8.0f}, {6, 6.0f}};
    t = s.finish(c);
    CHECK(t == 0 && !s.grammar_admits(t) && s.resample_with_grammar(lg.data(), (int)lg.size()) == 3);
    // a vocabulary that cannot continue the sentence ends the generation
    const std::vector<std::string> poor = {"", "x", "z"};
    std::vector<uint8_t> poor_eog = {1, 0, 0};
    Sampler s2(p);
    s2.set_grammar(g, &poor, &poor_eog);
    std::vector<float> lg2 = {0.0f, 1.0f, 2.0f};
    CHECK(s2.sample(lg2.data(), 3) == 0);
}

// the questions ProcessToken asks about the unsent text (host/server_context.cc), through the reference's entry name
static void test_stop_string_scan() {
    FakeBackend be;
    ServerParams sp;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    LlamaClientSlot slot;
    slot.params.antiprompt = {"User:", "</s>", "Use"};
    const size_t npos = std::string::npos;
    // a complete stop string: the earliest one wins, the first of the list on a tie; the search window is the stop string plus the newest piece
    CHECK(ctx.FindStoppingStrings("Hello User: hi", 8, true, slot) == 6 && slot.stopped_word && slot.stopping_word == "User:" && !slot.has_next_token);
    slot.stopped_word = false; slot.has_next_token = true; slot.stopping_word.clear();
    CHECK(ctx.FindStoppingStrings("Hello User: and a long tail", 2, true, slot) == npos && !slot.stopped_word && slot.has_next_token);   // outside the window
    CHECK(ctx.FindStoppingStrings("xx</s>yyUse", 3, true, slot) == 8 && slot.stopping_word == "Use");                                  // "</s>" lies outside, "Use" inside
    slot.stopped_word = false; slot.has_next_token = true;
    CHECK(ctx.FindStoppingStrings("", 0, true, slot) == npos);
    // text that may still become a stop string: the longest beginning of a word the text ends with, the earliest over the words
    CHECK(ctx.FindStoppingStrings("Hello Us", 2, false, slot) == 6);
    CHECK(ctx.FindStoppingStrings("Hello <", 1, false, slot) == 6);
    CHECK(ctx.FindStoppingStrings("Hello </", 1, false, slot) == 6);
    CHECK(ctx.FindStoppingStrings("Hello", 1, false, slot) == npos);
    CHECK(ctx.FindStoppingStrings("U", 1, false, slot) == 0);
    CHECK(ctx.FindStoppingStrings("", 0, false, slot) == npos);
    CHECK(!slot.stopped_word && slot.has_next_token);          // the open-stop question changes nothing
    slot.params.antiprompt.clear();
    CHECK(ctx.FindStoppingStrings("anything", 3, true, slot) == npos && ctx.FindStoppingStrings("anything", 3, false, slot) == npos);
}

static void test_slot_loop() {
    FakeBackend be;
    ServerParams sp;
    sp.n_parallel = 2;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    Json d = Json::object();
    d["prompt"] = "hello world"; d["n_predict"] = 8; d["temperature"] = 0.0; d["stream"] = false;
    const int id = ctx.RequestCompletion(d, false, false, -1);
    TaskResult r = ctx.NextResult(id);
    CHECK(!r.error && r.stop);
    const auto prompt = be.voc.tokenize("hello world", true);
    CHECK(r.result_json["tokens_evaluated"].as_int() == (int64_t)prompt.size());
    CHECK(r.result_json["tokens_predicted"].as_int() == 8);
    CHECK(r.result_json["content"].as_string() == expected_text(be, prompt, 8));
    CHECK(r.result_json["stopped_limit"].as_bool());
    for (const char *k : {"prompt_n", "prompt_ms", "prompt_per_second", "predicted_n", "predicted_ms", "predicted_per_second"}) CHECK(r.result_json["timings"].contains(k));
    ctx.RequestCancel(id);

    // streaming: partial contents concatenate to the full text
    d["stream"] = true; d["n_predict"] = 6;
    const int id2 = ctx.RequestCompletion(d, false, false, -1);
    std::string acc;
    while (true) {
        TaskResult pr = ctx.NextResult(id2);
        CHECK(!pr.error);
        acc += pr.result_json["content"].as_string();
        if (pr.stop) break;
    }
    CHECK(acc == expected_text(be, prompt, 6));

    // stop word: find a piece that the deterministic model emits at step 3 and use it as the stop string
    {
        int tok = prompt.back(), pos = (int)prompt.size() - 1;
        std::string before;
        std::string stopw;
        for (int i = 0; i < 4; i++) { tok = be.next_token(tok, pos); pos++; if (i < 3) before += be.voc.token_to_piece(tok); else stopw = be.voc.token_to_piece(tok); }
        if (!stopw.empty() && before.find(stopw) == std::string::npos) {
            Json ds = d;
            ds["stream"] = false; ds["n_predict"] = 20;
            Json st = Json::array(); st.push_back(stopw);
            ds["stop"] = st;
            const int id3 = ctx.RequestCompletion(ds, false, false, -1);
            TaskResult sr = ctx.NextResult(id3);
            CHECK(sr.result_json["stopped_word"].as_bool() && sr.result_json["stopping_word"].as_string() == stopw);
            CHECK(sr.result_json["content"].as_string() == before);
        }
    }

    // EOS ends generation
    be.eos_after = (int)prompt.size() + 2;
    Json de = d; de["stream"] = false; de["n_predict"] = 50;
    const int id4 = ctx.RequestCompletion(de, false, false, -1);
    TaskResult er = ctx.NextResult(id4);
    CHECK(er.result_json["stopped_eos"].as_bool() && er.result_json["tokens_predicted"].as_int() <= 5);
    be.eos_after = -1;

    // two requests at once -> both slots active, decode batches mix sequence ids (continuous batching)
    be.calls_seq.clear();
    be.decode_sleep_us = 500;
    Json a = d, b = d;
    a["stream"] = false; b["stream"] = false; a["n_predict"] = 60; b["n_predict"] = 60; b["prompt"] = "world hello";
    const int ia = ctx.RequestCompletion(a, false, false, -1), ib = ctx.RequestCompletion(b, false, false, -1);
    TaskResult ra = ctx.NextResult(ia), rb = ctx.NextResult(ib);
    CHECK(ra.stop && rb.stop && !ra.error && !rb.error);
    bool mixed = false;
    for (const auto &c : be.calls_seq) { std::set<int> s(c.begin(), c.end()); if (s.size() > 1) mixed = true; }
    CHECK(mixed);
    CHECK(ra.result_json["content"].as_string() == expected_text(be, prompt, 60));
    ctx.ReleaseResources();
}

static std::string b64(const std::string &raw) {
    static const char tbl[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    for (size_t i = 0; i < raw.size(); i += 3) {
        const unsigned b0 = (unsigned char)raw[i], b1 = i + 1 < raw.size() ? (unsigned char)raw[i + 1] : 0, b2 = i + 2 < raw.size() ? (unsigned char)raw[i + 2] : 0, w = b0 << 16 | b1 << 8 | b2;
        out += tbl[w >> 18 & 63]; out += tbl[w >> 12 & 63]; out += i + 1 < raw.size() ? tbl[w >> 6 & 63] : '='; out += i + 2 < raw.size() ? tbl[w & 63] : '=';
    }
    return out;
}
// image_data + [img-N] placeholders (llama_server_context.cc:557-623, 814-831, 1073-1129): the prompt is cut at the placeholders; text, an image's rows, text ...
// reach the model in that order at consecutive positions, the rows as embedding batches of at most n_batch
static void test_image_requests() {
    FakeBackend be;
    be.mm = true; be.nbatch = 16;
    ServerParams sp;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    auto image = [&](int id, int rows) { Json j = Json::object(); j["id"] = id; j["data"] = b64(std::string("IMG") + (char)rows); return j; };
    {
        Json d = Json::object();
        d["prompt"] = "see [img-7] and [img-3] now"; d["n_predict"] = 4; d["temperature"] = 0.0; d["stream"] = false; d["cache_prompt"] = true;
        Json imgs = Json::array(); imgs.push_back(image(3, 5)); imgs.push_back(image(7, 40));      // (listed in another order than the prompt names them)
        d["image_data"] = imgs;
        be.order.clear(); be.embd_calls.clear(); be.calls_pos.clear();
        const int id = ctx.RequestCompletion(d, false, false, -1);
        TaskResult r = ctx.NextResult(id);
        CHECK(!r.error && r.stop);
        const auto t0 = be.voc.tokenize("see ", true), t1 = be.voc.tokenize(" and ", false), t2 = be.voc.tokenize(" now", false);
        const int n0 = (int)t0.size(), n1 = (int)t1.size(), n2 = (int)t2.size();
        CHECK(r.result_json["tokens_evaluated"].as_int() == n0 + 40 + n1 + 5 + n2);
        // order and positions: text, 40 rows in batches of 16 / 16 / 8, text, 5 rows, text, then the generated tokens one by one
        CHECK(be.order.size() >= 7);
        CHECK(be.order[0] == "t" + std::to_string(n0) && be.order[1] == "e16" && be.order[2] == "e16" && be.order[3] == "e8");
        CHECK(be.order[4] == "t" + std::to_string(n1) && be.order[5] == "e5" && be.order[6] == "t" + std::to_string(n2));
        CHECK(be.embd_calls.size() == 4 && be.embd_calls[0].pos0 == n0 && be.embd_calls[1].pos0 == n0 + 16 && be.embd_calls[2].pos0 == n0 + 32);
        CHECK(be.embd_calls[3].pos0 == n0 + 40 + n1 && be.embd_calls[3].n == 5);
        CHECK(be.embd_calls[0].first == 40.0f && be.embd_calls[3].first == 5.0f);                // (each image's own rows)
        CHECK(be.calls_pos[1][0] == n0 + 40 && be.calls_pos[2][0] == n0 + 40 + n1 + 5);
        CHECK(be.calls_pos[3][0] == n0 + 40 + n1 + 5 + n2);                                      // the first generated token follows the prompt
        ctx.RequestCancel(id);
    }
    auto expect_error = [&](Json d, const char *what) {
        const int id = ctx.RequestCompletion(d, false, false, -1);
        TaskResult r = ctx.NextResult(id);
        CHECK(r.error);
        if (!r.error || r.result_json.dump().find(what) == std::string::npos) fprintf(stderr, "  expected an error naming '%s', got %s\n", what, r.result_json.dump().c_str());
        CHECK(r.result_json.dump().find(what) != std::string::npos);
        ctx.RequestCancel(id);
    };
    Json base = Json::object();
    base["n_predict"] = 2; base["temperature"] = 0.0; base["stream"] = false;
    {   // a placeholder without an image; bytes that are no image; no base64; a prompt that ends in an image; no placeholder at all; an encoder failure
        Json d = base; d["prompt"] = "a [img-9] b"; Json im = Json::array(); im.push_back(image(1, 3)); d["image_data"] = im;
        expect_error(d, "not found");
        d = base; d["prompt"] = "a [img-1] b"; im = Json::array(); { Json j = Json::object(); j["id"] = 1; j["data"] = b64("hello"); im.push_back(j); } d["image_data"] = im;
        expect_error(d, "failed to load image");
        d = base; d["prompt"] = "a [img-1] b"; im = Json::array(); { Json j = Json::object(); j["id"] = 1; j["data"] = "***"; im.push_back(j); } d["image_data"] = im;
        expect_error(d, "base64");
        d = base; d["prompt"] = "a [img-1]"; im = Json::array(); im.push_back(image(1, 3)); d["image_data"] = im;
        expect_error(d, "end in text");
        d = base; d["prompt"] = "no picture named"; im = Json::array(); im.push_back(image(1, 3)); d["image_data"] = im;
        expect_error(d, "names none");
        be.embed_fail_for = 6;
        d = base; d["prompt"] = "a [img-1] b"; im = Json::array(); im.push_back(image(1, 6)); d["image_data"] = im;
        expect_error(d, "Failed processing images");
        be.embed_fail_for = -1;
        d = base; d["prompt"] = "a [img-1] b"; im = Json::array(); im.push_back(image(1, 250)); d["image_data"] = im;       // 250 rows + text >= n_ctx 256
        expect_error(d, "do not fit");
    }
    {   // and the loop still serves: a text request, then an image request
        Json d = base; d["prompt"] = "hello world";
        const int id = ctx.RequestCompletion(d, false, false, -1);
        TaskResult r = ctx.NextResult(id);
        CHECK(!r.error && r.stop);
        ctx.RequestCancel(id);
        d = base; d["prompt"] = "x [img-0] y"; Json im = Json::array(); im.push_back(image(0, 2)); d["image_data"] = im;
        const int id2 = ctx.RequestCompletion(d, false, false, -1);
        r = ctx.NextResult(id2);
        CHECK(!r.error && r.stop);
        ctx.RequestCancel(id2);
    }
    // a context without a projector ignores image_data (the engine refuses such requests before they get here)
    FakeBackend plain;
    LlamaServerContext c2(&plain, sp);
    c2.Initialize();
    Json d = base; d["prompt"] = "a [img-1] b"; Json im = Json::array(); im.push_back(image(1, 3)); d["image_data"] = im;
    const int id = c2.RequestCompletion(d, false, false, -1);
    TaskResult r = c2.NextResult(id);
    CHECK(!r.error && plain.embd_calls.empty());
    c2.ReleaseResources();
    ctx.ReleaseResources();
}

static void test_prompt_cache_and_shift() {
    FakeBackend be;
    be.ctx = 32;
    ServerParams sp;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    Json d = Json::object();
    d["prompt"] = "hello world hello"; d["n_predict"] = 3; d["temperature"] = 0.0; d["cache_prompt"] = true;
    TaskResult r1 = ctx.NextResult(ctx.RequestCompletion(d, false, false, -1));
    CHECK(r1.stop);
    const size_t ncalls = be.calls_tokens.size();
    be.clear_ops();
    d["prompt"] = "hello world hello world";          // shares the first 4 tokens (BOS + 3 words) with the cached sequence
    TaskResult r2 = ctx.NextResult(ctx.RequestCompletion(d, false, false, -1));
    CHECK(r2.stop);
    const auto p2 = be.voc.tokenize("hello world hello world", true);
    CHECK(be.calls_tokens.size() > ncalls);
    // only the non-cached suffix was decoded as prompt: first new call holds 1 prompt token (the 5th), not 5
    CHECK(be.calls_tokens[ncalls].size() == 1 && be.calls_tokens[ncalls][0] == p2.back() && be.calls_pos[ncalls][0] == 4);
    bool rm4 = false;
    for (const auto &op : be.ops_snapshot()) if (op == "rm 0 4 -1") rm4 = true;
    CHECK(rm4);
    CHECK(r2.result_json["tokens_evaluated"].as_int() == (int64_t)p2.size());

    // context shift: generate past the slot context
    be.clear_ops();
    Json g = Json::object();
    g["prompt"] = "hello"; g["n_predict"] = 45; g["temperature"] = 0.0; g["n_keep"] = 1;
    TaskResult r3 = ctx.NextResult(ctx.RequestCompletion(g, false, false, -1));
    CHECK(r3.stop && r3.result_json["truncated"].as_bool());
    ctx.ReleaseResources();          // joins the loop thread: the fake backend's op log is only read once it is quiescent
    bool saw_rm = false, saw_add = false;
    for (const auto &op : be.kv_ops) {
        if (op.rfind("rm 0 2 ", 0) == 0) saw_rm = true;          // seq_rm(slot, n_keep + 1, n_keep + n_discard + 1)
        if (op.rfind("add 0 ", 0) == 0 && op.find(" -") != std::string::npos) saw_add = true;
    }
    CHECK(saw_rm && saw_add);
    CHECK(r3.result_json["tokens_predicted"].as_int() == 45);
}

static void test_kv_full_error() {
    FakeBackend be;
    be.fail_decode_over = 0;          // every decode reports "no KV slot"
    ServerParams sp;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    Json d = Json::object();
    d["prompt"] = "hello world"; d["n_predict"] = 4;
    TaskResult r = ctx.NextResult(ctx.RequestCompletion(d, false, false, -1));
    CHECK(r.error && r.result_json["content"].as_string().find("too big") != std::string::npos);
    ctx.ReleaseResources();
}

// a request carrying a token id outside the vocabulary fails alone; a backend failure is reported as such
static void test_bad_token_ids_and_backend_errors() {
    FakeBackend be;
    ServerParams sp;
    sp.n_parallel = 2;
    LlamaServerContext ctx(&be, sp);
    ctx.Initialize();
    Json good = Json::object();
    good["prompt"] = "hello world"; good["n_predict"] = 3;
    Json bad = Json::object();
    Json toks = Json::array();
    toks.push_back(Json((int64_t)5)); toks.push_back(Json((int64_t)(be.n_vocab() + 7)));
    bad["prompt_tokens"] = toks; bad["n_predict"] = 3;
    const int id_bad = ctx.RequestCompletion(bad, false, false, -1);
    const int id_good = ctx.RequestCompletion(good, false, false, -1);
    TaskResult rb = ctx.NextResult(id_bad);
    CHECK(rb.error && rb.result_json["content"].as_string().find("token id") != std::string::npos);
    TaskResult rg = ctx.NextResult(id_good);
    CHECK(!rg.error);                                          // the neighbour was not dragged down
    Json bad2 = Json::object();
    Json mixed = Json::array();
    mixed.push_back(Json("hi")); mixed.push_back(Json((int64_t)-4));
    bad2["prompt"] = mixed; bad2["n_predict"] = 2;
    TaskResult rb2 = ctx.NextResult(ctx.RequestCompletion(bad2, false, false, -1));
    CHECK(rb2.error);
    be.hard_fail = 1;
    TaskResult rf = ctx.NextResult(ctx.RequestCompletion(good, false, false, -1));
    CHECK(rf.error && rf.result_json["content"].as_string().find("device lost") != std::string::npos);
    CHECK(rf.result_json["content"].as_string().find("too big") == std::string::npos);
    be.hard_fail = 0;
    TaskResult ok = ctx.NextResult(ctx.RequestCompletion(good, false, false, -1));
    CHECK(!ok.error);                                          // the slots came back
    ctx.ReleaseResources();
}

// a GGUF whose counts, sizes or offsets lie about the file must be refused without sizing anything from them
static void test_gguf_hardening() {
    auto put = [](std::string &b, const void *p, size_t n) { b.append(reinterpret_cast<const char *>(p), n); };
    auto u32 = [&](std::string &b, uint32_t v) { put(b, &v, 4); };
    auto u64 = [&](std::string &b, uint64_t v) { put(b, &v, 8); };
    auto str = [&](std::string &b, const std::string &t) { u64(b, t.size()); b += t; };
    auto header = [&](uint64_t n_tensors, uint64_t n_kv) { std::string b; u32(b, 0x46554747u); u32(b, 3); u64(b, n_tensors); u64(b, n_kv); return b; };
    auto open_err = [&](const std::string &bytes) {
        char path[] = "/tmp/mi355_gguf_XXXXXX";
        const int fd = mkstemp(path);
        CHECK(fd >= 0);
        CHECK(write(fd, bytes.data(), bytes.size()) == (ssize_t)bytes.size());
        close(fd);
        GGUFFile f;
        const std::string err = f.open(path);
        unlink(path);
        return err;
    };
    auto tensor = [&](std::string &b, const std::string &name, std::initializer_list<uint64_t> ne, uint32_t type, uint64_t off) {
        str(b, name); u32(b, (uint32_t)ne.size()); for (uint64_t d : ne) u64(b, d); u32(b, type); u64(b, off);
    };
    {   // a well-formed one-tensor file loads
        std::string b = header(1, 1);
        str(b, "general.architecture"); u32(b, 8); str(b, "llama");
        tensor(b, "t", {32, 2}, 0, 0);
        while (b.size() % 32) b.push_back('\0');
        b.append(32 * 2 * 4, '\1');
        CHECK(open_err(b).empty());
    }
    CHECK(!open_err(header(1ull << 60, 0)).empty());             // tensor count far beyond the file: no resize()
    CHECK(!open_err(header(0, 1ull << 60)).empty());             // key count likewise
    {   // string array whose count exceeds the remaining bytes: no reserve()
        std::string b = header(0, 1);
        str(b, "tokenizer.ggml.tokens"); u32(b, 9); u32(b, 8); u64(b, 1ull << 58);
        CHECK(!open_err(b).empty());
    }
    {   // scalar array whose count * size wraps 64 bits
        std::string b = header(0, 1);
        str(b, "x"); u32(b, 9); u32(b, 10); u64(b, (1ull << 61) + 1);
        CHECK(!open_err(b).empty());
    }
    {   // a "negative" dimension (2^63 + ..) and a zero dimension
        std::string b = header(1, 0);
        tensor(b, "t", {0x8000000000000010ull, 2}, 0, 0);
        CHECK(!open_err(b).empty());
        std::string c = header(1, 0);
        tensor(c, "t", {32, 0}, 0, 0);
        CHECK(!open_err(c).empty());
    }
    {   // rows * row bytes overflows size_t; offset + bytes wraps
        std::string b = header(1, 0);
        tensor(b, "t", {1ull << 40, 1ull << 40, 1ull << 40}, 0, 0);
        while (b.size() % 32) b.push_back('\0');
        CHECK(!open_err(b).empty());
        std::string c = header(1, 0);
        tensor(c, "t", {32, 2}, 0, 0xffffffffffffffe0ull);
        while (c.size() % 32) c.push_back('\0');
        c.append(256, '\1');
        CHECK(!open_err(c).empty());
    }
    {   // tensor data past the end of the file
        std::string b = header(1, 0);
        tensor(b, "t", {32, 64}, 0, 0);
        while (b.size() % 32) b.push_back('\0');
        b.append(100, '\1');
        CHECK(!open_err(b).empty());
    }
}

static void test_engine() {
    LlamaEngine eng([](const Json &body, BackendInfo &info, std::string &err) -> std::unique_ptr<IBackend> {
        if (body["llama_model_path"].as_string() == "/bad") { err = "no such file"; return nullptr; }
        info.vram = 123; info.model_size = 456;
        return std::unique_ptr<IBackend>(new FakeBackend());
    });
    CHECK(eng.IsSupported("HandleChatCompletion") && !eng.IsSupported("Nope"));
    Json load = Json::object();
    load["llama_model_path"] = "/models/tiny-test.gguf"; load["ctx_len"] = 256; load["n_parallel"] = 2;
    CHECK(LlamaEngine::GetModelId(load) == "tiny-test");
    int code = 0; std::string msg;
    auto grab = [&](Json &&st, Json &&body) { code = (int)st["status_code"].as_int(); msg = body["message"].as_string(); };
    eng.LoadModel(load, grab);
    CHECK(code == 200 && msg == "Model loaded successfully");
    eng.LoadModel(load, grab);
    CHECK(code == 409);
    Json bad = Json::object(); bad["llama_model_path"] = "/bad";
    eng.LoadModel(bad, grab);
    CHECK(code == 500);
    Json none = Json::object();
    eng.LoadModel(none, grab);
    CHECK(code == 400);
    // `mmproj` (reference: src/llama_engine.cc:553-562 makes the context multimodal): the path goes to the backend factory, which loads the projector or fails the
    // load; anything but a path is refused, in the reference's load-error shape (:376-384)
    {
        Json mm = Json::object(); mm["llama_model_path"] = "/models/llava.gguf"; mm["mmproj"] = 17;
        Json st_mm, body_mm;
        eng.LoadModel(mm, [&](Json &&st, Json &&b) { st_mm = st; body_mm = b; });
        CHECK(st_mm["status_code"].as_int() == 500 && st_mm["has_error"].as_bool() && body_mm["message"].as_string() == "Failed to load model");
        CHECK(body_mm["error"].as_string().find("mmproj") != std::string::npos);
    }
    Json models;
    eng.GetModels(Json::object(), [&](Json &&, Json &&b) { models = b; });
    CHECK(models["data"].size() == 1 && models["data"].at(0)["id"].as_string() == "tiny-test" && models["data"].at(0)["vram"].as_int() == 123);

    // non-stream chat completion
    Json req = Json::object();
    req["model"] = "tiny-test"; req["max_tokens"] = 6; req["temperature"] = 0.0;
    Json msgs = Json::array(), m1 = Json::object();
    m1["role"] = "user"; m1["content"] = "hello world";
    msgs.push_back(m1);
    req["messages"] = msgs;
    std::mutex mu; std::condition_variable cv; bool done = false; Json body, status;
    eng.HandleChatCompletion(req, [&](Json &&st, Json &&b) { std::lock_guard<std::mutex> lk(mu); status = st; body = b; done = true; cv.notify_all(); });
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
    CHECK(status["status_code"].as_int() == 200 && !status["has_error"].as_bool() && status["is_done"].as_bool());
    CHECK(body["object"].as_string() == "chat.completion" && body["choices"].at(0)["message"]["role"].as_string() == "assistant");
    CHECK(body["usage"]["completion_tokens"].as_int() == 6 && body["usage"]["total_tokens"].as_int() == body["usage"]["prompt_tokens"].as_int() + 6);

    // an `image_url` content piece (reference: src/llama_engine.cc:854-900, multimodal contexts only): 400, synchronously, nothing generated
    {
        Json rq2 = req;
        Json pieces = Json::array(), pt = Json::object(), pi = Json::object(), url = Json::object();
        pt["type"] = "text"; pt["text"] = "what is this?";
        url["url"] = "data:image/png;base64,AAAA"; pi["type"] = "image_url"; pi["image_url"] = url;
        pieces.push_back(pt); pieces.push_back(pi);
        Json mi = Json::object(); mi["role"] = "user"; mi["content"] = pieces;
        Json ms2 = Json::array(); ms2.push_back(mi);
        rq2["messages"] = ms2;
        Json st_i, body_i; int calls = 0;
        eng.HandleChatCompletion(rq2, [&](Json &&st, Json &&b) { st_i = st; body_i = b; calls++; });
        CHECK(calls == 1 && st_i["status_code"].as_int() == 400 && st_i["has_error"].as_bool() && body_i["message"].as_string().find("image_url") != std::string::npos);
        // text-only content arrays still work (the first text piece is the content, :816-852)
        Json only = Json::array(); only.push_back(pt);
        mi["content"] = only; ms2 = Json::array(); ms2.push_back(mi); rq2["messages"] = ms2;
        done = false;
        eng.HandleChatCompletion(rq2, [&](Json &&st, Json &&b) { std::lock_guard<std::mutex> lk(mu); status = st; body = b; done = true; cv.notify_all(); });
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
        CHECK(status["status_code"].as_int() == 200 && body["object"].as_string() == "chat.completion");
    }

    // streaming
    req["stream"] = true;
    Json so = Json::object(); so["include_usage"] = true; req["stream_options"] = so;
    std::vector<std::string> chunks; done = false;
    eng.HandleChatCompletion(req, [&](Json &&st, Json &&b) {
        std::lock_guard<std::mutex> lk(mu);
        chunks.push_back(b["data"].as_string());
        if (st["is_done"].as_bool()) { done = true; cv.notify_all(); }
    });
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
    CHECK(chunks.size() >= 2);
    for (const auto &c : chunks) CHECK(c.rfind("data: ", 0) == 0);
    CHECK(chunks.back().find("data: [DONE]\n\n") != std::string::npos);
    CHECK(chunks.back().find("\"total_tokens\"") != std::string::npos);
    Json first;
    CHECK(Json::parse(chunks[0].substr(6, chunks[0].size() - 8), first) && first["object"].as_string() == "chat.completion.chunk");

    Json un = Json::object(); un["model"] = "tiny-test";
    eng.UnloadModel(un, grab);
    CHECK(code == 200);
    eng.HandleChatCompletion(req, grab);
    CHECK(code == 409);
}

// A model unloaded under a running stream, and under a stream still queued behind it (reference: src/llama_engine.cc:1028-1041 - the loop over
// model_loaded_external ends and the provider gets {data: "", has_error: true, is_stream: true} as its last callback); a stream stopped by StopInferencing
// ends without a terminal callback (:950-955) and its callback object is destroyed with the task.
static void test_engine_unload_mid_stream() {
    LlamaEngine eng([](const Json &, BackendInfo &, std::string &) -> std::unique_ptr<IBackend> {
        auto *fb = new FakeBackend(); fb->decode_sleep_us = 2000; return std::unique_ptr<IBackend>(fb);
    });
    Json load = Json::object();
    load["llama_model_path"] = "/models/slow.gguf"; load["ctx_len"] = 256; load["n_parallel"] = 1;
    int code = 0;
    auto grab = [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); };
    Json req = Json::object();
    req["model"] = "slow"; req["max_tokens"] = 200; req["temperature"] = 0.0; req["stream"] = true;
    Json msgs = Json::array(), m1 = Json::object();
    m1["role"] = "user"; m1["content"] = "hello world";
    msgs.push_back(m1);
    req["messages"] = msgs;
    struct Seen { std::mutex mu; std::condition_variable cv; int chunks = 0, terminal = 0, after_terminal = 0; Json last_status, last_body; };
    auto watch = [](std::shared_ptr<Seen> s, std::shared_ptr<int> token) {
        return [s, token](Json &&st, Json &&b) {
            std::lock_guard<std::mutex> lk(s->mu);
            if (s->terminal) s->after_terminal++;
            if (st["is_done"].as_bool() || st["has_error"].as_bool()) { s->terminal++; s->last_status = st; s->last_body = b; }
            else s->chunks++;
            s->cv.notify_all();
        };
    };
    for (int round = 0; round < 2; round++) {       // 0: unload under the streams; 1: StopInferencing
        eng.LoadModel(load, grab);
        CHECK(code == 200);
        auto a = std::make_shared<Seen>(), b = std::make_shared<Seen>();
        auto ta = std::make_shared<int>(0), tb = std::make_shared<int>(0);
        std::weak_ptr<int> wa = ta, wb = tb;
        eng.HandleChatCompletion(req, watch(a, std::move(ta)));
        if (round == 0) eng.HandleChatCompletion(req, watch(b, std::move(tb)));   // one worker (n_parallel 1): waits in the queue behind the first
        else tb.reset();
        {   // (polled: gcc 11's ThreadSanitizer does not see the unlock inside condition_variable::wait_for's pthread_cond_clockwait and reports the next lock)
            bool got = false;
            for (int i = 0; i < 4000 && !got; i++) {
                { std::lock_guard<std::mutex> lk(a->mu); got = a->chunks >= 2; }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(5));
            }
            CHECK(got);
        }
        Json un = Json::object(); un["model"] = "slow";
        if (round == 0) {
            eng.UnloadModel(un, grab);                // joins the workers: both streams have had their last callback when it returns
            CHECK(code == 200);
            for (auto &s : {a, b}) {
                std::lock_guard<std::mutex> lk(s->mu);
                CHECK(s->terminal == 1 && s->after_terminal == 0);
                CHECK(s->last_status["has_error"].as_bool() && s->last_status["is_stream"].as_bool() && !s->last_status["is_done"].as_bool());
                CHECK(s->last_status["status_code"].as_int() == 200 && s->last_body["data"].is_string() && s->last_body["data"].as_string().empty());
            }
            { std::lock_guard<std::mutex> lk(b->mu); CHECK(b->chunks == 0); }
            CHECK(a->chunks < 200);
            CHECK(wa.expired() && wb.expired());      // nothing holds the callbacks any more
        } else {
            eng.StopInferencing("slow");
            for (int i = 0; i < 2000 && !wa.expired(); i++) std::this_thread::sleep_for(std::chrono::milliseconds(5));
            CHECK(wa.expired());                      // the task ended and let go of its callback ...
            { std::lock_guard<std::mutex> lk(a->mu); CHECK(a->terminal == 0 && a->chunks < 200); }   // ... without a terminal chunk, as the reference
            eng.UnloadModel(un, grab);
            CHECK(code == 200);
        }
    }
}

// `host_tests --tokenize model.gguf cases.json` : tokenizes every string of the JSON array with the GGUF's tokenizer and
// prints {"ids": [[...], ...], "pieces": [[...], ...], "bos":, "eos":} — compared against HF `tokenizers` by test_host_logic.py
static int tokenize_cli(const char *gguf, const char *cases) {
    GGUFFile f;
    std::string err;
    err = f.open(gguf);
    if (!err.empty()) { fprintf(stderr, "gguf: %s\n", err.c_str()); return 2; }
    Vocab v;
    if (!v.load(f, err)) { fprintf(stderr, "vocab: %s\n", err.c_str()); return 2; }
    FILE *fp = fopen(cases, "rb");
    if (!fp) return 2;
    std::string txt;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) txt.append(buf, n);
    fclose(fp);
    Json arr;
    if (!Json::parse(txt, arr, &err)) { fprintf(stderr, "cases: %s\n", err.c_str()); return 2; }
    Json out = Json::object(), ids = Json::array(), rt = Json::array();
    for (size_t i = 0; i < arr.size(); i++) {
        const bool special = arr.at(i).is_object();
        const std::string text = special ? arr.at(i)["text"].as_string() : arr.at(i).as_string();
        const auto t = v.tokenize(text, false, special);
        Json row = Json::array();
        for (int32_t id : t) row.push_back((int64_t)id);
        ids.push_back(row);
        rt.push_back(v.detokenize(t, true));
    }
    out["ids"] = ids; out["roundtrip"] = rt; out["bos"] = v.bos(); out["eos"] = v.eos(); out["n"] = v.n_tokens();
    printf("%s\n", out.dump().c_str());
    return 0;
}

static std::string strip_eos(std::string t) { if (t.size() >= 4 && t.compare(t.size() - 4, 4, "</s>") == 0) t.resize(t.size() - 4); return t; }

static void test_engine_grammar_requests() {
    LlamaEngine eng([](const Json &, BackendInfo &, std::string &) -> std::unique_ptr<IBackend> { return std::unique_ptr<IBackend>(new FakeBackend()); });
    Json load = Json::object();
    load["llama_model_path"] = "/models/g.gguf"; load["ctx_len"] = 256; load["n_parallel"] = 1;
    int code = 0;
    eng.LoadModel(load, [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); });
    CHECK(code == 200);
    auto complete = [&](Json req, Json &status) {
        std::mutex mu; std::condition_variable cv; bool done = false; Json body;
        Json msgs = Json::array(), m1 = Json::object();
        m1["role"] = "user"; m1["content"] = "hello";
        msgs.push_back(m1);
        req["model"] = "g"; req["messages"] = msgs;
        eng.HandleChatCompletion(req, [&](Json &&st, Json &&b) { std::lock_guard<std::mutex> lk(mu); status = st; body = b; done = true; cv.notify_all(); });
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return body;
    };
    // the backend's logits prefer pseudo-random tokens; the grammar leaves one sentence, spelt with whatever pieces the vocabulary has, then only the end
    for (double temp : {0.0, 0.9}) {
        Json req = Json::object(), st;
        req["max_tokens"] = 40; req["temperature"] = temp; req["seed"] = 5; req["grammar"] = "root ::= \"hello\" \" world\" [!.]\n";
        Json body = complete(req, st);
        CHECK(st["status_code"].as_int() == 200);
        // (the end token's text is part of the content, as in the reference: common_token_to_piece(ctx, tok) with special = true, src/llama_server_context.cc:720)
        const std::string text = strip_eos(body["choices"].at(0)["message"]["content"].as_string());
        CHECK(text == "hello world!" || text == "hello world.");
        CHECK(body["choices"].at(0)["finish_reason"].as_string() == "stop");
    }
    {   // response_format json_object: the text is a prefix of a JSON object (the token budget ends it), and a schema shapes it
        Json req = Json::object(), st, rf = Json::object();
        rf["type"] = "json_object";
        req["max_tokens"] = 24; req["temperature"] = 0.7; req["seed"] = 11; req["response_format"] = rf;
        const std::string text = strip_eos(complete(req, st)["choices"].at(0)["message"]["content"].as_string());
        CHECK(st["status_code"].as_int() == 200 && !text.empty() && text[0] == '{');
        std::string gbnf, err;
        CHECK(json_schema_to_gbnf(Json(), gbnf, err));
        GrammarMatcher m(Grammar::parse(gbnf, err));
        CHECK(m.accept(text));
        Json schema;
        CHECK(Json::parse("{\"type\":\"object\",\"properties\":{\"ok\":{\"type\":\"boolean\"}},\"required\":[\"ok\"]}", schema));
        Json js = Json::object();
        js["schema"] = schema;
        rf["type"] = "json_schema"; rf["json_schema"] = js;
        req["response_format"] = rf; req["max_tokens"] = 60;
        Json body = complete(req, st);
        Json parsed;
        CHECK(st["status_code"].as_int() == 200 && Json::parse(strip_eos(body["choices"].at(0)["message"]["content"].as_string()), parsed) && parsed["ok"].is_bool());
        CHECK(body["choices"].at(0)["finish_reason"].as_string() == "stop");
        Json bad_schema;
        CHECK(Json::parse("{\"type\":\"frob\"}", bad_schema));
        js["schema"] = bad_schema; rf["json_schema"] = js; req["response_format"] = rf;
        Json eb = complete(req, st);
        CHECK(st["status_code"].as_int() == 400 && st["has_error"].as_bool() && eb["message"].as_string().find("response_format") != std::string::npos);
    }
    {   // a grammar that does not parse fails its request with the parser's message; the model keeps serving
        Json req = Json::object(), st;
        req["max_tokens"] = 4; req["grammar"] = "root ::= missing-rule\n";
        Json eb = complete(req, st);
        CHECK(st["has_error"].as_bool() && st["status_code"].as_int() == 400 && eb["message"].as_string().find("undefined rule missing-rule") != std::string::npos);
        Json ok = Json::object();
        ok["max_tokens"] = 3; ok["temperature"] = 0.0;
        complete(ok, st);
        CHECK(st["status_code"].as_int() == 200);
    }
    // grammar_file at load: its text constrains every completion of that model; a missing file fails the load
    char path[] = "/tmp/mi355_grammar_XXXXXX";
    const int fd = mkstemp(path);
    const char *gtext = "root ::= \"st\" \"o\"+ \"p\"\n";
    CHECK(fd >= 0 && write(fd, gtext, strlen(gtext)) == (ssize_t)strlen(gtext));
    close(fd);
    Json load2 = Json::object();
    load2["llama_model_path"] = "/models/h.gguf"; load2["ctx_len"] = 256; load2["grammar_file"] = path; load2["model"] = "h";
    eng.LoadModel(load2, [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); });
    CHECK(code == 200);
    {
        std::mutex mu; std::condition_variable cv; bool done = false; Json body, st;
        Json req = Json::object(), msgs = Json::array(), m1 = Json::object();
        m1["role"] = "user"; m1["content"] = "hello";
        msgs.push_back(m1);
        req["model"] = "h"; req["messages"] = msgs; req["max_tokens"] = 30; req["temperature"] = 1.0; req["seed"] = 3; req["grammar"] = "root ::= \"ignored\"\n";
        eng.HandleChatCompletion(req, [&](Json &&s2, Json &&b) { std::lock_guard<std::mutex> lk(mu); st = s2; body = b; done = true; cv.notify_all(); });
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        const std::string text = strip_eos(body["choices"].at(0)["message"]["content"].as_string());
        CHECK(text.size() >= 4 && text.compare(0, 3, "sto") == 0 && text.find_first_not_of('o', 2) == text.size() - 1 && text.back() == 'p');
    }
    unlink(path);
    Json load3 = Json::object();
    load3["llama_model_path"] = "/models/i.gguf"; load3["grammar_file"] = "/nonexistent/grammar.gbnf";
    eng.LoadModel(load3, [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); });
    CHECK(code == 500);
}

static void test_embeddings() {
    LlamaEngine eng([](const Json &, BackendInfo &, std::string &) -> std::unique_ptr<IBackend> { return std::unique_ptr<IBackend>(new FakeBackend()); });
    Json load = Json::object();
    load["llama_model_path"] = "/models/emb.gguf"; load["n_parallel"] = 2;
    int code = 0;
    eng.LoadModel(load, [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); });
    CHECK(code == 200);
    std::mutex mu; std::condition_variable cv; bool done = false; Json body, status;
    auto wait_cb = [&](Json &&st, Json &&b) { std::lock_guard<std::mutex> lk(mu); status = st; body = b; done = true; cv.notify_all(); };
    auto run = [&](const Json &req) { done = false; eng.HandleEmbedding(req, wait_cb); std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); };
    Json req = Json::object();
    req["model"] = "emb"; req["input"] = "hello world";
    run(req);
    CHECK(status["status_code"].as_int() == 200 && body["object"].as_string() == "list" && body["model"].as_string() == "emb");
    CHECK(body["data"].size() == 1 && body["data"].at(0)["object"].as_string() == "embedding" && body["data"].at(0)["index"].as_int() == 0);
    const Json &e = body["data"].at(0)["embedding"];
    CHECK(e.size() == 8);
    double n2 = 0;
    for (const Json &x : e.items()) n2 += x.as_double() * x.as_double();
    CHECK(std::fabs(n2 - 1.0) < 1e-5);                          // L2-normalised (common_embd_normalize)
    CHECK(body["usage"]["prompt_tokens"].as_int() == 3 && body["usage"]["total_tokens"].as_int() == 3);
    // array of strings + token arrays, base64 encoding
    Json arr = Json::array();
    arr.push_back("hello"); arr.push_back("world hello");
    Json toks = Json::array(); toks.push_back(5); toks.push_back(270); toks.push_back(271);
    arr.push_back(toks);
    req["input"] = arr; req["encoding_format"] = "base64";
    run(req);
    CHECK(body["data"].size() == 3 && body["data"].at(2)["index"].as_int() == 2);
    CHECK(body["data"].at(0)["embedding"].is_string() && body["data"].at(0)["embedding"].as_string().size() == 44);   // 8 floats = 32 B -> 44 chars
    CHECK(body["usage"]["prompt_tokens"].as_int() == 2 + 3 + 3);
    // a chat completion afterwards still samples (the embeddings switch follows the tick)
    Json chat = Json::object();
    chat["model"] = "emb"; chat["max_tokens"] = 4; chat["temperature"] = 0.0;
    Json msgs = Json::array(), m1 = Json::object();
    m1["role"] = "user"; m1["content"] = "hello";
    msgs.push_back(m1); chat["messages"] = msgs;
    done = false;
    eng.HandleChatCompletion(chat, wait_cb);
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
    CHECK(status["status_code"].as_int() == 200 && body["object"].as_string() == "chat.completion");

    // models whose metadata asks for pooling over the sequence (llama_get_embeddings_seq): mean of the tokens' hidden states / the first token's
    for (int pool : {1, 2, 3}) {
        FakeBackend *fb = nullptr;
        LlamaEngine pe([&](const Json &, BackendInfo &, std::string &) -> std::unique_ptr<IBackend> { fb = new FakeBackend(); fb->pooling = pool; return std::unique_ptr<IBackend>(fb); });
        Json pl = Json::object();
        pl["llama_model_path"] = "/models/pool.gguf";
        pe.LoadModel(pl, [&](Json &&st, Json &&) { code = (int)st["status_code"].as_int(); });
        CHECK(code == 200 && fb);
        Json pr = Json::object();
        pr["model"] = "pool"; pr["input"] = "hello world";
        done = false;
        pe.HandleEmbedding(pr, wait_cb);
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
        CHECK(status["status_code"].as_int() == 200 && body["data"].size() == 1);
        const std::vector<int32_t> tk = fb->calls_tokens.back();
        CHECK(tk.size() == 3);
        std::vector<double> want(8, 0.0);
        auto f = [](int tok, int k) { return (double)((tok + 3 * k) % 7) - 3.0; };
        for (int k = 0; k < 8; k++) {
            if (pool == 1) { float acc = 0.0f; for (int32_t t : tk) acc += (float)f(t, k); want[(size_t)k] = (double)(acc * (1.0f / 3.0f)); }
            else want[(size_t)k] = f(pool == 2 ? tk.front() : tk.back(), k);
        }
        double nn = 0;
        for (double v : want) nn += v * v;
        nn = std::sqrt(nn);
        const Json &pe8 = body["data"].at(0)["embedding"];
        bool same = pe8.size() == 8;
        for (int k = 0; same && k < 8; k++) same = std::fabs(pe8.at((size_t)k).as_double() - want[(size_t)k] / nn) < 1e-6;
        CHECK(same);
        // every token of a pooled prompt was flagged; with "last" only the last one
        const auto &fl = fb->last_flag_index;
        int flagged = 0;
        for (int v : fl) flagged += v >= 0;
        CHECK(flagged == (pool == 3 ? 1 : 3));
    }
}

// `host_tests --api-shapes`: runs the engine façade over the fake backend through the request kinds of the reference's surface and prints every status /
// body pair as JSON — tests/test_golden_ops.py compares them with tests/golden/api_shapes_v1.json (transcribed from src/llama_engine.cc)
static int api_shapes_cli() {
    LlamaEngine eng([](const Json &body, BackendInfo &info, std::string &err) -> std::unique_ptr<IBackend> {
        if (body["llama_model_path"].as_string() == "/bad") { err = "no such file"; return nullptr; }
        info.vram = 123; info.model_size = 456;
        return std::unique_ptr<IBackend>(new FakeBackend());
    });
    Json out = Json::object();
    auto rec = [&](const char *name) { return [&out, name](Json &&st, Json &&b) { Json e = Json::object(); e["status"] = st; e["body"] = b; out[name] = e; }; };
    Json load = Json::object();
    load["llama_model_path"] = "/models/tiny-test.gguf"; load["ctx_len"] = 256; load["n_parallel"] = 2;
    eng.LoadModel(Json::object(), rec("load_model_no_id"));
    eng.LoadModel(load, rec("load_model_ok"));
    eng.LoadModel(load, rec("load_model_again"));
    Json bad = Json::object(); bad["llama_model_path"] = "/bad";
    eng.LoadModel(bad, rec("load_model_failed"));
    Json id = Json::object(); id["model"] = "tiny-test";
    eng.GetModelStatus(id, rec("get_model_status_ok"));
    eng.GetModels(Json::object(), rec("get_models"));
    Json req = Json::object();
    req["model"] = "tiny-test"; req["max_tokens"] = 5; req["temperature"] = 0.0;
    Json msgs = Json::array(), m1 = Json::object();
    m1["role"] = "user"; m1["content"] = "hello world";
    msgs.push_back(m1);
    req["messages"] = msgs;
    std::mutex mu; std::condition_variable cv; bool done = false;
    eng.HandleChatCompletion(req, [&](Json &&st, Json &&b) {
        std::lock_guard<std::mutex> lk(mu);
        Json e = Json::object(); e["status"] = st; e["body"] = b; out["chat_completion"] = e; done = true; cv.notify_all();
    });
    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
    for (int usage = 0; usage < 2; usage++) {
        req["stream"] = true;
        Json so = Json::object(); so["include_usage"] = usage != 0; req["stream_options"] = so;
        Json frames = Json::array();
        done = false;
        eng.HandleChatCompletion(req, [&](Json &&st, Json &&b) {
            std::lock_guard<std::mutex> lk(mu);
            Json e = Json::object(); e["status"] = st; e["body"] = b; frames.push_back(e);
            if (st["is_done"].as_bool() || st["has_error"].as_bool()) { done = true; cv.notify_all(); }
        });
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
        out[usage ? "stream_with_usage" : "stream"] = frames;
    }
    eng.UnloadModel(id, rec("unload_model_ok"));
    eng.GetModelStatus(id, rec("model_not_loaded"));
    printf("%s\n", out.dump().c_str());
    return 0;
}

static const char *volatile g_stage = "start";

int main(int argc, char **argv) {
    if (argc == 4 && std::string(argv[1]) == "--tokenize") return tokenize_cli(argv[2], argv[3]);
    if (argc == 2 && std::string(argv[1]) == "--api-shapes") return api_shapes_cli();
    // a run that hangs is ended from outside (timeout's SIGTERM): say which group of checks it was in
    signal(SIGTERM, [](int) { const char *m = g_stage; if (write(2, "hung in: ", 9) < 0 || write(2, m, strlen(m)) < 0 || write(2, "\n", 1) < 0) {} _exit(124); });
#define STAGE(f) do { g_stage = #f; f(); } while (0)
    STAGE(test_json);
    STAGE(test_vocab);
    STAGE(test_vocab_wordpiece);
    STAGE(test_sampler);
    STAGE(test_sampler_topk_matches_full_sort);
    STAGE(test_sampler_mirostat_dynatemp);
    STAGE(test_sampler_front_plan_matches_full_chain);
    STAGE(test_grammar_parse_and_match);
    STAGE(test_json_schema_grammar);
    STAGE(test_sampler_with_grammar);
    STAGE(test_stop_string_scan);
    STAGE(test_slot_loop);
    STAGE(test_image_requests);
    STAGE(test_prompt_cache_and_shift);
    STAGE(test_kv_full_error);
    STAGE(test_bad_token_ids_and_backend_errors);
    STAGE(test_gguf_hardening);
    STAGE(test_engine);
    STAGE(test_engine_unload_mid_stream);
    STAGE(test_engine_grammar_requests);
    STAGE(test_embeddings);
    if (g_fail) { printf("%d check(s) failed\n", g_fail); return 1; }
    printf("all host-logic checks passed\n");
    return 0;
}
