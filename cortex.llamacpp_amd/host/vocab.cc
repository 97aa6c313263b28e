#include "vocab.h"

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstring>
#include <queue>

#include "gguf.h"

namespace mi355 {

namespace {

size_t utf8_len(unsigned char c) {
    if (c < 0x80) return 1;
    if ((c & 0xE0) == 0xC0) return 2;
    if ((c & 0xF0) == 0xE0) return 3;
    if ((c & 0xF8) == 0xF0) return 4;
    return 1;
}

// GPT-2 byte <-> unicode mapping: printable bytes map to themselves, the rest to U+0100.. in order
struct ByteMap {
    std::string to_utf8[256];
    std::unordered_map<std::string, uint8_t> from_utf8;
    ByteMap() {
        int n = 0;
        for (int b = 0; b < 256; b++) {
            const bool keep = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
            const unsigned cp = keep ? (unsigned)b : 256u + (unsigned)n++;
            std::string s;
            if (cp < 0x80) s += (char)cp;
            else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
            to_utf8[b] = s;
            from_utf8[s] = (uint8_t)b;
        }
    }
};
const ByteMap &byte_map() { static const ByteMap m; return m; }

bool is_letter(unsigned cp) { return (cp >= 'a' && cp <= 'z') || (cp >= 'A' && cp <= 'Z') || cp >= 0x80; }
bool is_digit(unsigned cp) { return cp >= '0' && cp <= '9'; }
bool is_space(unsigned cp) { return cp == ' ' || cp == '\t' || cp == '\n' || cp == '\r' || cp == 0x0B || cp == 0x0C; }

}  // namespace

bool Vocab::load(const GGUFFile &f, std::string &err) {
    const GGUFValue *toks = f.find("tokenizer.ggml.tokens");
    if (!toks || toks->type != GV_ARR || toks->elem_type != GV_STR) { err = "no tokenizer.ggml.tokens"; return false; }
    tokens_ = toks->strs;
    model_ = f.get_s("tokenizer.ggml.model", "llama");
    const size_t n = tokens_.size();
    scores_.assign(n, 0.0f);
    types_.assign(n, TT_NORMAL);
    if (const GGUFValue *sc = f.find("tokenizer.ggml.scores"); sc && sc->type == GV_ARR && sc->elem_type == GV_F32 && sc->u == n)
        memcpy(scores_.data(), sc->raw, n * 4);
    if (const GGUFValue *tt = f.find("tokenizer.ggml.token_type"); tt && tt->type == GV_ARR && (tt->elem_type == GV_I32 || tt->elem_type == GV_U32) && tt->u == n)
        for (size_t i = 0; i < n; i++) { int32_t v; memcpy(&v, tt->raw + 4 * i, 4); types_[i] = v; }
    if (const GGUFValue *mg = f.find("tokenizer.ggml.merges"); mg && mg->type == GV_ARR && mg->elem_type == GV_STR) {
        int rank = 0;
        for (const std::string &m : mg->strs) {
            const size_t sp = m.find(' ', 1);
            if (sp == std::string::npos) continue;
            merge_rank_[{m.substr(0, sp), m.substr(sp + 1)}] = rank++;
        }
    }
    bos_ = (int)(int64_t)f.get_u("tokenizer.ggml.bos_token_id", model_ == "llama" ? 1 : (uint64_t)-1);
    eos_ = (int)(int64_t)f.get_u("tokenizer.ggml.eos_token_id", model_ == "llama" ? 2 : (uint64_t)-1);
    eot_ = (int)(int64_t)f.get_u("tokenizer.ggml.eot_token_id", (uint64_t)-1);
    unk_ = (int)(int64_t)f.get_u("tokenizer.ggml.unknown_token_id", 0);
    add_bos_ = f.get_b("tokenizer.ggml.add_bos_token", model_ == "llama" || model_ == "bert");   // (WordPiece vocabularies put [CLS] ... [SEP] around a text by default)
    add_eos_ = f.get_b("tokenizer.ggml.add_eos_token", false);
    add_space_prefix_ = f.get_b("tokenizer.ggml.add_space_prefix", true);
    if (model_ == "bert") {
        // (older conversions carry the classifier token as bos and the separator as eos; llama.cpp's defaults are BERT's 101 / 102 / 100)
        cls_ = (int)(int64_t)f.get_u("tokenizer.ggml.cls_token_id", f.get_u("tokenizer.ggml.bos_token_id", 101));
        sep_ = (int)(int64_t)f.get_u("tokenizer.ggml.seperator_token_id", f.get_u("tokenizer.ggml.eos_token_id", 102));
        unk_ = (int)(int64_t)f.get_u("tokenizer.ggml.unknown_token_id", 100);
        bos_ = cls_; eos_ = sep_;
        if (cls_ < 0 || cls_ >= (int)n || sep_ < 0 || sep_ >= (int)n || unk_ < 0 || unk_ >= (int)n) { err = "bert vocabulary: cls / sep / unk token ids out of range"; return false; }
    }
    build_index();
    return true;
}

void Vocab::init_spm(const std::vector<std::string> &tokens, const std::vector<float> &scores, const std::vector<int> &types,
                     int bos, int eos, int unk, bool add_bos) {
    model_ = "llama";
    tokens_ = tokens; scores_ = scores; types_ = types;
    bos_ = bos; eos_ = eos; unk_ = unk; add_bos_ = add_bos; eot_ = -1;
    build_index();
}

void Vocab::init_wpm(const std::vector<std::string> &tokens, int cls, int sep, int unk) {
    model_ = "bert";
    tokens_ = tokens; scores_.assign(tokens.size(), 0.0f); types_.assign(tokens.size(), TT_NORMAL);
    for (int id : {cls, sep, unk}) if (id >= 0 && id < (int)tokens.size()) types_[(size_t)id] = TT_CONTROL;
    cls_ = bos_ = cls; sep_ = eos_ = sep; unk_ = unk; add_bos_ = true; eot_ = -1;
    build_index();
}

void Vocab::build_index() {
    index_.clear();
    special_ids_.clear();
    max_token_len_ = 0;
    for (int i = 0; i < (int)tokens_.size(); i++) {
        index_.emplace(tokens_[(size_t)i], i);
        max_token_len_ = std::max(max_token_len_, tokens_[(size_t)i].size());
        if (types_[(size_t)i] == TT_CONTROL || types_[(size_t)i] == TT_USER_DEFINED) special_ids_.push_back(i);
        // well-known end-of-generation markers (llama-3 <|eot_id|>, chatml <|im_end|>, ...)
        const std::string &t = tokens_[(size_t)i];
        if (t == "<|eot_id|>" || t == "<|im_end|>" || t == "<|end|>" || t == "<end_of_turn>" || t == "<|endoftext|>" || t == "<EOT>") {
            if (eot_ < 0 && (t == "<|eot_id|>" || t == "<|im_end|>" || t == "<end_of_turn>")) eot_ = i;
            eog_extra_[i] = 1;
        }
    }
    std::sort(special_ids_.begin(), special_ids_.end(), [&](int a, int b) { return tokens_[(size_t)a].size() > tokens_[(size_t)b].size(); });
}

int Vocab::byte_token(uint8_t b) const {
    char buf[8];
    snprintf(buf, sizeof buf, "<0x%02X>", b);
    auto it = index_.find(buf);
    return it == index_.end() ? unk_ : it->second;
}

// SentencePiece-style: start from UTF-8 characters, repeatedly merge the adjacent pair whose concatenation is the
// highest-scoring vocabulary entry (leftmost on ties); what cannot be merged into a known piece falls back to bytes.
void Vocab::tokenize_spm(const std::string &text, std::vector<int32_t> &out) const {
    struct Sym { int prev, next; size_t off, len; };
    std::vector<Sym> syms;
    for (size_t off = 0; off < text.size();) {
        size_t l = std::min(utf8_len((unsigned char)text[off]), text.size() - off);
        syms.push_back({(int)syms.size() - 1, (int)syms.size() + 1, off, l});
        off += l;
    }
    if (syms.empty()) return;
    syms.back().next = -1;
    struct Bigram { int left, right; float score; size_t size; };
    auto cmp = [](const Bigram &a, const Bigram &b) { return a.score < b.score || (a.score == b.score && a.left > b.left); };
    std::priority_queue<Bigram, std::vector<Bigram>, decltype(cmp)> q(cmp);
    auto try_add = [&](int l, int r) {
        if (l < 0 || r < 0) return;
        const std::string piece = text.substr(syms[(size_t)l].off, syms[(size_t)l].len + syms[(size_t)r].len);
        auto it = index_.find(piece);
        if (it == index_.end()) return;
        q.push({l, r, scores_[(size_t)it->second], piece.size()});
    };
    for (int i = 1; i < (int)syms.size(); i++) try_add(i - 1, i);
    while (!q.empty()) {
        const Bigram b = q.top();
        q.pop();
        Sym &L = syms[(size_t)b.left], &R = syms[(size_t)b.right];
        if (L.len == 0 || R.len == 0 || L.len + R.len != b.size) continue;   // stale
        L.len += R.len;
        R.len = 0;
        L.next = R.next;
        if (R.next >= 0) syms[(size_t)R.next].prev = b.left;
        try_add(L.prev, b.left);
        try_add(b.left, L.next);
    }
    for (int i = 0; i >= 0; i = syms[(size_t)i].next) {
        const Sym &s = syms[(size_t)i];
        const std::string piece = text.substr(s.off, s.len);
        auto it = index_.find(piece);
        if (it != index_.end()) out.push_back(it->second);
        else for (unsigned char c : piece) out.push_back(byte_token(c));
    }
}

void Vocab::bpe_word(const std::string &word, std::vector<int32_t> &out) const {
    // word is already in the byte->unicode alphabet; split into its characters
    std::vector<std::string> parts;
    for (size_t off = 0; off < word.size();) {
        const size_t l = std::min(utf8_len((unsigned char)word[off]), word.size() - off);
        parts.push_back(word.substr(off, l));
        off += l;
    }
    while (parts.size() > 1) {
        int best = -1, best_rank = INT_MAX;
        for (size_t i = 0; i + 1 < parts.size(); i++) {
            auto it = merge_rank_.find({parts[i], parts[i + 1]});
            if (it != merge_rank_.end() && it->second < best_rank) { best_rank = it->second; best = (int)i; }
        }
        if (best < 0) break;
        parts[(size_t)best] += parts[(size_t)best + 1];
        parts.erase(parts.begin() + best + 1);
    }
    for (const std::string &p : parts) {
        auto it = index_.find(p);
        if (it != index_.end()) { out.push_back(it->second); continue; }
        for (size_t off = 0; off < p.size();) {    // unknown merge result: emit its single characters
            const size_t l = std::min(utf8_len((unsigned char)p[off]), p.size() - off);
            auto ic = index_.find(p.substr(off, l));
            out.push_back(ic != index_.end() ? ic->second : unk_);
            off += l;
        }
    }
}

// llama-bpe pre-tokenizer, hand-written:
//   (?i:'s|'t|'re|'ve|'m|'ll|'d) | [^\r\n\p{L}\p{N}]?\p{L}+ | \p{N}{1,3} | ?[^\s\p{L}\p{N}]+[\r\n]* | \s*[\r\n]+ | \s+(?!\S) | \s+
void Vocab::tokenize_bpe(const std::string &text, std::vector<int32_t> &out) const {
    std::vector<unsigned> cps;
    std::vector<size_t> offs;
    for (size_t off = 0; off < text.size();) {
        const size_t l = std::min(utf8_len((unsigned char)text[off]), text.size() - off);
        unsigned cp = (unsigned char)text[off];
        if (l == 2) cp = ((cp & 0x1F) << 6) | ((unsigned char)text[off + 1] & 0x3F);
        else if (l == 3) cp = ((cp & 0x0F) << 12) | (((unsigned char)text[off + 1] & 0x3F) << 6) | ((unsigned char)text[off + 2] & 0x3F);
        else if (l == 4) cp = 0x10000;
        cps.push_back(cp);
        offs.push_back(off);
        off += l;
    }
    offs.push_back(text.size());
    const size_t n = cps.size();
    auto emit = [&](size_t a, size_t b) {
        std::string w;
        for (size_t i = offs[a]; i < offs[b]; i++) w += byte_map().to_utf8[(unsigned char)text[i]];
        bpe_word(w, out);
    };
    size_t i = 0;
    while (i < n) {
        const unsigned c = cps[i];
        // contractions
        if (c == '\'' && i + 1 < n) {
            auto low = [&](size_t k) { unsigned x = cps[k]; return (x >= 'A' && x <= 'Z') ? x + 32 : x; };
            const unsigned c1 = low(i + 1);
            if (c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd') { emit(i, i + 2); i += 2; continue; }
            if (i + 2 < n) {
                const unsigned c2 = low(i + 2);
                if ((c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e') || (c1 == 'l' && c2 == 'l')) { emit(i, i + 3); i += 3; continue; }
            }
        }
        // [^\r\n\p{L}\p{N}]?\p{L}+
        {
            size_t j = i;
            if (!is_letter(c) && !is_digit(c) && c != '\r' && c != '\n' && j + 1 < n && is_letter(cps[j + 1])) j++;
            if (is_letter(cps[j])) {
                size_t k = j;
                while (k < n && is_letter(cps[k])) k++;
                emit(i, k);
                i = k;
                continue;
            }
        }
        if (is_digit(c)) {   // \p{N}{1,3}
            size_t k = i;
            while (k < n && k < i + 3 && is_digit(cps[k])) k++;
            emit(i, k);
            i = k;
            continue;
        }
        // " ?[^\s\p{L}\p{N}]+[\r\n]*"
        {
            size_t j = i;
            if (c == ' ' && j + 1 < n) j++;
            if (!is_space(cps[j]) && !is_letter(cps[j]) && !is_digit(cps[j])) {
                size_t k = j;
                while (k < n && !is_space(cps[k]) && !is_letter(cps[k]) && !is_digit(cps[k])) k++;
                while (k < n && (cps[k] == '\r' || cps[k] == '\n')) k++;
                emit(i, k);
                i = k;
                continue;
            }
        }
        if (is_space(c)) {
            size_t k = i;
            while (k < n && is_space(cps[k])) k++;
            // \s*[\r\n]+ : run of whitespace ending in newlines
            size_t last_nl = 0;
            bool has_nl = false;
            for (size_t t = i; t < k; t++) if (cps[t] == '\r' || cps[t] == '\n') { last_nl = t; has_nl = true; }
            if (has_nl) { emit(i, last_nl + 1); i = last_nl + 1; continue; }
            // \s+(?!\S): leave the last space to prefix the next word when something follows
            if (k < n && k - i > 1) { emit(i, k - 1); i = k - 1; continue; }
            emit(i, k);
            i = k;
            continue;
        }
        emit(i, i + 1);
        i++;
    }
}

// WordPiece (llama.cpp's llm_tokenizer_wpm): see vocab.h
namespace {
bool wpm_is_cjk(unsigned cp) {
    return (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x20000 && cp <= 0x2A6DF) || (cp >= 0x2A700 && cp <= 0x2B73F) ||
           (cp >= 0x2B740 && cp <= 0x2B81F) || (cp >= 0x2B920 && cp <= 0x2CEAF) || (cp >= 0xF900 && cp <= 0xFAFF) || (cp >= 0x2F800 && cp <= 0x2FA1F);
}
bool wpm_is_punct(unsigned cp) {
    if (cp < 0x80) return (cp >= 33 && cp <= 47) || (cp >= 58 && cp <= 64) || (cp >= 91 && cp <= 96) || (cp >= 123 && cp <= 126);   // ASCII punctuation and symbols
    return (cp >= 0x2010 && cp <= 0x2027) || (cp >= 0x2030 && cp <= 0x205E) || (cp >= 0x3001 && cp <= 0x3003) || (cp >= 0x3008 && cp <= 0x3011) ||
           cp == 0xA1 || cp == 0xA7 || cp == 0xAB || cp == 0xB6 || cp == 0xB7 || cp == 0xBB || cp == 0xBF || (cp >= 0xFF01 && cp <= 0xFF0F);
}
// lower-case + base letter of the Latin-1 / Latin Extended-A range (what NFD followed by dropping the combining marks leaves)
unsigned wpm_fold(unsigned cp) {
    if (cp >= 'A' && cp <= 'Z') return cp + 32;
    if (cp < 0xC0) return cp;
    if (cp <= 0xFF) {
        // base letter of 0xC0 .. 0xFF after canonical decomposition, lower-cased; 0 = no ASCII base (the letter keeps itself, lower-cased)
        static const unsigned char base[64] = {97, 97, 97, 97, 97, 97, 0, 99, 101, 101, 101, 101, 105, 105, 105, 105, 0, 110, 111, 111, 111, 111, 111, 0, 0, 117, 117, 117, 117, 121, 0, 0, 97, 97, 97, 97, 97, 97, 0, 99, 101, 101, 101, 101, 105, 105, 105, 105, 0, 110, 111, 111, 111, 111, 111, 0, 0, 117, 117, 117, 117, 121, 0, 121};
        if (base[cp - 0xC0]) return base[cp - 0xC0];
        return (cp <= 0xDE && cp != 0xD7) ? cp + 32 : cp;
    }
    if (cp <= 0x17F) {
        static const unsigned char ext[128] = {97, 97, 97, 97, 97, 97, 99, 99, 99, 99, 99, 99, 99, 99, 100, 100, 0, 0, 101, 101, 101, 101, 101, 101, 101, 101, 101, 101, 103, 103, 103, 103, 103, 103, 103, 103, 104, 104, 0, 0, 105, 105, 105, 105, 105, 105, 105, 105, 105, 0, 0, 0, 106, 106, 107, 107, 0, 108, 108, 108, 108, 108, 108, 0, 0, 0, 0, 110, 110, 110, 110, 110, 110, 0, 0, 0, 111, 111, 111, 111, 111, 111, 0, 0, 114, 114, 114, 114, 114, 114, 115, 115, 115, 115, 115, 115, 115, 115, 116, 116, 116, 116, 0, 0, 117, 117, 117, 117, 117, 117, 117, 117, 117, 117, 117, 117, 119, 119, 121, 121, 121, 122, 122, 122, 122, 122, 122, 0};         // 0x100 .. 0x17F likewise
        if (ext[cp - 0x100]) return ext[cp - 0x100];
        // letters without a decomposition (d / h / l / o / t with stroke, eng, ligatures): Python's str.lower() pairs, which are not all even / odd
        switch (cp) {
            case 0x110: return 0x111; case 0x126: return 0x127; case 0x132: return 0x133; case 0x13F: return 0x140; case 0x141: return 0x142;
            case 0x14A: return 0x14B; case 0x152: return 0x153; case 0x166: return 0x167; default: return cp;
        }
    }
    if (cp >= 0x391 && cp <= 0x3A9 && cp != 0x3A2) return cp + 32;      // Greek capitals
    if (cp >= 0x410 && cp <= 0x42F) return cp + 32;                      // Cyrillic capitals
    if (cp >= 0x400 && cp <= 0x40F) return cp + 80;
    return cp;
}
void utf8_append(std::string &s, unsigned cp) {
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
}
}  // namespace

void Vocab::tokenize_wpm(const std::string &text, std::vector<int32_t> &out) const {
    std::vector<std::string> words(1);
    for (size_t off = 0; off < text.size();) {
        const unsigned char c0 = (unsigned char)text[off];
        size_t l = std::min(utf8_len(c0), text.size() - off);
        unsigned cp = c0;
        if (l == 2) cp = ((c0 & 0x1F) << 6) | ((unsigned char)text[off + 1] & 0x3F);
        else if (l == 3) cp = ((c0 & 0x0F) << 12) | (((unsigned char)text[off + 1] & 0x3F) << 6) | ((unsigned char)text[off + 2] & 0x3F);
        else if (l == 4) cp = ((c0 & 0x07) << 18) | (((unsigned char)text[off + 1] & 0x3F) << 12) | (((unsigned char)text[off + 2] & 0x3F) << 6) | ((unsigned char)text[off + 3] & 0x3F);
        off += l;
        const bool space = is_space(cp) || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) || cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
        if (space) { if (!words.back().empty()) words.emplace_back(); continue; }
        if (cp == 0 || cp == 0xFFFD || cp < 0x20 || (cp >= 0x7F && cp < 0xA0)) continue;                 // control characters
        if (cp >= 0x300 && cp <= 0x36F) continue;                                                          // combining marks
        if (wpm_is_punct(cp) || wpm_is_cjk(cp)) {
            if (!words.back().empty()) words.emplace_back();
            utf8_append(words.back(), wpm_fold(cp));
            words.emplace_back();
        } else {
            utf8_append(words.back(), wpm_fold(cp));
        }
    }
    if (words.back().empty()) words.pop_back();
    for (const std::string &word : words) {
        if (word.empty()) continue;
        const std::string w1 = "\xE2\x96\x81" + word;
        const size_t n = w1.size(), mark = out.size();
        for (size_t i = 0; i < n;) {
            bool match = false;
            for (size_t j = std::min(n, i + max_token_len_ + 1); j > i; j--) {
                auto it = index_.find(w1.substr(i, j - i));
                if (it != index_.end()) { out.push_back(it->second); match = true; i = j; break; }
            }
            if (!match) { out.resize(mark); break; }
        }
        if (out.size() == mark) out.push_back(unk_);
    }
}

std::vector<int32_t> Vocab::tokenize(const std::string &text, bool add_special, bool parse_special) const {
    std::vector<int32_t> out;
    if (model_ == "bert") {
        if (add_special) out.push_back(cls_);
        tokenize_wpm(text, out);
        if (add_special) out.push_back(sep_);
        return out;
    }
    if (add_special && add_bos_ && bos_ >= 0) out.push_back(bos_);
    // split on special tokens first when asked to
    std::vector<std::pair<bool, std::string>> frags;   // (is_special_id, text) ; special carries the id as decimal string
    frags.emplace_back(false, text);
    if (parse_special) {
        for (int sid : special_ids_) {
            const std::string &st = tokens_[(size_t)sid];
            if (st.empty()) continue;
            std::vector<std::pair<bool, std::string>> next;
            for (auto &fr : frags) {
                if (fr.first) { next.push_back(fr); continue; }
                size_t pos = 0;
                while (true) {
                    const size_t hit = fr.second.find(st, pos);
                    if (hit == std::string::npos) { if (pos < fr.second.size()) next.emplace_back(false, fr.second.substr(pos)); break; }
                    if (hit > pos) next.emplace_back(false, fr.second.substr(pos, hit - pos));
                    next.emplace_back(true, std::to_string(sid));
                    pos = hit + st.size();
                }
            }
            frags.swap(next);
        }
    }
    bool first = true;
    for (const auto &fr : frags) {
        if (fr.first) { out.push_back(atoi(fr.second.c_str())); first = false; continue; }
        if (fr.second.empty()) continue;
        if (model_ == "gpt2") {
            tokenize_bpe(fr.second, out);
        } else {
            std::string t;
            if (add_space_prefix_ && first) t = " ";
            t += fr.second;
            std::string esc;
            for (char c : t) { if (c == ' ') esc += "\xE2\x96\x81"; else esc += c; }
            tokenize_spm(esc, out);
        }
        first = false;
    }
    if (add_special && add_eos_ && eos_ >= 0) out.push_back(eos_);
    return out;
}

std::string Vocab::token_to_piece(int32_t id, bool special) const {
    if (id < 0 || id >= n_tokens()) return "";
    const std::string &t = tokens_[(size_t)id];
    const int ty = types_[(size_t)id];
    if (ty == TT_CONTROL || ty == TT_UNKNOWN) return special ? t : "";
    if (model_ == "gpt2") {
        if (ty == TT_USER_DEFINED) return t;
        std::string out;
        for (size_t off = 0; off < t.size();) {
            const size_t l = std::min(utf8_len((unsigned char)t[off]), t.size() - off);
            auto it = byte_map().from_utf8.find(t.substr(off, l));
            if (it != byte_map().from_utf8.end()) out += (char)it->second;
            else out += t.substr(off, l);
            off += l;
        }
        return out;
    }
    if (ty == TT_BYTE && t.size() == 6 && t[0] == '<' && t[1] == '0' && t[2] == 'x') return std::string(1, (char)strtol(t.c_str() + 3, nullptr, 16));
    std::string out;
    for (size_t i = 0; i < t.size();) {
        if (i + 2 < t.size() && (unsigned char)t[i] == 0xE2 && (unsigned char)t[i + 1] == 0x96 && (unsigned char)t[i + 2] == 0x81) { out += ' '; i += 3; }
        else out += t[i++];
    }
    return out;
}

std::string Vocab::detokenize(const std::vector<int32_t> &ids, bool special) const {
    std::string s;
    for (int32_t id : ids) s += token_to_piece(id, special);
    return s;
}

}  // namespace mi355
