// grammar.cc — GBNF text -> rules, and the breadth-first pushdown matcher over them (see grammar.h).
#include "grammar.h"

#include <algorithm>
#include <map>
#include <set>

namespace mi355 {

bool CharSet::touches(uint32_t lo, uint32_t hi) const {
    if (!negated) {
        for (const auto &r : ranges) if (r.first <= hi && r.second >= lo) return true;
        return false;
    }
    // negated: is some code point of [lo, hi] outside every listed range?
    std::vector<std::pair<uint32_t, uint32_t>> s(ranges);
    std::sort(s.begin(), s.end());
    uint64_t cur = lo;
    for (const auto &r : s) {
        if (r.second < cur) continue;
        if (r.first > cur) return true;
        cur = (uint64_t)r.second + 1;
        if (cur > hi) return false;
    }
    return cur <= hi;
}

// ---------------------------------------------------------------------------------------------------------------- GBNF text
namespace {

struct GbnfParser {
    const char *p, *end;
    std::vector<GrammarRule> rules;
    std::vector<bool> defined;
    std::map<std::string, int> ids;
    std::string err;

    bool fail(const std::string &m) {
        if (err.empty()) {
            int line = 1;
            for (const char *q = begin; q < p && q < end; q++) if (*q == '\n') line++;
            err = m + " (line " + std::to_string(line) + ")";
        }
        return false;
    }
    const char *begin = nullptr;

    int rule_id(const std::string &name) {
        auto it = ids.find(name);
        if (it != ids.end()) return it->second;
        const int id = (int)rules.size();
        rules.push_back(GrammarRule{name, {}});
        defined.push_back(false);
        ids[name] = id;
        return id;
    }
    int fresh_rule(const std::string &base) {
        for (int n = (int)rules.size();; n++) {
            const std::string name = base + "_" + std::to_string(n);
            if (!ids.count(name)) { const int id = rule_id(name); defined[(size_t)id] = true; return id; }
        }
    }
    void space(bool newline_ok) {
        while (p < end) {
            if (*p == ' ' || *p == '\t') p++;
            else if (*p == '#') { while (p < end && *p != '\n' && *p != '\r') p++; }
            else if (newline_ok && (*p == '\n' || *p == '\r')) p++;
            else break;
        }
    }
    static bool name_char(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '-' || c == '_'; }
    bool name(std::string &out) {
        const char *s = p;
        while (p < end && name_char(*p)) p++;
        if (p == s) return fail("expected a rule name");
        out.assign(s, p);
        return true;
    }
    static int hexval(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
    // one (possibly escaped, possibly multi-byte) character of a literal or a class
    bool one_char(uint32_t &cp) {
        if (p >= end) return fail("unexpected end of grammar");
        const unsigned char c = (unsigned char)*p;
        if (c == '\\') {
            if (p + 1 >= end) return fail("dangling backslash");
            const char e = p[1];
            p += 2;
            int nhex = 0;
            switch (e) {
                case 'n': cp = '\n'; return true;
                case 'r': cp = '\r'; return true;
                case 't': cp = '\t'; return true;
                case '\\': case '"': case '[': case ']': case '.': case '-': case '^': case '/': cp = (unsigned char)e; return true;
                case 'x': nhex = 2; break;
                case 'u': nhex = 4; break;
                case 'U': nhex = 8; break;
                default: return fail(std::string("unknown escape \\") + e);
            }
            cp = 0;
            for (int i = 0; i < nhex; i++) {
                if (p >= end || hexval(*p) < 0) return fail("bad hexadecimal escape");
                cp = (cp << 4) | (uint32_t)hexval(*p++);
            }
            return true;
        }
        int n = c < 0x80 ? 0 : (c & 0xE0) == 0xC0 ? 1 : (c & 0xF0) == 0xE0 ? 2 : (c & 0xF8) == 0xF0 ? 3 : -1;
        if (n < 0 || p + n >= end + (n == 0 ? 1 : 0)) return fail("invalid UTF-8 in grammar text");
        cp = n == 0 ? c : (uint32_t)(c & (0xFF >> (n + 2)));
        p++;
        for (int i = 0; i < n; i++) {
            if (p >= end || ((unsigned char)*p & 0xC0) != 0x80) return fail("invalid UTF-8 in grammar text");
            cp = (cp << 6) | ((unsigned char)*p++ & 0x3F);
        }
        return true;
    }
    static GrammarSymbol sym_of(uint32_t lo, uint32_t hi) { GrammarSymbol s; s.set.ranges.push_back({lo, hi}); return s; }
    static GrammarSymbol ref_of(int rule) { GrammarSymbol s; s.rule = rule; return s; }

    // sub{lo,hi} (hi < 0: unbounded) appended to seq
    void repeat(GrammarSeq &seq, const GrammarSeq &sub, int lo, int hi, const std::string &owner) {
        for (int i = 0; i < lo; i++) seq.insert(seq.end(), sub.begin(), sub.end());
        if (hi < 0) {                                   // R ::= sub R |
            const int r = fresh_rule(owner);
            GrammarSeq a(sub);
            a.push_back(ref_of(r));
            rules[(size_t)r].alts = {a, GrammarSeq()};
            seq.push_back(ref_of(r));
            return;
        }
        int last = -1;                                  // O_k ::= sub O_(k-1) |      (hi - lo of them, innermost first)
        for (int i = 0; i < hi - lo; i++) {
            const int r = fresh_rule(owner);
            GrammarSeq a(sub);
            if (last >= 0) a.push_back(ref_of(last));
            rules[(size_t)r].alts = {a, GrammarSeq()};
            last = r;
        }
        if (last >= 0) seq.push_back(ref_of(last));
    }

    bool sequence(GrammarSeq &seq, const std::string &owner, bool nested) {
        size_t last_start = seq.size();
        while (p < end) {
            const char c = *p;
            if (c == '"') {
                p++;
                last_start = seq.size();
                while (p < end && *p != '"') {
                    uint32_t cp;
                    if (!one_char(cp)) return false;
                    seq.push_back(sym_of(cp, cp));
                }
                if (p >= end) return fail("unterminated string literal");
                p++;
                space(nested);
            } else if (c == '[') {
                p++;
                GrammarSymbol s;
                if (p < end && *p == '^') { s.set.negated = true; p++; }
                while (p < end && *p != ']') {
                    uint32_t lo, hi;
                    if (!one_char(lo)) return false;
                    hi = lo;
                    if (p + 1 < end && *p == '-' && p[1] != ']') { p++; if (!one_char(hi)) return false; }
                    if (hi < lo) return fail("character range out of order");
                    s.set.ranges.push_back({lo, hi});
                }
                if (p >= end) return fail("unterminated character class");
                p++;
                last_start = seq.size();
                seq.push_back(s);
                space(nested);
            } else if (c == '.') {
                p++;
                GrammarSymbol s;
                s.set.negated = true;                   // "not in the empty set": any code point
                last_start = seq.size();
                seq.push_back(s);
                space(nested);
            } else if (c == '(') {
                p++;
                space(true);
                const int sub = fresh_rule(owner);
                if (!alternates(sub, owner, true)) return false;
                if (p >= end || *p != ')') return fail("expected )");
                p++;
                last_start = seq.size();
                seq.push_back(ref_of(sub));
                space(nested);
            } else if (name_char(c)) {
                std::string nm;
                if (!name(nm)) return false;
                last_start = seq.size();
                seq.push_back(ref_of(rule_id(nm)));
                space(nested);
            } else if (c == '*' || c == '+' || c == '?' || c == '{') {
                if (last_start == seq.size()) return fail(std::string("nothing to repeat before ") + c);
                int lo = 0, hi = -1;
                if (c == '*') { p++; }
                else if (c == '+') { lo = 1; p++; }
                else if (c == '?') { hi = 1; p++; }
                else {
                    p++;
                    space(nested);
                    auto number = [&](int &v) { const char *s = p; long n = 0; while (p < end && *p >= '0' && *p <= '9') { n = n * 10 + (*p++ - '0'); if (n > 100000) return false; } v = (int)n; return p > s; };
                    if (!number(lo)) return fail("expected a repetition count");
                    space(nested);
                    hi = lo;
                    if (p < end && *p == ',') {
                        p++;
                        space(nested);
                        hi = -1;
                        if (p < end && *p != '}') { if (!number(hi)) return fail("expected a repetition count"); if (hi < lo) return fail("repetition bounds out of order"); }
                        space(nested);
                    }
                    if (p >= end || *p != '}') return fail("expected }");
                    p++;
                }
                const GrammarSeq sub(seq.begin() + (long)last_start, seq.end());
                seq.resize(last_start);
                repeat(seq, sub, lo, hi, owner);
                last_start = seq.size();                // (a second operator in a row has nothing to apply to)
                space(nested);
            } else {
                break;
            }
        }
        return true;
    }
    bool alternates(int rule, const std::string &owner, bool nested) {
        std::vector<GrammarSeq> alts;
        for (;;) {
            GrammarSeq seq;
            if (!sequence(seq, owner, nested)) return false;
            alts.push_back(std::move(seq));
            if (p < end && *p == '|') { p++; space(true); continue; }
            break;
        }
        rules[(size_t)rule].alts = std::move(alts);
        return true;
    }
    bool rule() {
        std::string nm;
        if (!name(nm)) return false;
        space(false);
        if (p + 2 >= end || p[0] != ':' || p[1] != ':' || p[2] != '=') return fail("expected ::= after " + nm);
        p += 3;
        space(true);
        const int id = rule_id(nm);
        if (defined[(size_t)id]) return fail("rule " + nm + " is defined twice");
        defined[(size_t)id] = true;
        if (!alternates(id, nm, false)) return false;
        if (p < end && *p != '\n' && *p != '\r') return fail("unexpected character in rule " + nm);
        space(true);
        return true;
    }
};

}  // namespace

std::shared_ptr<const Grammar> Grammar::parse(const std::string &text, std::string &err, const std::string &root) {
    GbnfParser ps;
    ps.p = ps.begin = text.data();
    ps.end = text.data() + text.size();
    ps.space(true);
    while (ps.p < ps.end) {
        if (!ps.rule()) { err = ps.err; return nullptr; }
    }
    if (ps.rules.empty()) { err = "empty grammar"; return nullptr; }
    for (size_t i = 0; i < ps.rules.size(); i++) if (!ps.defined[i]) { err = "undefined rule " + ps.rules[i].name; return nullptr; }
    auto it = ps.ids.find(root);
    if (it == ps.ids.end()) { err = "grammar has no '" + root + "' rule"; return nullptr; }
    // left recursion would make the matcher expand a rule for ever: find it here.  nullable rules first, then the "can start with" graph
    const size_t n = ps.rules.size();
    std::vector<bool> nullable(n, false);
    for (bool again = true; again;) {
        again = false;
        for (size_t r = 0; r < n; r++) {
            if (nullable[r]) continue;
            for (const auto &alt : ps.rules[r].alts) {
                bool all = true;
                for (const auto &s : alt) if (s.rule < 0 || !nullable[(size_t)s.rule]) { all = false; break; }
                if (all) { nullable[r] = true; again = true; break; }
            }
        }
    }
    std::vector<std::set<int>> first(n);
    for (size_t r = 0; r < n; r++)
        for (const auto &alt : ps.rules[r].alts)
            for (const auto &s : alt) {
                if (s.rule < 0) break;
                first[r].insert(s.rule);
                if (!nullable[(size_t)s.rule]) break;
            }
    for (size_t r = 0; r < n; r++) {
        std::set<int> seen;
        std::vector<int> todo(first[r].begin(), first[r].end());
        while (!todo.empty()) {
            const int t = todo.back();
            todo.pop_back();
            if (t == (int)r) { err = "left recursion in rule " + ps.rules[r].name; return nullptr; }
            if (!seen.insert(t).second) continue;
            todo.insert(todo.end(), first[(size_t)t].begin(), first[(size_t)t].end());
        }
    }
    auto g = std::make_shared<Grammar>();
    g->rules_ = std::move(ps.rules);
    g->root_ = it->second;
    return g;
}

// ---------------------------------------------------------------------------------------------------------------- matching
static constexpr size_t kMaxStacks = 8192;
GrammarMatcher::GrammarMatcher(std::shared_ptr<const Grammar> g) : g_(std::move(g)) { reset(); }

void GrammarMatcher::reset() {
    stacks_.clear();
    tail_ = Utf8Tail();
    if (!g_) return;
    const auto &root = g_->rules()[(size_t)g_->root()];
    for (size_t a = 0; a < root.alts.size(); a++) settle(Stack{Frame{g_->root(), (int)a, 0}}, stacks_);
}

bool GrammarMatcher::same(const Stack &a, const Stack &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) if (a[i].rule != b[i].rule || a[i].alt != b[i].alt || a[i].pos != b[i].pos) return false;
    return true;
}

void GrammarMatcher::settle(Stack st, std::vector<Stack> &out) const {
    for (;;) {
        if (st.empty()) break;                                                   // the grammar can end here
        const Frame f = st.back();
        const GrammarSeq &seq = g_->rules()[(size_t)f.rule].alts[(size_t)f.alt];
        if (f.pos >= (int)seq.size()) { st.pop_back(); continue; }               // this alternative is finished
        const GrammarSymbol &s = seq[(size_t)f.pos];
        if (s.rule < 0) break;                                                   // a range symbol on top: settled
        // a reference: step past it in this frame (dropping the frame if that was its last symbol), then try every alternative of the rule
        if (f.pos + 1 >= (int)seq.size()) st.pop_back(); else st.back().pos++;
        const auto &alts = g_->rules()[(size_t)s.rule].alts;
        for (size_t a = 0; a < alts.size(); a++) {
            Stack next(st);
            next.push_back(Frame{s.rule, (int)a, 0});
            settle(std::move(next), out);
        }
        return;
    }
    for (const auto &o : out) if (same(o, st)) return;
    if (out.size() >= kMaxStacks) return;      // a pathologically ambiguous grammar: keep what there is rather than grow without bound
    out.push_back(std::move(st));
}

const CharSet *GrammarMatcher::top_set(const Stack &st) const {
    if (st.empty()) return nullptr;
    const Frame &f = st.back();
    return &g_->rules()[(size_t)f.rule].alts[(size_t)f.alt][(size_t)f.pos].set;
}

void GrammarMatcher::step(const std::vector<Stack> &from, uint32_t cp, std::vector<Stack> &to) const {
    to.clear();
    for (const auto &st : from) {
        const CharSet *cs = top_set(st);
        if (!cs || !cs->has(cp)) continue;
        Stack next(st);
        next.back().pos++;
        settle(std::move(next), to);
    }
}

bool GrammarMatcher::run(const std::string &piece, std::vector<Stack> &stacks, Utf8Tail &tail) const {
    std::vector<Stack> tmp;
    for (const char ch : piece) {
        const unsigned char b = (unsigned char)ch;
        uint32_t cp = 0;
        bool whole = false;
        if (tail.remain > 0) {
            if ((b & 0xC0) != 0x80) { tail.remain = -1; return false; }
            tail.value = (tail.value << 6) | (b & 0x3Fu);
            if (--tail.remain == 0) {
                cp = tail.value;
                whole = true;
                if (cp < tail.least || cp > 0x10FFFFu || (cp >= 0xD800u && cp <= 0xDFFFu)) { tail.remain = -1; return false; }   // overlong form, out of range, surrogate
            }
        } else if (b < 0x80) { cp = b; whole = true; }
        else if ((b & 0xE0) == 0xC0) { tail.value = b & 0x1Fu; tail.remain = 1; tail.least = 0x80u; }
        else if ((b & 0xF0) == 0xE0) { tail.value = b & 0x0Fu; tail.remain = 2; tail.least = 0x800u; }
        else if ((b & 0xF8) == 0xF0) { tail.value = b & 0x07u; tail.remain = 3; tail.least = 0x10000u; }
        else { tail.remain = -1; return false; }
        if (whole) {
            step(stacks, cp, tmp);
            stacks.swap(tmp);
            if (stacks.empty()) return false;
        }
    }
    if (tail.remain > 0) {          // the piece stops inside a code point: the bits seen so far pin it down to a range
        const int sh = 6 * tail.remain;
        uint32_t lo = tail.value << sh, hi = lo | ((1u << sh) - 1u);
        if (lo < tail.least) lo = tail.least;                       // (only well-formed completions count)
        if (hi > 0x10FFFFu) hi = 0x10FFFFu;
        if (lo > hi) return false;
        for (const auto &st : stacks) { const CharSet *cs = top_set(st); if (cs && cs->touches(lo, hi)) return true; }
        return false;
    }
    return true;
}

bool GrammarMatcher::admits(const std::string &piece) const {
    if (piece.empty() || stacks_.empty() || tail_.remain < 0) return false;
    if (tail_.remain == 0 && (unsigned char)piece[0] < 0x80) {       // cheap refusal on the first byte: most of a vocabulary fails here
        bool any = false;
        for (const auto &st : stacks_) { const CharSet *cs = top_set(st); if (cs && cs->has((unsigned char)piece[0])) { any = true; break; } }
        if (!any) return false;
    }
    std::vector<Stack> stacks(stacks_);
    Utf8Tail tail = tail_;
    return run(piece, stacks, tail);
}

bool GrammarMatcher::accept(const std::string &piece) {
    if (piece.empty()) return !stacks_.empty();
    if (!run(piece, stacks_, tail_)) { stacks_.clear(); return false; }
    return true;
}

bool GrammarMatcher::can_end() const {
    if (tail_.remain != 0) return false;
    for (const auto &st : stacks_) if (st.empty()) return true;
    return false;
}

}  // namespace mi355
