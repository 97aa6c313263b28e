// json.h — small self-contained JSON value (parse / dump) for the host-side mirror of the reference's request and
// response bodies.  The reference uses jsoncpp (Json::Value) at the EngineI boundary and nlohmann::json inside the slot
// loop (base/cortex-common/enginei.h:8, src/llama_server_context.h); neither is available offline, so the same shapes are
// carried in this type.  Object keys keep insertion order.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace mi355 {

class Json {
  public:
    enum Type { Null, Bool, Int, Double, String, Array, Object };
    using Members = std::vector<std::pair<std::string, Json>>;

    Json() = default;
    Json(std::nullptr_t) {}
    Json(bool b) : t_(Bool), b_(b) {}
    Json(int v) : t_(Int), i_(v) {}
    Json(unsigned v) : t_(Int), i_(v) {}
    Json(long v) : t_(Int), i_(v) {}
    Json(long long v) : t_(Int), i_(v) {}
    Json(unsigned long v) : t_(Int), i_((int64_t)v) {}
    Json(unsigned long long v) : t_(Int), i_((int64_t)v) {}
    Json(float v) : t_(Double), d_(v) {}
    Json(double v) : t_(Double), d_(v) {}
    Json(const char *s) : t_(String), s_(s ? s : "") {}
    Json(const std::string &s) : t_(String), s_(s) {}
    Json(std::string &&s) : t_(String), s_(std::move(s)) {}

    static Json array() { Json j; j.t_ = Array; return j; }
    static Json object() { Json j; j.t_ = Object; return j; }
    template <typename T> static Json array_of(const std::vector<T> &v) { Json j = array(); for (const auto &e : v) j.push_back(Json(e)); return j; }

    Type type() const { return t_; }
    bool is_null() const { return t_ == Null; }
    bool is_bool() const { return t_ == Bool; }
    bool is_number() const { return t_ == Int || t_ == Double; }
    bool is_int() const { return t_ == Int; }
    bool is_string() const { return t_ == String; }
    bool is_array() const { return t_ == Array; }
    bool is_object() const { return t_ == Object; }

    bool as_bool(bool def = false) const { return t_ == Bool ? b_ : t_ == Int ? i_ != 0 : def; }
    int64_t as_int(int64_t def = 0) const { return t_ == Int ? i_ : t_ == Double ? (int64_t)d_ : t_ == Bool ? (int64_t)b_ : def; }
    double as_double(double def = 0.0) const { return t_ == Double ? d_ : t_ == Int ? (double)i_ : def; }
    const std::string &as_string() const { static const std::string e; return t_ == String ? s_ : e; }
    std::string str_or(const std::string &def) const { return t_ == String ? s_ : def; }

    // arrays
    size_t size() const { return t_ == Array ? a_.size() : t_ == Object ? o_.size() : 0; }
    bool empty() const { return t_ == Null || (t_ == String && s_.empty()) || ((t_ == Array || t_ == Object) && size() == 0); }
    void push_back(Json v) { if (t_ != Array) { *this = array(); } a_.push_back(std::move(v)); }
    const Json &at(size_t i) const { static const Json n; return (t_ == Array && i < a_.size()) ? a_[i] : n; }
    const std::vector<Json> &items() const { return a_; }
    std::vector<Json> &items() { return a_; }

    // objects
    bool contains(const std::string &k) const { return find(k) != nullptr; }
    const Json *find(const std::string &k) const {
        if (t_ != Object) return nullptr;
        for (const auto &m : o_) if (m.first == k) return &m.second;
        return nullptr;
    }
    const Json &operator[](const std::string &k) const { static const Json n; const Json *p = find(k); return p ? *p : n; }
    Json &operator[](const std::string &k) {
        if (t_ != Object) *this = object();
        for (auto &m : o_) if (m.first == k) return m.second;
        o_.emplace_back(k, Json());
        return o_.back().second;
    }
    const Members &members() const { return o_; }
    // typed lookup with default (the reference's json_value helper)
    template <typename T> T value(const std::string &k, const T &def) const;

    std::string dump() const { std::string out; dump_to(out); return out; }

    static bool parse(const std::string &text, Json &out, std::string *err = nullptr) {
        Parser p{text.c_str(), text.c_str() + text.size(), ""};
        p.ws();
        if (!p.value(out)) { if (err) *err = p.err.empty() ? "parse error" : p.err; return false; }
        p.ws();
        if (p.p != p.end) { if (err) *err = "trailing characters"; return false; }
        return true;
    }

  private:
    Type t_ = Null;
    bool b_ = false;
    int64_t i_ = 0;
    double d_ = 0.0;
    std::string s_;
    std::vector<Json> a_;
    Members o_;

    // Strings are emitted as valid UTF-8: a byte that does not start or continue a well-formed sequence is replaced by
    // U+FFFD (generated text can end in, or contain, stray byte-fallback tokens).
    static void dump_string(const std::string &s, std::string &out) {
        out += '"';
        const size_t n = s.size();
        for (size_t i = 0; i < n;) {
            const unsigned char c = (unsigned char)s[i];
            if (c < 0x80) {
                switch (c) {
                    case '"': out += "\\\""; break;
                    case '\\': out += "\\\\"; break;
                    case '\n': out += "\\n"; break;
                    case '\r': out += "\\r"; break;
                    case '\t': out += "\\t"; break;
                    case '\b': out += "\\b"; break;
                    case '\f': out += "\\f"; break;
                    default:
                        if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); out += b; }
                        else out += (char)c;
                }
                i++;
                continue;
            }
            size_t len = 0;
            uint32_t cp = 0;
            if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1Fu; }
            else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0Fu; }
            else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07u; }
            bool ok = len != 0 && i + len <= n;
            for (size_t k = 1; ok && k < len; k++) {
                const unsigned char cc = (unsigned char)s[i + k];
                if ((cc & 0xC0) != 0x80) ok = false;
                cp = (cp << 6) | (cc & 0x3Fu);
            }
            if (ok && ((len == 3 && (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF))) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)))) ok = false;
            if (ok) { out.append(s, i, len); i += len; }
            else { out += "\xEF\xBF\xBD"; i++; }
        }
        out += '"';
    }
    void dump_to(std::string &out) const {
        switch (t_) {
            case Null: out += "null"; break;
            case Bool: out += b_ ? "true" : "false"; break;
            case Int: out += std::to_string(i_); break;
            case Double: {
                if (!std::isfinite(d_)) { out += "null"; break; }
                char b[40];
                snprintf(b, sizeof b, "%.17g", d_);
                if (!strpbrk(b, ".eE")) { size_t n = strlen(b); b[n] = '.'; b[n + 1] = '0'; b[n + 2] = 0; }
                out += b;
                break;
            }
            case String: dump_string(s_, out); break;
            case Array:
                out += '[';
                for (size_t i = 0; i < a_.size(); i++) { if (i) out += ','; a_[i].dump_to(out); }
                out += ']';
                break;
            case Object:
                out += '{';
                for (size_t i = 0; i < o_.size(); i++) {
                    if (i) out += ',';
                    dump_string(o_[i].first, out);
                    out += ':';
                    o_[i].second.dump_to(out);
                }
                out += '}';
                break;
        }
    }

    struct Parser {
        const char *p, *end;
        std::string err;
        void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++; }
        bool lit(const char *s) { size_t n = strlen(s); if ((size_t)(end - p) >= n && !memcmp(p, s, n)) { p += n; return true; } return false; }
        static void utf8(unsigned cp, std::string &o) {
            if (cp < 0x80) o += (char)cp;
            else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
            else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
        }
        bool hex4(unsigned &v) {
            if (end - p < 4) return false;
            v = 0;
            for (int i = 0; i < 4; i++) {
                const char c = *p++;
                v <<= 4;
                if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
                else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
                else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
                else return false;
            }
            return true;
        }
        bool string(std::string &o) {
            if (p >= end || *p != '"') { err = "expected string"; return false; }
            p++;
            while (p < end && *p != '"') {
                if (*p == '\\') {
                    p++;
                    if (p >= end) return false;
                    const char c = *p++;
                    switch (c) {
                        case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
                        case 'b': o += '\b'; break; case 'f': o += '\f'; break; case '/': o += '/'; break;
                        case '\\': o += '\\'; break; case '"': o += '"'; break;
                        case 'u': {
                            unsigned cp;
                            if (!hex4(cp)) { err = "bad \\u escape"; return false; }
                            if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                                p += 2;
                                unsigned lo;
                                if (!hex4(lo)) return false;
                                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            }
                            utf8(cp, o);
                            break;
                        }
                        default: err = "bad escape"; return false;
                    }
                } else {
                    o += *p++;
                }
            }
            if (p >= end) { err = "unterminated string"; return false; }
            p++;
            return true;
        }
        bool value(Json &out) {
            ws();
            if (p >= end) { err = "unexpected end"; return false; }
            if (*p == '{') {
                p++;
                out = Json::object();
                ws();
                if (p < end && *p == '}') { p++; return true; }
                while (true) {
                    ws();
                    std::string k;
                    if (!string(k)) return false;
                    ws();
                    if (p >= end || *p != ':') { err = "expected ':'"; return false; }
                    p++;
                    Json v;
                    if (!value(v)) return false;
                    out.o_.emplace_back(std::move(k), std::move(v));
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == '}') { p++; return true; }
                    err = "expected ',' or '}'";
                    return false;
                }
            }
            if (*p == '[') {
                p++;
                out = Json::array();
                ws();
                if (p < end && *p == ']') { p++; return true; }
                while (true) {
                    Json v;
                    if (!value(v)) return false;
                    out.a_.push_back(std::move(v));
                    ws();
                    if (p < end && *p == ',') { p++; continue; }
                    if (p < end && *p == ']') { p++; return true; }
                    err = "expected ',' or ']'";
                    return false;
                }
            }
            if (*p == '"') { std::string s; if (!string(s)) return false; out = Json(std::move(s)); return true; }
            if (lit("true")) { out = Json(true); return true; }
            if (lit("false")) { out = Json(false); return true; }
            if (lit("null")) { out = Json(); return true; }
            const char *s = p;
            bool is_d = false;
            if (p < end && (*p == '-' || *p == '+')) p++;
            while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '-' || *p == '+')) {
                if (*p == '.' || *p == 'e' || *p == 'E') is_d = true;
                p++;
            }
            if (p == s) { err = "unexpected character"; return false; }
            const std::string num(s, p);
            if (is_d) out = Json(strtod(num.c_str(), nullptr));
            else out = Json((long long)strtoll(num.c_str(), nullptr, 10));
            return true;
        }
    };
};

template <> inline bool Json::value<bool>(const std::string &k, const bool &def) const { const Json *p = find(k); return (p && !p->is_null()) ? p->as_bool(def) : def; }
template <> inline int Json::value<int>(const std::string &k, const int &def) const { const Json *p = find(k); return (p && p->is_number()) ? (int)p->as_int(def) : def; }
template <> inline int64_t Json::value<int64_t>(const std::string &k, const int64_t &def) const { const Json *p = find(k); return (p && p->is_number()) ? p->as_int(def) : def; }
template <> inline uint32_t Json::value<uint32_t>(const std::string &k, const uint32_t &def) const { const Json *p = find(k); return (p && p->is_number()) ? (uint32_t)p->as_int(def) : def; }
template <> inline float Json::value<float>(const std::string &k, const float &def) const { const Json *p = find(k); return (p && p->is_number()) ? (float)p->as_double(def) : def; }
template <> inline double Json::value<double>(const std::string &k, const double &def) const { const Json *p = find(k); return (p && p->is_number()) ? p->as_double(def) : def; }
template <> inline std::string Json::value<std::string>(const std::string &k, const std::string &def) const { const Json *p = find(k); return (p && p->is_string()) ? p->as_string() : def; }

}  // namespace mi355
