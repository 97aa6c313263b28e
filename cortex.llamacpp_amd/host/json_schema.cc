// json_schema.cc — response_format's JSON schema as a GBNF grammar (grammar.h: json_schema_to_gbnf).  The reference hands the schema of
// `response_format` {type: "json_object" | "json_schema"} to llama.cpp's json_schema_to_grammar (src/llama_engine.cc:794-801; the converter is
// in the absent submodule).  This one is written against the JSON Schema keywords themselves: every schema node becomes a rule whose language
// is the compact-or-spaced JSON text of the values the node admits.
#include <map>
#include <set>

#include "grammar.h"
#include "json.h"

namespace mi355 {
namespace {

const char *kWs = "[ \\t\\n]{0,20}";
const char *kChar = "[^\"\\\\\\x00-\\x1F\\x7F] | \"\\\\\" ( [\"\\\\/bfnrt] | \"u\" [0-9a-fA-F]{4} )";
const char *kString = "\"\\\"\" char* \"\\\"\"";
const char *kInteger = "\"-\"? ( \"0\" | [1-9] [0-9]{0,15} )";
const char *kNumber = "integer ( \".\" [0-9]{1,16} )? ( [eE] [-+]? [0-9]{1,3} )?";
const char *kValue = "object | array | string | number | boolean | null";
const char *kObject = "\"{\" ws ( string ws \":\" ws value ( ws \",\" ws string ws \":\" ws value )* ws )? \"}\"";
const char *kArray = "\"[\" ws ( value ( ws \",\" ws value )* ws )? \"]\"";

struct Conv {
    const Json &root;
    std::vector<std::pair<std::string, std::string>> rules;   // in order of creation
    std::map<std::string, size_t> index;
    std::map<std::string, std::string> refs;                   // $ref -> rule name
    std::string err;
    int depth = 0;

    explicit Conv(const Json &r) : root(r) {}

    bool fail(const std::string &m) { if (err.empty()) err = m; return false; }

    static std::string quote(const std::string &text) {        // text as a GBNF string literal
        std::string o = "\"";
        for (const char c : text) {
            switch (c) {
                case '\\': o += "\\\\"; break;
                case '"': o += "\\\""; break;
                case '\n': o += "\\n"; break;
                case '\r': o += "\\r"; break;
                case '\t': o += "\\t"; break;
                default: o += c;
            }
        }
        return o + "\"";
    }
    static std::string clean(const std::string &hint) {
        std::string o;
        for (const char c : hint) o += ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) ? c : '-';
        if (o.empty()) o = "r";
        return o;
    }
    // a rule called `hint` (or hint-N if that name is taken by another body)
    std::string add(const std::string &hint, const std::string &body) {
        std::string name = clean(hint);
        for (int n = 1;; n++) {
            auto it = index.find(name);
            if (it == index.end()) break;
            if (rules[it->second].second == body) return name;
            name = clean(hint) + "-" + std::to_string(n);
        }
        index[name] = rules.size();
        rules.emplace_back(name, body);
        return name;
    }
    std::string reserve(const std::string &hint) {             // a name whose body comes later (recursive $ref)
        std::string name = clean(hint);
        for (int n = 1; index.count(name); n++) name = clean(hint) + "-" + std::to_string(n);
        index[name] = rules.size();
        rules.emplace_back(name, "");
        return name;
    }
    std::string prim(const std::string &which) {
        if (index.count(which)) return which;
        if (which == "ws") return add("ws", kWs);
        if (which == "char") return add("char", kChar);
        if (which == "string") { prim("char"); return add("string", kString); }
        if (which == "integer") return add("integer", kInteger);
        if (which == "number") { prim("integer"); return add("number", kNumber); }
        if (which == "boolean") return add("boolean", "\"true\" | \"false\"");
        if (which == "null") return add("null", "\"null\"");
        if (which == "value" || which == "object" || which == "array") {
            // mutually recursive: reserve the three names first
            const bool have = index.count("value") != 0;
            if (!have) {
                reserve("value"); reserve("object"); reserve("array");
                prim("ws"); prim("string"); prim("number"); prim("boolean"); prim("null");
                rules[index["value"]].second = kValue;
                rules[index["object"]].second = kObject;
                rules[index["array"]].second = kArray;
            }
            return which;
        }
        return which;
    }

    const Json *resolve(const std::string &ref) {
        if (ref.compare(0, 2, "#/") != 0) return nullptr;
        const Json *cur = &root;
        size_t at = 2;
        while (at <= ref.size()) {
            size_t sl = ref.find('/', at);
            if (sl == std::string::npos) sl = ref.size();
            std::string key = ref.substr(at, sl - at);
            for (size_t k; (k = key.find("~1")) != std::string::npos;) key.replace(k, 2, "/");
            for (size_t k; (k = key.find("~0")) != std::string::npos;) key.replace(k, 2, "~");
            if (cur->is_object()) { cur = cur->find(key); if (!cur) return nullptr; }
            else if (cur->is_array()) { cur = &cur->at((size_t)atoi(key.c_str())); if (cur->is_null()) return nullptr; }
            else return nullptr;
            at = sl + 1;
        }
        return cur;
    }

    std::string literal(const Json &v) { return quote(v.dump()); }

    // the rule (name or inline expression in parentheses) for one schema node
    bool visit(const Json &s, const std::string &hint, std::string &out) {
        if (++depth > 64) return fail("schema nests too deeply");
        const bool ok = visit_inner(s, hint, out);
        depth--;
        return ok;
    }
    bool visit_inner(const Json &s, const std::string &hint, std::string &out) {
        if (s.is_bool()) {
            if (!s.as_bool()) return fail("a schema of `false` admits nothing");
            out = prim("value");
            return true;
        }
        if (!s.is_object() || s.size() == 0) { out = prim(hint == "root" ? "object" : "value"); return true; }   // no constraint (root: the json_object mode)
        if (s["$ref"].is_string()) {
            const std::string &ref = s["$ref"].as_string();
            auto it = refs.find(ref);
            if (it != refs.end()) { out = it->second; return true; }
            const Json *target = resolve(ref);
            if (!target) return fail("cannot resolve $ref " + ref);
            const size_t sl = ref.rfind('/');
            const std::string name = reserve(sl == std::string::npos ? "ref" : ref.substr(sl + 1));
            refs[ref] = name;
            std::string body;
            if (!visit(*target, name + "-def", body)) return false;
            rules[index[name]].second = body;
            out = name;
            return true;
        }
        if (s.contains("const")) { out = add(hint, literal(s["const"])); return true; }
        if (s["enum"].is_array()) {
            if (s["enum"].size() == 0) return fail("empty enum");
            std::string body;
            for (const auto &v : s["enum"].items()) body += (body.empty() ? "" : " | ") + literal(v);
            out = add(hint, body);
            return true;
        }
        for (const char *kw : {"anyOf", "oneOf"}) {
            if (!s[kw].is_array()) continue;
            if (s[kw].size() == 0) return fail(std::string("empty ") + kw);
            std::string body;
            int i = 0;
            for (const auto &alt : s[kw].items()) {
                std::string r;
                if (!visit(alt, hint + "-" + std::to_string(i++), r)) return false;
                body += (body.empty() ? "" : " | ") + r;
            }
            out = add(hint, body);
            return true;
        }
        if (s["allOf"].is_array()) {                            // objects only: the union of the properties, every `required` list
            Json merged = Json::object();
            merged["type"] = "object";
            Json props = Json::object(), req = Json::array();
            for (const auto &part0 : s["allOf"].items()) {
                const Json *part = &part0;
                if ((*part)["$ref"].is_string()) { part = resolve((*part)["$ref"].as_string()); if (!part) return fail("cannot resolve $ref in allOf"); }
                if ((*part)["properties"].is_object()) for (const auto &kv : (*part)["properties"].members()) props[kv.first] = kv.second;
                else if (part->contains("type") && (*part)["type"].str_or("") != "object") return fail("allOf is supported for object schemas only");
                if ((*part)["required"].is_array()) for (const auto &r : (*part)["required"].items()) req.push_back(r);
            }
            merged["properties"] = props;
            merged["required"] = req;
            if (s.contains("additionalProperties")) merged["additionalProperties"] = s["additionalProperties"];
            return visit(merged, hint, out);
        }
        if (s["type"].is_array()) {
            std::string body;
            for (const auto &t : s["type"].items()) {
                Json one = s;
                one["type"] = t;
                std::string r;
                if (!visit(one, hint + "-" + t.str_or("t"), r)) return false;
                body += (body.empty() ? "" : " | ") + r;
            }
            if (body.empty()) return fail("empty type list");
            out = add(hint, body);
            return true;
        }
        std::string type = s["type"].str_or("");
        if (type.empty()) {
            if (s.contains("properties") || s.contains("additionalProperties") || s.contains("required")) type = "object";
            else if (s.contains("items") || s.contains("prefixItems")) type = "array";
            else if (s.contains("minLength") || s.contains("maxLength") || s.contains("pattern") || s.contains("format")) type = "string";
            else { out = prim("value"); return true; }
        }
        if (type == "boolean" || type == "null" || type == "number" || type == "integer") { out = prim(type); return true; }
        if (type == "string") {
            const int lo = s.value<int>("minLength", 0), hi = s.value<int>("maxLength", -1);
            if (lo <= 0 && hi < 0) { out = prim("string"); return true; }
            if (hi >= 0 && hi < lo) return fail("maxLength below minLength");
            prim("char");
            std::string rep = "{" + std::to_string(lo < 0 ? 0 : lo) + ",";
            if (hi >= 0 && hi - lo <= 4096) rep += std::to_string(hi);
            rep += "}";
            out = add(hint, "\"\\\"\" char" + rep + " \"\\\"\"");
            return true;
        }
        if (type == "array") {
            prim("ws");
            if (s["prefixItems"].is_array() && s["prefixItems"].size() > 0) {      // a tuple: one item per position
                std::string body = "\"[\" ws ";
                int i = 0;
                for (const auto &it : s["prefixItems"].items()) {
                    std::string r;
                    if (!visit(it, hint + "-" + std::to_string(i), r)) return false;
                    body += (i ? "ws \",\" ws " : "") + r + " ";
                    i++;
                }
                out = add(hint, body + "ws \"]\"");
                return true;
            }
            std::string item;
            if (s.contains("items")) { if (!visit(s["items"], hint + "-item", item)) return false; }
            else item = prim("value");
            const int lo = std::max(0, s.value<int>("minItems", 0)), hi = s.value<int>("maxItems", -1);
            if (hi >= 0 && hi < lo) return fail("maxItems below minItems");
            if (hi == 0) { out = add(hint, "\"[\" ws \"]\""); return true; }
            std::string more = "( ws \",\" ws " + item + " )";
            std::string rep = "{" + std::to_string(lo > 0 ? lo - 1 : 0) + "," + (hi >= 0 ? std::to_string(hi - 1) : "") + "}";
            std::string inner = item + " " + more + rep + " ws";
            out = add(hint, lo > 0 ? "\"[\" ws " + inner + " \"]\"" : "\"[\" ws ( " + inner + " )? \"]\"");
            return true;
        }
        if (type == "object") {
            prim("ws");
            const Json &props = s["properties"];
            const bool has_props = props.is_object() && props.size() > 0;
            const Json &ap = s["additionalProperties"];
            // (absent additionalProperties: JSON Schema allows extras; with declared properties the grammar keeps to them, as a model asked for a
            // shape should produce that shape - the choice upstream's converter makes too)
            const bool extras = has_props ? (ap.is_object() || (ap.is_bool() && ap.as_bool())) : !(ap.is_bool() && !ap.as_bool());
            std::string extra_kv;
            if (extras) {
                std::string vr;
                if (ap.is_object()) { if (!visit(ap, hint + "-extra", vr)) return false; }
                else vr = prim("value");
                extra_kv = add(hint + "-extra-kv", prim("string") + " ws \":\" ws " + vr);
            }
            if (!has_props) {
                if (!extras) { out = add(hint, "\"{\" ws \"}\""); return true; }
                out = add(hint, "\"{\" ws ( " + extra_kv + " ( ws \",\" ws " + extra_kv + " )* ws )? \"}\"");
                return true;
            }
            std::set<std::string> required;
            if (s["required"].is_array()) for (const auto &r : s["required"].items()) required.insert(r.str_or(""));
            std::vector<std::string> req_kv, opt_kv;
            for (const auto &kv : props.members()) {
                std::string vr;
                if (!visit(kv.second, hint + "-" + kv.first, vr)) return false;
                const std::string rule = add(hint + "-" + kv.first + "-kv", quote(Json(kv.first).dump()) + " ws \":\" ws " + vr);
                (required.count(kv.first) ? req_kv : opt_kv).push_back(rule);
            }
            // optional members keep their order; each may be left out.  tail_i = the optional members i.. (and extras), every one led by a comma
            std::string tail;                                   // rule name of tail_i, built from the back
            if (extras) tail = add(hint + "-extras", "( ws \",\" ws " + extra_kv + " )*");
            std::vector<std::string> tails(opt_kv.size() + 1);
            tails[opt_kv.size()] = tail;
            for (size_t i = opt_kv.size(); i-- > 0;) {
                std::string body = "( ws \",\" ws " + opt_kv[i] + " )?";
                if (!tails[i + 1].empty()) body += " " + tails[i + 1];
                tails[i] = add(hint + "-tail-" + std::to_string(i), body);
            }
            std::string body = "\"{\" ws ";
            if (!req_kv.empty()) {
                for (size_t i = 0; i < req_kv.size(); i++) body += (i ? "ws \",\" ws " : "") + req_kv[i] + " ";
                if (!tails[0].empty()) body += tails[0] + " ";
                body += "ws \"}\"";
            } else {
                // nothing is required: whichever member comes first carries no comma.  first_i = member i then tail_(i+1), or first_(i+1)
                std::string first;
                if (extras) first = add(hint + "-first-x", extra_kv + " " + tail);
                for (size_t i = opt_kv.size(); i-- > 0;) {
                    std::string b = opt_kv[i];
                    if (!tails[i + 1].empty()) b += " " + tails[i + 1];
                    if (!first.empty()) b += " | " + first;
                    first = add(hint + "-first-" + std::to_string(i), b);
                }
                body += "( " + first + " ws )? \"}\"";
            }
            out = add(hint, body);
            return true;
        }
        return fail("unsupported schema type " + type);
    }
};

}  // namespace

bool json_schema_to_gbnf(const Json &schema, std::string &gbnf, std::string &err) {
    Conv c(schema);
    std::string top;
    if (!c.visit(schema, "root", top)) { err = c.err.empty() ? "cannot convert the schema" : c.err; return false; }
    gbnf.clear();
    if (top != "root") gbnf += "root ::= " + top + "\n";
    for (const auto &r : c.rules) gbnf += r.first + " ::= " + r.second + "\n";
    return true;
}

}  // namespace mi355
