// grammar.h — GBNF grammars for constrained sampling.  The reference takes a grammar three ways: the request's `grammar` string
// (src/chat_completion_request.h:160 -> src/llama_server_context.cc:473), the load option `grammar_file` whose text replaces it
// (src/llama_engine.cc:573-585, 812-814), and `response_format` {type: json_object | json_schema} converted with json_schema_to_grammar
// (src/llama_engine.cc:794-801); the constraint itself lives in the llama.cpp submodule (absent from the mount: SURVEY.md §0), so this
// file restates the published GBNF language (llama.cpp grammars/README.md) and its sampling contract:
//   * a candidate token is admissible iff the bytes of its piece can extend the text matched so far (pieces may stop in the middle of a
//     UTF-8 sequence: the code point is then matched when it completes); end-of-generation tokens are admissible iff the grammar can end;
//   * the accepted token advances the match.
// Design: rules are alternatives of sequences over two kinds of symbols, a set of code-point ranges (possibly negated) or a rule
// reference; repetition operators are rewritten into helper rules when the text is parsed.  A match state is a set of parse stacks whose
// tops are always range symbols (a pushdown recogniser run breadth-first); advancing by a code point keeps the stacks whose top admits it.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace mi355 {

struct CharSet {
    std::vector<std::pair<uint32_t, uint32_t>> ranges;   // inclusive
    bool negated = false;
    bool has(uint32_t cp) const {
        bool in = false;
        for (const auto &r : ranges) if (cp >= r.first && cp <= r.second) { in = true; break; }
        return in != negated;
    }
    // could SOME code point of [lo, hi] be in the set? (a piece that ends inside a UTF-8 sequence pins down only a range)
    bool touches(uint32_t lo, uint32_t hi) const;
};

struct GrammarSymbol {
    int rule = -1;          // >= 0: reference to that rule; otherwise `set`
    CharSet set;
};
typedef std::vector<GrammarSymbol> GrammarSeq;
struct GrammarRule {
    std::string name;
    std::vector<GrammarSeq> alts;
};

class Grammar {
  public:
    // parses GBNF text; nullptr + err on a syntax error, an undefined rule, a missing root or left recursion
    static std::shared_ptr<const Grammar> parse(const std::string &text, std::string &err, const std::string &root = "root");
    const std::vector<GrammarRule> &rules() const { return rules_; }
    int root() const { return root_; }

  private:
    std::vector<GrammarRule> rules_;
    int root_ = -1;
};

// incremental UTF-8 decoding across pieces
struct Utf8Tail {
    uint32_t value = 0;
    uint32_t least = 0;      // the smallest code point this sequence length may encode (anything below is an overlong form: refused)
    int remain = 0;          // continuation bytes still missing; -1: the byte stream was not UTF-8
};

class GrammarMatcher {
  public:
    explicit GrammarMatcher(std::shared_ptr<const Grammar> g);
    void reset();
    bool admits(const std::string &piece) const;     // could these bytes come next?
    bool accept(const std::string &piece);           // advance; false (and the state is dead) if they could not
    bool can_end() const;                            // the text so far is a complete sentence of the grammar
    bool dead() const { return stacks_.empty(); }
    size_t n_stacks() const { return stacks_.size(); }

  private:
    struct Frame { int rule, alt, pos; };
    typedef std::vector<Frame> Stack;
    static bool same(const Stack &a, const Stack &b);
    void settle(Stack st, std::vector<Stack> &out) const;                         // run rule references until a range symbol is on top
    void step(const std::vector<Stack> &from, uint32_t cp, std::vector<Stack> &to) const;
    const CharSet *top_set(const Stack &st) const;
    bool run(const std::string &piece, std::vector<Stack> &stacks, Utf8Tail &tail) const;

    std::shared_ptr<const Grammar> g_;
    std::vector<Stack> stacks_;
    Utf8Tail tail_;
};

// response_format -> GBNF.  `schema` is the JSON schema object (null / empty: any JSON object, the reference's json_object mode).  Supported: type
// (also as a list), enum, const, properties / required / additionalProperties, items / prefixItems / minItems / maxItems, minLength / maxLength,
// anyOf / oneOf, allOf of object schemas, local $ref (#/$defs/.., #/definitions/..).  false + err for what cannot be expressed.
class Json;
bool json_schema_to_gbnf(const Json &schema, std::string &gbnf, std::string &err);

}  // namespace mi355
