// vocab.h — tokenizer read from GGUF metadata (tokenizer.ggml.*).  Stands in for the llama_vocab calls the reference makes
// (SURVEY.md §8b): common_tokenize (src/llama_server_context.cc:395,398,410,536,644,936,992), common_token_to_piece
// (:136,720), llama_vocab_bos/eos/is_eog/n_tokens (:512,517,792), llama_add_bos_token (:238).
//   "llama" model : SentencePiece-style BPE with scores, space prefix U+2581, byte fallback <0xXX>   (Llama-2, TinyLlama, Mixtral)
//   "gpt2"  model : byte-level BPE with merge ranks (Llama-3); the pre-tokenizer is a hand-written splitter for the
//                   llama-bpe pattern (ASCII letter/digit classes, non-ASCII code points treated as letters).
//   "bert"  model : WordPiece as llama.cpp's WPM tokenizer runs it (encoder / embedding models: the reference's embedding smoke model is one, Makefile:6):
//                   lower-case, accents of Latin-1 / Latin Extended-A letters stripped, split on whitespace with every punctuation mark, ASCII symbol and CJK
//                   ideograph a word of its own; each word gets the U+2581 prefix and is cut greedily into the longest vocabulary entries from the left, a word
//                   with an uncovered rest becomes the unknown token; [CLS] ... [SEP] around the text when specials are asked for.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace mi355 {

class GGUFFile;

enum TokenType { TT_UNDEFINED = 0, TT_NORMAL = 1, TT_UNKNOWN = 2, TT_CONTROL = 3, TT_USER_DEFINED = 4, TT_UNUSED = 5, TT_BYTE = 6 };

class Vocab {
  public:
    bool load(const GGUFFile &f, std::string &err);
    // synthetic / test construction
    void init_spm(const std::vector<std::string> &tokens, const std::vector<float> &scores, const std::vector<int> &types,
                  int bos, int eos, int unk, bool add_bos);

    void init_wpm(const std::vector<std::string> &tokens, int cls, int sep, int unk);      // a "bert" vocabulary (tests)

    int n_tokens() const { return (int)tokens_.size(); }
    int bos() const { return bos_; }
    int eos() const { return eos_; }
    int eot() const { return eot_; }
    bool add_bos() const { return add_bos_; }
    bool add_eos() const { return add_eos_; }
    bool is_eog(int id) const { return id >= 0 && (id == eos_ || id == eot_ || eog_extra_.count(id)); }
    bool is_control(int id) const { return id >= 0 && id < n_tokens() && types_[(size_t)id] == TT_CONTROL; }
    bool has_vocab() const { return !tokens_.empty(); }
    const std::string &model() const { return model_; }

    // common_tokenize(vocab, text, add_special, parse_special)
    std::vector<int32_t> tokenize(const std::string &text, bool add_special, bool parse_special = false) const;
    // common_token_to_piece(ctx, token, special = true)
    std::string token_to_piece(int32_t id, bool special = true) const;
    std::string detokenize(const std::vector<int32_t> &ids, bool special = false) const;

  private:
    void build_index();
    void tokenize_spm(const std::string &text, std::vector<int32_t> &out) const;
    void tokenize_bpe(const std::string &text, std::vector<int32_t> &out) const;
    void tokenize_wpm(const std::string &text, std::vector<int32_t> &out) const;
    void bpe_word(const std::string &word, std::vector<int32_t> &out) const;
    int byte_token(uint8_t b) const;

    std::string model_ = "llama";
    std::vector<std::string> tokens_;
    std::vector<float> scores_;
    std::vector<int> types_;
    std::unordered_map<std::string, int> index_;
    std::map<std::pair<std::string, std::string>, int> merge_rank_;
    std::vector<int> special_ids_;             // control / user-defined tokens, longest text first
    std::map<int, int> eog_extra_;
    int bos_ = -1, eos_ = -1, eot_ = -1, unk_ = 0;
    int cls_ = -1, sep_ = -1;                  // "bert" vocabularies: the tokens put around a text
    size_t max_token_len_ = 0;
    bool add_bos_ = true, add_eos_ = false, add_space_prefix_ = true;
};

}  // namespace mi355
