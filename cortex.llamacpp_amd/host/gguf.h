// gguf.h — GGUF v2/v3 container reader (mmap).  Replaces the slice of ggml's gguf.cpp that
// common_init_from_params (reference: src/llama_server_context.cc:207) relies on; format per SURVEY.md §A.4.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace mi355 {

enum GGUFValueType : uint32_t {
    GV_U8 = 0, GV_I8, GV_U16, GV_I16, GV_U32, GV_I32, GV_F32, GV_BOOL, GV_STR, GV_ARR, GV_U64, GV_I64, GV_F64,
};

struct GGUFValue {
    uint32_t type = 0;         // GGUFValueType
    uint32_t elem_type = 0;    // for arrays
    uint64_t u = 0;            // integer payload (sign-extended for signed types), array length for arrays
    double f = 0.0;            // numeric payload as double
    std::string s;             // string payload
    std::vector<std::string> strs;   // string arrays
    const uint8_t *raw = nullptr;    // numeric arrays: pointer into the mapping
};

struct GGUFTensorInfo {
    std::string name;
    int n_dims = 0;
    int64_t ne[4] = {1, 1, 1, 1};
    int type = 0;
    uint64_t offset = 0;       // relative to data section
    const uint8_t *data = nullptr;
    size_t bytes = 0;
};

class GGUFFile {
  public:
    GGUFFile() = default;
    ~GGUFFile();
    GGUFFile(const GGUFFile &) = delete;
    GGUFFile &operator=(const GGUFFile &) = delete;

    // returns empty string on success, else an error message
    std::string open(const std::string &path);

    const GGUFValue *find(const std::string &key) const;
    uint64_t get_u(const std::string &key, uint64_t def) const;
    double get_f(const std::string &key, double def) const;
    std::string get_s(const std::string &key, const std::string &def) const;
    bool get_b(const std::string &key, bool def) const;
    const GGUFTensorInfo *tensor(const std::string &name) const;

    uint32_t version = 0;
    uint64_t alignment = 32;
    std::map<std::string, GGUFValue> kv;
    std::vector<GGUFTensorInfo> tensors;
    size_t file_size = 0;

  private:
    int fd_ = -1;
    uint8_t *map_ = nullptr;
    std::map<std::string, size_t> index_;
};

size_t ggml_type_row_bytes(int type, int64_t n);   // 0 if unsupported
const char *ggml_type_name(int type);

}  // namespace mi355
