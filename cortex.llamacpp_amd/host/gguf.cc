#include "gguf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>

namespace mi355 {

namespace {
struct Reader {
    const uint8_t *p, *end;
    bool bad = false;
    template <typename T> T get() {
        T v{};
        if (p + sizeof(T) > end) { bad = true; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (bad || n > (uint64_t)(end - p)) { bad = true; return {}; }
        std::string s(reinterpret_cast<const char *>(p), (size_t)n);
        p += n;
        return s;
    }
};
const int kScalarSize[13] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};

void read_scalar(Reader &r, uint32_t type, GGUFValue &v) {
    switch (type) {
        case GV_U8: v.u = r.get<uint8_t>(); v.f = (double)v.u; break;
        case GV_I8: { int8_t x = r.get<int8_t>(); v.u = (uint64_t)(int64_t)x; v.f = x; break; }
        case GV_U16: v.u = r.get<uint16_t>(); v.f = (double)v.u; break;
        case GV_I16: { int16_t x = r.get<int16_t>(); v.u = (uint64_t)(int64_t)x; v.f = x; break; }
        case GV_U32: v.u = r.get<uint32_t>(); v.f = (double)v.u; break;
        case GV_I32: { int32_t x = r.get<int32_t>(); v.u = (uint64_t)(int64_t)x; v.f = x; break; }
        case GV_F32: { float x = r.get<float>(); v.f = x; v.u = (uint64_t)(int64_t)x; break; }
        case GV_BOOL: v.u = r.get<uint8_t>() != 0; v.f = (double)v.u; break;
        case GV_U64: v.u = r.get<uint64_t>(); v.f = (double)v.u; break;
        case GV_I64: { int64_t x = r.get<int64_t>(); v.u = (uint64_t)x; v.f = (double)x; break; }
        case GV_F64: { double x = r.get<double>(); v.f = x; v.u = (uint64_t)(int64_t)x; break; }
        default: r.bad = true;
    }
}
}  // namespace

size_t ggml_type_row_bytes(int type, int64_t n) {
    int be = 0, bb = 0;
    switch (type) {
        case 0: be = 1; bb = 4; break;      // f32
        case 1: be = 1; bb = 2; break;      // f16
        case 2: be = 32; bb = 18; break;    // q4_0
        case 3: be = 32; bb = 20; break;    // q4_1   (sizes of the types this backend has no kernels for are known too: such a file is
        case 6: be = 32; bb = 22; break;    // q5_0    refused by name at load, not by a size mismatch)
        case 7: be = 32; bb = 24; break;    // q5_1
        case 8: be = 32; bb = 34; break;    // q8_0
        case 10: be = 256; bb = 84; break;  // q2_K
        case 11: be = 256; bb = 110; break; // q3_K
        case 20: be = 32; bb = 18; break;   // iq4_nl
        case 30: be = 1; bb = 2; break;     // bf16
        case 12: be = 256; bb = 144; break; // q4_K
        case 13: be = 256; bb = 176; break; // q5_K
        case 14: be = 256; bb = 210; break; // q6_K
        default: return 0;
    }
    if (n % be) return 0;
    return (size_t)(n / be) * bb;
}

const char *ggml_type_name(int type) {
    switch (type) {
        case 0: return "f32"; case 1: return "f16"; case 2: return "q4_0"; case 3: return "q4_1";
        case 6: return "q5_0"; case 7: return "q5_1"; case 8: return "q8_0"; case 10: return "q2_K";
        case 11: return "q3_K"; case 12: return "q4_K"; case 13: return "q5_K"; case 14: return "q6_K";
        case 15: return "q8_K"; case 30: return "bf16";
    }
    return "unknown";
}

GGUFFile::~GGUFFile() {
    if (map_) munmap(map_, file_size);
    if (fd_ >= 0) close(fd_);
}

std::string GGUFFile::open(const std::string &path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) return "cannot open " + path;
    struct stat st;
    if (fstat(fd_, &st) != 0) return "cannot stat " + path;
    file_size = (size_t)st.st_size;
    if (file_size < 24) return "file too small to be GGUF";
    void *m = mmap(nullptr, file_size, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) return "mmap failed";
    map_ = static_cast<uint8_t *>(m);
    Reader r{map_, map_ + file_size};
    const uint32_t magic = r.get<uint32_t>();
    version = r.get<uint32_t>();
    if (magic != 0x46554747u) return "bad magic (not GGUF)";
    if (version < 2 || version > 3) return "unsupported GGUF version " + std::to_string(version);
    const uint64_t n_tensors = r.get<uint64_t>(), n_kv = r.get<uint64_t>();
    // counts come from the file: nothing is sized from them before they are checked against the bytes that are actually
    // there (a key / value pair takes at least 12 bytes, a tensor record at least 24, a string element at least 8)
    const uint64_t remaining = (uint64_t)(r.end - r.p);
    if (r.bad || n_kv > remaining / 12 || n_tensors > remaining / 24) return "truncated or corrupt GGUF header (counts exceed the file size)";
    for (uint64_t i = 0; i < n_kv && !r.bad; i++) {
        std::string key = r.str();
        GGUFValue v;
        v.type = r.get<uint32_t>();
        if (v.type == GV_STR) {
            v.s = r.str();
        } else if (v.type == GV_ARR) {
            v.elem_type = r.get<uint32_t>();
            v.u = r.get<uint64_t>();
            if (v.elem_type == GV_STR) {
                if (v.u > (uint64_t)(r.end - r.p) / 8) { r.bad = true; break; }
                v.strs.reserve((size_t)v.u);
                for (uint64_t j = 0; j < v.u && !r.bad; j++) v.strs.push_back(r.str());
            } else if (v.elem_type < 13 && kScalarSize[v.elem_type]) {
                v.raw = r.p;
                const uint64_t esz = (uint64_t)kScalarSize[v.elem_type];
                if (v.u > (uint64_t)(r.end - r.p) / esz) r.bad = true; else r.p += v.u * esz;
            } else {
                r.bad = true;
            }
        } else {
            read_scalar(r, v.type, v);
        }
        kv.emplace(std::move(key), std::move(v));
    }
    if (r.bad) return "truncated or corrupt GGUF metadata";
    alignment = get_u("general.alignment", 32);
    if (alignment == 0 || (alignment & (alignment - 1))) return "bad general.alignment";
    tensors.resize((size_t)n_tensors);
    for (auto &t : tensors) {
        t.name = r.str();
        t.n_dims = (int)r.get<uint32_t>();
        if (t.n_dims < 0 || t.n_dims > 4) { r.bad = true; break; }
        for (int d = 0; d < t.n_dims; d++) {
            const uint64_t ne = r.get<uint64_t>();
            if (ne == 0 || ne > (uint64_t)1 << 40) { r.bad = true; break; }     // (ne stays 1 for the unused dimensions)
            t.ne[d] = (int64_t)ne;
        }
        if (r.bad) break;
        t.type = (int)r.get<uint32_t>();
        t.offset = r.get<uint64_t>();
        if (r.bad) break;
    }
    if (r.bad) return "truncated or corrupt GGUF tensor table";
    uint64_t data_off = (uint64_t)(r.p - map_);
    data_off = (data_off + alignment - 1) / alignment * alignment;
    if (data_off > file_size) return "truncated GGUF file (no tensor data)";
    const uint64_t data_bytes = (uint64_t)file_size - data_off;
    for (size_t i = 0; i < tensors.size(); i++) {
        auto &t = tensors[i];
        // rows * row bytes and offset + bytes in arithmetic that cannot wrap (every ne is in [1, 2^40])
        unsigned __int128 rows = (unsigned __int128)(uint64_t)t.ne[1] * (uint64_t)t.ne[2] * (uint64_t)t.ne[3];
        const size_t rb = ggml_type_row_bytes(t.type, t.ne[0]);
        const unsigned __int128 bytes = rows * rb;
        if (rb) {
            if (bytes > data_bytes || t.offset > data_bytes - (uint64_t)bytes) return "tensor " + t.name + " runs past end of file";
            if (t.offset % alignment) return "tensor " + t.name + " is not aligned to general.alignment";
        }
        t.bytes = rb ? (size_t)bytes : 0;
        t.data = rb ? map_ + data_off + t.offset : nullptr;      // (unsupported type: no bytes are ever read; the loader rejects it by name)
        index_[t.name] = i;
    }
    return {};
}

const GGUFValue *GGUFFile::find(const std::string &key) const {
    auto it = kv.find(key);
    return it == kv.end() ? nullptr : &it->second;
}
uint64_t GGUFFile::get_u(const std::string &key, uint64_t def) const {
    const GGUFValue *v = find(key);
    return (v && v->type != GV_STR && v->type != GV_ARR) ? v->u : def;
}
double GGUFFile::get_f(const std::string &key, double def) const {
    const GGUFValue *v = find(key);
    return (v && v->type != GV_STR && v->type != GV_ARR) ? v->f : def;
}
std::string GGUFFile::get_s(const std::string &key, const std::string &def) const {
    const GGUFValue *v = find(key);
    return (v && v->type == GV_STR) ? v->s : def;
}
bool GGUFFile::get_b(const std::string &key, bool def) const {
    const GGUFValue *v = find(key);
    return (v && v->type != GV_STR && v->type != GV_ARR) ? v->u != 0 : def;
}
const GGUFTensorInfo *GGUFFile::tensor(const std::string &name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors[it->second];
}

}  // namespace mi355
