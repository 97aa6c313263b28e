// c_api.cc — the C-ABI declared in include/mi355_llama.h, over host/runtime.{h,cc}.
#include "../../include/mi355_llama.h"

#include <atomic>
#include <memory>
#include <new>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../host/clip.h"
#include "../host/engine.h"
#include "../host/hip_backend.h"
#include "../host/log.h"
#include "../host/runtime.h"
#include "../host/tp_comm.h"
#include "../host/tp_split.h"
#include "../host/vocab.h"

namespace mi355 {
const std::string &last_error_string();
}

using namespace mi355;

struct mi355_model {
    Model *m;
    Vocab vocab;            // loaded on first tokenizer call
    bool vocab_tried = false, vocab_ok = false;
};
typedef void (*mi355_release_fn)(void *);
struct mi355_engine {
    LlamaEngine eng{make_hip_backend};
    std::atomic<mi355_release_fn> release{nullptr};     // mi355_engine_set_release_callback
};
struct mi355_clip { ClipModel m; };
struct mi355_context {
    Context *c;
    std::vector<const char *> prof_names;
};

static thread_local std::string t_err;
static int g_op_mmq_planes = 1, g_op_mmq_ksplit = 1;
static bool g_backend_ok = false;

static void fail(const std::string &s) { t_err = s; }

// No C++ exception may cross the C ABI (the reference's own rule for its plugin boundary, enginei.h): allocation failures
// and anything else thrown below an entry point become that entry point's error return and a last_error message.
#define MI355_GUARD(on_error, ...)                                              \
    try { __VA_ARGS__ }                                                         \
    catch (const std::bad_alloc &) { fail("out of host memory"); on_error; }    \
    catch (const std::exception &ex) { fail(std::string("internal error: ") + ex.what()); on_error; } \
    catch (...) { fail("internal error"); on_error; }

extern "C" {

int mi355_backend_init(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        fail("no HIP device visible: this backend has no CPU fallback");
        g_backend_ok = false;
        return MI355_ERR_NO_DEVICE;
    }
    g_backend_ok = true;
    return MI355_OK;
}
void mi355_backend_free(void) { g_backend_ok = false; }
int mi355_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
const char *mi355_last_error(void) {
    if (t_err.empty() && !last_error_string().empty()) return last_error_string().c_str();
    return t_err.c_str();
}
int64_t mi355_time_us(void) {
    using namespace std::chrono;
    return duration_cast<microseconds>(steady_clock::now().time_since_epoch()).count();
}
const char *mi355_print_system_info(void) {
    static std::string s;
    int n = mi355_device_count();
    s = "mi355-llama | HIP devices = " + std::to_string(n);
    if (n > 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, 0) == hipSuccess)
            s += std::string(" | ") + p.name + " | " + p.gcnArchName + " | CUs = " + std::to_string(p.multiProcessorCount) +
                 " | HBM = " + std::to_string(p.totalGlobalMem >> 30) + " GiB";
    }
    return s.c_str();
}

mi355_model_params mi355_model_default_params(void) {
    mi355_model_params p{};
    p.n_gpu_layers = 300; p.main_gpu = 0; p.use_mmap = 1; p.use_mlock = 0; p.tp_rank = 0; p.tp_size = 1; p.prefill_planes = -1;
    return p;
}

mi355_model *mi355_model_load_from_file(const char *path, mi355_model_params params) {
    if (!path) { fail("null path"); return nullptr; }
    if (!g_backend_ok && mi355_backend_init() != MI355_OK) return nullptr;
    if (params.n_gpu_layers <= 0) { fail("ngl=0 requested: this backend is device-only (no CPU path)"); return nullptr; }
    if (params.tp_size > 1 && (tp_size() != params.tp_size || tp_rank() != params.tp_rank)) {
        fail("tp_size > 1 needs the process's row-split group first (mi355_tp_init with the same rank / size)");
        return nullptr;
    }
    MI355_GUARD(return nullptr,
        std::string err;
        int status = 0;
        Model *m = model_load(path, params.main_gpu, err, status, params.prefill_planes, params.tp_rank, params.tp_size);
        if (!m) { fail(err); return nullptr; }
        mi355_model *h = new (std::nothrow) mi355_model;
        if (!h) { delete m; fail("out of host memory"); return nullptr; }
        h->m = m;
        return h;
    )
}
void mi355_model_free(mi355_model *model) {
    if (!model) return;
    delete model->m;
    delete model;
}
int32_t mi355_model_n_vocab(const mi355_model *m) { return m->m->hp.n_vocab; }
int32_t mi355_model_n_embd(const mi355_model *m) { return m->m->hp.n_embd; }
int32_t mi355_model_n_layer(const mi355_model *m) { return m->m->hp.n_layer; }
int32_t mi355_model_n_head(const mi355_model *m) { return m->m->hp.n_head; }
int32_t mi355_model_n_head_kv(const mi355_model *m) { return m->m->hp.n_head_kv; }
int32_t mi355_model_n_ctx_train(const mi355_model *m) { return m->m->hp.n_ctx_train; }
uint64_t mi355_model_size(const mi355_model *m) { return m->m->file_tensor_bytes; }
uint64_t mi355_model_cpu_buffer(const mi355_model *m) { return m->m->host_bytes; }
uint64_t mi355_model_other_buffer(const mi355_model *m) { return m->m->device_bytes; }
uint64_t mi355_model_bytes_per_token(const mi355_model *m) { return m->m->bytes_per_token; }
uint64_t mi355_model_planes_bytes(const mi355_model *m) { return m->m->planes_bytes; }
const char *mi355_model_desc(const mi355_model *m) { return m->m->desc.c_str(); }
int mi355_model_meta_str(const mi355_model *m, const char *key, char *buf, size_t buf_size) {
    if (!m || !key || !buf || !buf_size) return 0;
    const GGUFValue *v = m->m->file->find(key);
    if (!v || v->type == GV_ARR) return 0;
    std::string s;
    if (v->type == GV_STR) s = v->s;
    else if (v->type == GV_F32 || v->type == GV_F64) { char t[64]; snprintf(t, sizeof t, "%g", v->f); s = t; }
    else if (v->type == GV_BOOL) s = v->u ? "true" : "false";
    else s = std::to_string((long long)(int64_t)v->u);
    snprintf(buf, buf_size, "%s", s.c_str());
    return 1;
}

mi355_context_params mi355_context_default_params(void) {
    mi355_context_params p{};
    p.n_ctx = 2048; p.n_batch = 2048; p.n_ubatch = 512; p.n_seq_max = 1;
    p.type_k = MI355_TYPE_F16; p.type_v = MI355_TYPE_F16; p.flash_attn = 1; p.embeddings = 0; p.use_graphs = 1; p.logits_to_host = 1;
    return p;
}

mi355_context *mi355_context_new(mi355_model *model, mi355_context_params params) {
    if (!model) { fail("null model"); return nullptr; }
    ContextParams cp;
    cp.n_ctx = params.n_ctx; cp.n_batch = params.n_batch; cp.n_ubatch = params.n_ubatch ? params.n_ubatch : params.n_batch;
    cp.n_seq_max = params.n_seq_max ? params.n_seq_max : 1;
    cp.type_k = params.type_k; cp.type_v = params.type_v;
    cp.flash_attn = params.flash_attn != 0 || params.type_k != MI355_TYPE_F16 || params.type_v != MI355_TYPE_F16;
    cp.embeddings = params.embeddings != 0;
    cp.use_graphs = params.use_graphs != 0;
    cp.logits_to_host = params.logits_to_host != 0;
    if (cp.n_ctx == 0 || cp.n_batch == 0 || cp.n_seq_max > 64) { fail("bad context parameters (n_ctx and n_batch must be positive, n_seq_max at most 64)"); return nullptr; }
    MI355_GUARD(return nullptr,
        std::unique_ptr<Context> c(new Context(model->m, cp));
        std::string err;
        if (!c->init(err)) { fail(err); return nullptr; }
        mi355_context *h = new mi355_context{c.get(), {}};
        c.release();
        return h;
    )
}
void mi355_context_free(mi355_context *ctx) {
    if (!ctx) return;
    delete ctx->c;
    delete ctx;
}
uint32_t mi355_n_ctx(const mi355_context *ctx) { return ctx->c->cp.n_ctx; }
uint32_t mi355_n_batch(const mi355_context *ctx) { return ctx->c->cp.n_batch; }
uint32_t mi355_n_ubatch(const mi355_context *ctx) { return ctx->c->cp.n_ubatch; }
uint64_t mi355_context_device_bytes(const mi355_context *ctx) { return ctx->c->device_bytes; }

mi355_batch mi355_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max) {
    mi355_batch b{};
    // llama_batch_init: embd != 0 allocates n_tokens x embd floats and NO token array
    if (embd > 0) b.embd = (float *)calloc((size_t)n_tokens * (size_t)embd, sizeof(float));
    else b.token = (mi355_token *)calloc((size_t)n_tokens, sizeof(mi355_token));
    b.pos = (mi355_pos *)calloc((size_t)n_tokens, sizeof(mi355_pos));
    b.n_seq_id = (int32_t *)calloc((size_t)n_tokens, sizeof(int32_t));
    b.seq_id = (mi355_seq_id **)calloc((size_t)n_tokens + 1, sizeof(mi355_seq_id *));
    for (int i = 0; i < n_tokens; i++) b.seq_id[i] = (mi355_seq_id *)calloc((size_t)(n_seq_max > 0 ? n_seq_max : 1), sizeof(mi355_seq_id));
    b.seq_id[n_tokens] = nullptr;
    b.logits = (int8_t *)calloc((size_t)n_tokens, 1);
    return b;
}
void mi355_batch_free(mi355_batch b) {
    free(b.token); free(b.embd); free(b.pos); free(b.n_seq_id); free(b.logits);
    if (b.seq_id) {
        for (int i = 0; b.seq_id[i]; i++) free(b.seq_id[i]);
        free(b.seq_id);
    }
}

int32_t mi355_decode(mi355_context *ctx, mi355_batch batch) {
    if (!ctx) return MI355_ERR_ARG;
    MI355_GUARD(return MI355_ERR_ARG,
        // (llama_batch: token ids, or - image embeddings - rows of n_embd floats in embd with token == NULL)
        const int rc = ctx->c->decode(batch.n_tokens, batch.embd ? nullptr : batch.token, batch.pos, batch.n_seq_id, batch.seq_id, batch.logits, batch.embd);
        if (rc < 0) fail(ctx->c->last_error);
        return rc;
    )
}
// n greedy single-token steps in one call: llama_decode, the logits row made host-visible (llama_get_logits_ith), the arg-max fed back - the inner loop of a
// generation as the reference's C++ slot loop runs it, without a scripting caller's per-call overhead in every step
int32_t mi355_greedy_steps(mi355_context *ctx, mi355_token first, mi355_pos pos0, mi355_seq_id seq, int32_t n, mi355_token *out_tokens) {
    if (!ctx || n < 0) return MI355_ERR_ARG;
    MI355_GUARD(return MI355_ERR_ARG,
        int32_t tok = first, one = 1, sid = seq;
        int32_t *sp = &sid;
        const int8_t flag = 1;
        for (int32_t i = 0; i < n; i++) {
            int32_t pos = pos0 + i;
            const int rc = ctx->c->decode(1, &tok, &pos, &one, &sp, &flag);
            if (rc != 0) { if (rc < 0) fail(ctx->c->last_error); return i; }
            if (!ctx->c->logits_ith(0)) { fail("no logits row"); return i; }
            tok = ctx->c->argmax_ith(0);
            if (tok < 0) { fail(ctx->c->last_error.empty() ? "arg-max failed" : ctx->c->last_error); return i; }
            if (out_tokens) out_tokens[i] = tok;
        }
        return n;
    )
}
float *mi355_get_logits_ith(mi355_context *ctx, int32_t i) {
    float *p = ctx->c->logits_ith(i);
    if (!p) fail(ctx->c->last_error.empty() ? "no logits for that batch row" : ctx->c->last_error);      // (why: a step an in-kernel wait gave up on says so here)
    return p;
}
int32_t mi355_get_argmax_ith(mi355_context *ctx, int32_t i) { return ctx->c->argmax_ith(i); }
int32_t mi355_get_topk_ith(mi355_context *ctx, int32_t i, int32_t k, int32_t n_adj, const int32_t *adj_tok, const float *adj_bias, const int32_t *adj_count,
                           float penalty_repeat, float penalty_freq, float penalty_present, int32_t *toks_out, float *logits_out) {
    if (!ctx || !toks_out || !logits_out || n_adj < 0 || n_adj > TOPK_MAX_ADJ || (n_adj > 0 && (!adj_tok || !adj_bias || !adj_count))) { fail("bad arguments"); return -1; }
    TopkAdj a{};
    a.n = n_adj; a.repeat = penalty_repeat; a.freq = penalty_freq; a.present = penalty_present;
    for (int j = 0; j < n_adj; j++) { a.tok[j] = adj_tok[j]; a.bias[j] = adj_bias[j]; a.cnt[j] = adj_count[j]; }
    const int r = ctx->c->topk_ith(i, k, a, toks_out, logits_out);
    if (r < 0) fail(ctx->c->last_error.empty() ? "top-k failed" : ctx->c->last_error);
    return r;
}
int64_t mi355_debug_mega_steps(const mi355_context *ctx) { return ctx->c->mega_steps; }
int64_t mi355_debug_engine_steps(const mi355_context *ctx) { return ctx->c->engine_steps; }
int64_t mi355_debug_fused_skipped_steps(const mi355_context *ctx) { return ctx->c->fused_skipped_steps; }
int64_t mi355_debug_qkv_attn_launches(const mi355_context *ctx) { return ctx->c->qkv_attn_launches; }
int64_t mi355_debug_qkv_attn_plan(int32_t type_q, int32_t type_k, int32_t type_v, int32_t type_o, int32_t n_embd, int32_t n_head, int32_t n_head_kv, int32_t head_dim,
                                  int32_t type_kv, int32_t n_kv, int32_t *slots_out) {
    int slots = 0;
    const size_t lds = qkv_attn_out_plan_lds(type_q, type_k, type_v, type_o, n_embd, n_head, n_head_kv, head_dim, type_kv, n_kv, &slots);
    if (slots_out) *slots_out = slots;
    return (int64_t)lds;
}
void mi355_set_embeddings(mi355_context *ctx, int32_t enabled) { ctx->c->embeddings_enabled = enabled != 0 || ctx->c->model->hp.encoder; }
float *mi355_get_embeddings_ith(mi355_context *ctx, int32_t i) { return ctx->c->embeddings_ith(i); }
void mi355_synchronize(mi355_context *ctx) { ctx->c->synchronize(); }

// ---- LLaVA image path (host/clip.h)
mi355_clip *mi355_clip_model_load(const char *path, int32_t main_gpu) {
    if (!path) { fail("clip_model_load: no path"); return nullptr; }
    MI355_GUARD(return nullptr,
        std::unique_ptr<mi355_clip> c(new mi355_clip);
        const std::string err = c->m.load(path, main_gpu < 0 ? 0 : main_gpu);
        if (!err.empty()) { fail(err); return nullptr; }
        return c.release();
    )
}
void mi355_clip_free(mi355_clip *clip) { delete clip; }
int32_t mi355_clip_n_mmproj_embd(const mi355_clip *clip) { return clip ? clip->m.proj_dim : 0; }
int32_t mi355_clip_n_patches(const mi355_clip *clip) { return clip ? clip->m.n_patches() : 0; }
int32_t mi355_clip_image_size(const mi355_clip *clip) { return clip ? clip->m.image_size : 0; }
int32_t mi355_clip_max_image_rows(const mi355_clip *clip) { return clip ? clip->m.max_image_rows() : 0; }
int32_t mi355_clip_image_preprocess_grid(const mi355_clip *clip, const uint8_t *rgb, int32_t nx, int32_t ny, float *out, size_t out_floats, int32_t *grid_w, int32_t *grid_h) {
    if (!clip || !rgb || !out || nx <= 0 || ny <= 0) { fail("clip_image_preprocess_grid: bad arguments"); return MI355_ERR_ARG; }
    MI355_GUARD(return MI355_ERR_ARG,
        ClipImageU8 img;
        img.nx = nx; img.ny = ny; img.rgb.assign(rgb, rgb + (size_t)3 * nx * ny);
        std::vector<std::vector<float>> imgs;
        int gw = 0, gh = 0;
        clip->m.preprocess_all(img, imgs, gw, gh);
        const size_t per = (size_t)3 * clip->m.image_size * clip->m.image_size;
        if (out_floats < per * imgs.size()) { fail("clip_image_preprocess_grid: output buffer too small"); return MI355_ERR_ARG; }
        for (size_t i = 0; i < imgs.size(); i++) memcpy(out + i * per, imgs[i].data(), per * sizeof(float));
        if (grid_w) *grid_w = gw;
        if (grid_h) *grid_h = gh;
        return (int32_t)imgs.size();
    )
}
int32_t mi355_clip_image_load_from_bytes(const uint8_t *bytes, size_t n_bytes, int32_t *nx, int32_t *ny, uint8_t *rgb_out, size_t rgb_cap) {
    MI355_GUARD(return MI355_ERR_ARG,
        ClipImageU8 img;
        const std::string err = clip_image_load_from_bytes(bytes, n_bytes, img);
        if (!err.empty()) { fail(err); return MI355_ERR_ARG; }
        if (nx) *nx = img.nx;
        if (ny) *ny = img.ny;
        if (rgb_out) {
            if (rgb_cap < img.rgb.size()) { fail("clip_image_load_from_bytes: rgb_cap too small"); return MI355_ERR_ARG; }
            memcpy(rgb_out, img.rgb.data(), img.rgb.size());
        }
        return MI355_OK;
    )
}
int32_t mi355_clip_image_preprocess(const mi355_clip *clip, const uint8_t *rgb, int32_t nx, int32_t ny, float *out) {
    if (!clip || !rgb || !out || nx <= 0 || ny <= 0) { fail("clip_image_preprocess: bad arguments"); return MI355_ERR_ARG; }
    MI355_GUARD(return MI355_ERR_ARG,
        ClipImageU8 img;
        img.nx = nx; img.ny = ny; img.rgb.assign(rgb, rgb + (size_t)3 * nx * ny);
        std::vector<float> f;
        clip->m.preprocess(img, f);
        memcpy(out, f.data(), f.size() * sizeof(float));
        return MI355_OK;
    )
}
int32_t mi355_clip_image_encode(mi355_clip *clip, const float *img, float *out) {
    if (!clip || !img || !out) { fail("clip_image_encode: bad arguments"); return MI355_ERR_ARG; }
    MI355_GUARD(return MI355_ERR_ARG,
        const std::string err = clip->m.encode(img, out);
        if (!err.empty()) { fail(err); return MI355_ERR_ARG; }
        return MI355_OK;
    )
}
int32_t mi355_llava_image_embed_from_bytes(mi355_clip *clip, const uint8_t *bytes, size_t n_bytes, float *out, size_t out_floats) {
    if (!clip || !bytes || !out) { fail("llava_image_embed: bad arguments"); return MI355_ERR_ARG; }
    MI355_GUARD(return MI355_ERR_ARG,
        ClipImageU8 img;
        std::string err = clip_image_load_from_bytes(bytes, n_bytes, img);
        if (!err.empty()) { fail(err); return MI355_ERR_ARG; }
        std::vector<float> rows;
        int n_rows = 0;
        err = clip->m.embed(img, rows, n_rows);
        if (!err.empty()) { fail(err); return MI355_ERR_ARG; }
        if (out_floats < rows.size()) { fail("llava_image_embed: output buffer too small (" + std::to_string(n_rows) + " rows; size it with mi355_clip_max_image_rows)"); return MI355_ERR_ARG; }
        memcpy(out, rows.data(), rows.size() * sizeof(float));
        return n_rows;
    )
}

void mi355_kv_cache_clear(mi355_context *ctx) { ctx->c->kv_clear(); }
// sequence ids index a 64-bit mask per cell: anything outside [0, 64) is refused here (seq < 0 = "every sequence" only
// where upstream defines it: seq_rm)
static bool seq_in_range(const mi355_context *, mi355_seq_id seq) { return seq >= 0 && seq < 64; }
int32_t mi355_kv_cache_seq_rm(mi355_context *ctx, mi355_seq_id seq, mi355_pos p0, mi355_pos p1) {
    if (!ctx) return 0;
    if (seq >= 0 && !seq_in_range(ctx, seq)) { fail("seq_rm: sequence id out of range"); return 0; }
    return ctx->c->kv_seq_rm(seq, p0, p1) ? 1 : 0;
}
void mi355_kv_cache_seq_cp(mi355_context *ctx, mi355_seq_id src, mi355_seq_id dst, mi355_pos p0, mi355_pos p1) {
    if (!ctx) return;
    if (!seq_in_range(ctx, src) || !seq_in_range(ctx, dst)) { fail("seq_cp: sequence id out of range"); return; }
    ctx->c->kv_seq_cp(src, dst, p0, p1);
}
void mi355_kv_cache_seq_add(mi355_context *ctx, mi355_seq_id seq, mi355_pos p0, mi355_pos p1, mi355_pos delta) {
    if (!ctx) return;
    if (!seq_in_range(ctx, seq)) { fail("seq_add: sequence id out of range"); return; }
    ctx->c->kv_seq_add(seq, p0, p1, delta);
}
int32_t mi355_kv_cache_used_cells(const mi355_context *ctx) { return ctx->c->kv_used_cells(); }

void mi355_debug_enable_taps(mi355_context *ctx, int32_t enabled) { ctx->c->set_debug_taps(enabled != 0); }
int32_t mi355_debug_layer_out(mi355_context *ctx, int32_t il, float *dst, size_t dst_floats) { return ctx->c->debug_layer_out(il, dst, dst_floats); }

void mi355_profile_enable(mi355_context *ctx, int32_t enabled) { ctx->c->set_profile(enabled != 0); }
int32_t mi355_profile_last_decode(mi355_context *ctx, const char **names, float *us, int32_t cap) {
    const auto &p = ctx->c->last_profile();
    int n = 0;
    for (const auto &e : p) {
        if (n >= cap) break;
        names[n] = e.name.c_str();
        us[n] = e.us;
        n++;
    }
    return n;
}
int mi355_debug_force_moe_ids(mi355_context *ctx, const int32_t *ids, int32_t n_layer, int32_t n_tokens, int32_t k) {
    if (!ctx || !ctx->c) return MI355_ERR_ARG;
    if (ctx->c->force_moe_ids(ids, n_layer, n_tokens, k) != 0) { fail(ctx->c->last_error.empty() ? "force_moe_ids failed" : ctx->c->last_error); return MI355_ERR_ARG; }
    return MI355_OK;
}
double mi355_bench_weight_sweep(mi355_context *ctx, int iters, uint64_t *bytes_per_sweep) { return ctx->c->bench_weight_sweep(iters, bytes_per_sweep); }
double mi355_bench_weight_sweep2(mi355_context *ctx, int iters, uint64_t *bytes_per_sweep, int32_t *launches_per_sweep) {
    int n = 0;
    const double us = ctx->c->bench_weight_sweep(iters, bytes_per_sweep, &n);
    if (launches_per_sweep) *launches_per_sweep = n;
    return us;
}

}  // extern "C"

// ------------------------------------------------------------------ per-op wrappers
namespace {
struct DevBuf {
    void *p = nullptr;
    size_t n = 0;
    explicit DevBuf(size_t bytes) : n(bytes) { if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) p = nullptr; else (void)hipMemset(p, 0, bytes ? bytes : 16); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
    bool up(const void *src, size_t bytes) { return p && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess; }
    bool down(void *dst, size_t bytes) { return p && hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost) == hipSuccess; }
};
bool need_device() {
    if (g_backend_ok) return true;
    return mi355_backend_init() == MI355_OK;
}
struct ActBufs {
    DevBuf qs, d, bs, qs0, d0;
    ActQuant q;
    ActBufs(size_t K, size_t T) : qs(T * K), d(T * (K / 256 + 1) * 4), bs(T * (K / 16 + 1) * 2), qs0(T * K), d0(T * (K / 32 + 1) * 2) {
        q.qs = qs.as<int8_t>(); q.d = d.as<float>(); q.bsums = bs.as<int16_t>(); q.qs0 = qs0.as<int8_t>(); q.d0 = d0.as<uint16_t>();
    }
    bool ok() const { return qs.p && d.p && bs.p && qs0.p && d0.p; }
};
int hip_fail(hipError_t e, const char *what) {
    fail(std::string(what) + ": " + hipGetErrorString(e));
    return MI355_ERR_HIP;
}
}  // namespace

extern "C" double hbm_read_probe(size_t bytes, int iters);

extern "C" {

int mi355_op_quantize_act(int32_t act_type, const float *x, int64_t n, int64_t rows, void *out_blocks) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if ((act_type != MI355_TYPE_Q8_K && act_type != MI355_TYPE_Q8_0) || n % 256) { fail("bad args"); return MI355_ERR_ARG; }
    DevBuf dx((size_t)n * rows * 4);
    ActBufs ab((size_t)n, (size_t)rows);
    const size_t ob = act_type == MI355_TYPE_Q8_K ? (size_t)(n / 256) * 292 * rows : (size_t)(n / 32) * 34 * rows;
    DevBuf dout(ob);
    if (!dx.up(x, (size_t)n * rows * 4) || !ab.ok() || !dout.p) { fail("device alloc/copy failed"); return MI355_ERR_OOM; }
    hipError_t e = launch_quantize(dx.as<float>(), (int)n, (int)rows, ab.q, act_type == MI355_TYPE_Q8_K, act_type == MI355_TYPE_Q8_0, nullptr);
    if (e == hipSuccess) e = act_type == MI355_TYPE_Q8_K ? launch_pack_q8k_blocks(ab.q, (int)n, (int)rows, dout.as<uint8_t>(), nullptr)
                                                         : launch_pack_q80_blocks(ab.q, (int)n, (int)rows, dout.as<uint8_t>(), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "quantize_act");
    return dout.down(out_blocks, ob) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_ffn_gate_up(int32_t type, const void *Wg, const void *Wu, int64_t N, int64_t K, const float *x, int64_t T, float *y) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t grow = ggml_row_bytes(type, K), drow = dev_row_bytes(type, K);
    const size_t pb = mmq_planes_bytes(type, N, (int)K);
    if (!grow || K % 256 || !pb || N % 32) { fail("bad type / K / N"); return MI355_ERR_ARG; }
    if (!mmq_planes_swiglu_ok(type, type, (int)N, (int)K, (int)T)) { fail("shape does not take the SwiGLU launch (set mmq_tiles = 4 to force it)"); return MI355_ERR_ARG; }
    DevBuf wsrc(grow * N), wdev(drow * N), pg(pb), pu(pb), dx((size_t)K * T * 4), dy((size_t)N * T * 4);
    ActBufs ab((size_t)K, (size_t)T);
    if (!wsrc.p || !wdev.p || !pg.p || !pu.p || !dx.up(x, (size_t)K * T * 4) || !dy.p || !ab.ok()) { fail("device alloc/copy failed"); return MI355_ERR_OOM; }
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        if (!wsrc.up(i ? Wu : Wg, grow * N)) { fail("device copy failed"); return MI355_ERR_OOM; }
        e = launch_repack_rows(type, wsrc.as<uint8_t>(), wdev.as<uint8_t>(), K, N, nullptr);
        if (e == hipSuccess) e = launch_mmq_expand(type, wdev.as<uint8_t>(), drow, (int)N, (int)K, (i ? pu : pg).as<uint8_t>(), nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e == hipSuccess) e = launch_quantize(dx.as<float>(), (int)K, (int)T, ab.q, true, false, nullptr);
    if (e == hipSuccess) e = launch_mmq_planes_swiglu(type, pg.as<uint8_t>(), pu.as<uint8_t>(), (int)N, (int)K, (int)T, ab.q, dy.as<float>(), (int)N, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "ffn_gate_up");
    return dy.down(y, (size_t)N * T * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_mul_mat(int32_t type, const void *W, int64_t N, int64_t K, const float *x, int64_t T, float *y, int32_t *isum, int32_t *msum) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t grow = ggml_row_bytes(type, K), drow = dev_row_bytes(type, K);
    if (!grow || K % 256) { fail("bad type / K"); return MI355_ERR_ARG; }
    DevBuf wsrc(grow * N), wdev(drow * N), dx((size_t)K * T * 4), dy((size_t)N * T * 4);
    ActBufs ab((size_t)K, (size_t)T);
    if (!wsrc.up(W, grow * N) || !wdev.p || !dx.up(x, (size_t)K * T * 4) || !dy.p || !ab.ok()) { fail("device alloc/copy failed"); return MI355_ERR_OOM; }
    hipError_t e = launch_repack_rows(type, wsrc.as<uint8_t>(), wdev.as<uint8_t>(), K, N, nullptr);
    if (e != hipSuccess) return hip_fail(e, "repack");
    const bool quant = type == T_Q4_K || type == T_Q5_K || type == T_Q6_K || type == T_Q8_0 || type == T_Q2_K || type == T_Q3_K || type == T_Q4_0 || type == T_Q5_0 || type == T_IQ4_NL;
    if (quant) {
        e = launch_quantize(dx.as<float>(), (int)K, (int)T, ab.q, !act_is_q80(type), act_is_q80(type), nullptr);
        if (e != hipSuccess) return hip_fail(e, "quantize");
        if (mmq_q80_applicable(type, (int)K, (int)T)) {
            e = launch_mmq_q80(wdev.as<uint8_t>(), drow, (int)N, (int)K, (int)T, ab.q, dy.as<float>(), (int)N, nullptr, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_fail(e, "mmq_q80");
        } else
        if (mmq_q80_copy_bytes(type, N, (int)K) && mmq_q80_applicable(T_Q8_0, (int)K, (int)T) && g_op_mmq_planes) {   // Q4_0 / Q5_0 / IQ4_NL prompt batches: exact Q8_0-layout copy
            DevBuf cp(mmq_q80_copy_bytes(type, N, (int)K));
            if (!cp.p) return MI355_ERR_OOM;
            e = launch_expand_q80_copy(type, wdev.as<uint8_t>(), drow, (int)N, (int)K, cp.as<uint8_t>(), nullptr);
            if (e == hipSuccess) e = launch_mmq_q80(cp.as<uint8_t>(), dev_row_bytes(T_Q8_0, K), (int)N, (int)K, (int)T, ab.q, dy.as<float>(), (int)N, nullptr, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_fail(e, "mmq_q80 (copy)");
        } else
        if (g_op_mmq_ksplit && mmq_ksplit_applicable(type, (int)K, (int)T)) {
            DevBuf bh(mmq_prep_bytes((int)K, (int)T)), bl(mmq_prep_bytes((int)K, (int)T));
            if (!bh.p || !bl.p) return MI355_ERR_OOM;
            e = launch_mmq_prep(ab.q, (int)K, (int)T, bh.as<int8_t>(), bl.as<int8_t>(), nullptr);
            if (e == hipSuccess) e = launch_mmq_ksplit(type, wdev.as<uint8_t>(), drow, (int)N, (int)K, (int)T, ab.q, bh.as<int8_t>(), bl.as<int8_t>(), dy.as<float>(), (int)N, nullptr, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_fail(e, "mmq_ksplit");
        } else
        if (mmq_applicable(type, (int)K, (int)T) || ((type == T_Q2_K || type == T_Q3_K) && T >= 32 && g_op_mmq_planes)) {   // (Q2_K / Q3_K: planes only)
            DevBuf bh(mmq_prep_bytes((int)K, (int)T)), bl(mmq_prep_bytes((int)K, (int)T));
            if (!bh.p || !bl.p) return MI355_ERR_OOM;
            e = launch_mmq_prep(ab.q, (int)K, (int)T, bh.as<int8_t>(), bl.as<int8_t>(), nullptr);
            if (g_op_mmq_planes) {
                DevBuf pl(mmq_planes_bytes(type, N, (int)K));
                if (!pl.p) return MI355_ERR_OOM;
                if (e == hipSuccess) e = launch_mmq_expand(type, wdev.as<uint8_t>(), drow, (int)N, (int)K, pl.as<uint8_t>(), nullptr);
                // K-split partial sums: only tensors with few rows split (a large N x T never does, and gets no workspace)
                const size_t ws_bytes = (size_t)4 * T * N * sizeof(float) <= ((size_t)256 << 20) ? (size_t)4 * T * N * sizeof(float) : 0;
                DevBuf wsb(ws_bytes ? ws_bytes : 16);
                if (!wsb.p) return MI355_ERR_OOM;
                MMQWorkspace wsp; wsp.p = ws_bytes ? wsb.as<float>() : nullptr; wsp.bytes = ws_bytes;
                if (e == hipSuccess) e = launch_mmq_planes(type, pl.as<uint8_t>(), (int)N, (int)K, (int)T, ab.q, bh.as<int8_t>(), bl.as<int8_t>(), dy.as<float>(), (int)N, nullptr, nullptr, wsp);
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            } else
            if (e == hipSuccess) e = launch_mmq(type, wdev.as<uint8_t>(), drow, (int)N, (int)K, (int)T, ab.q, bh.as<int8_t>(), bl.as<int8_t>(), dy.as<float>(), (int)N, nullptr, nullptr);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e != hipSuccess) return hip_fail(e, "mmq");
        } else
        for (int64_t t0 = 0; t0 < T;) {
            const int64_t rem = T - t0;
            const int nt = rem >= 16 ? 16 : rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1;
            MMVQArgs a{};
            a.n_seg = 1; a.K = (int)K; a.T = nt; a.epi = EPI_STORE;
            a.seg[0].W = wdev.as<uint8_t>(); a.seg[0].out = dy.as<float>() + t0 * N; a.seg[0].type = type; a.seg[0].n_rows = (int)N;
            a.seg[0].ld_out = (int)N; a.seg[0].row_bytes = drow;
            a.aq = ab.q.qs + t0 * K; a.ad = ab.q.d + t0 * (K / 256); a.abs = ab.q.bsums + t0 * (K / 16);
            a.aq0 = ab.q.qs0 + t0 * K; a.ad0 = ab.q.d0 + t0 * (K / 32);
            e = launch_mmvq(a, nullptr);
            if (e != hipSuccess) return hip_fail(e, "mmvq");
            t0 += nt;
        }
        if (isum && msum) {
            const int64_t nblk = act_is_q80(type) ? K / 32 : K / 256;
            DevBuf di((size_t)N * nblk * 4), dm((size_t)N * nblk * 4);
            for (int64_t t = 0; t < T; t++) {
                MMVQArgs a{};
                a.n_seg = 1; a.K = (int)K; a.T = 1; a.epi = EPI_STORE;
                a.seg[0].W = wdev.as<uint8_t>(); a.seg[0].type = type; a.seg[0].n_rows = (int)N; a.seg[0].row_bytes = drow;
                a.aq = ab.q.qs + t * K; a.ad = ab.q.d + t * (K / 256); a.abs = ab.q.bsums + t * (K / 16);
                a.aq0 = ab.q.qs0 + t * K; a.ad0 = ab.q.d0 + t * (K / 32);
                e = launch_mmvq_ints(a, di.as<int32_t>(), dm.as<int32_t>(), nullptr);
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (e != hipSuccess) return hip_fail(e, "mmvq_ints");
                di.down(isum + t * N * nblk, (size_t)N * nblk * 4);
                dm.down(msum + t * N * nblk, (size_t)N * nblk * 4);
            }
        }
    } else {
        // (the model path's choice: batches against f16 tensors on the matrix cores, everything else one wave per row and token)
        if (mmf16_applicable(type, (int)N, (int)K, (int)T, wdev.p, dx.p, dy.p) && (N & 3) == 0)
            e = launch_mmf16(wdev.as<uint8_t>(), (int)N, (int)K, dx.as<float>(), (int)T, dy.as<float>(), (int)N, nullptr, nullptr);
        else
            e = launch_mmv_float(type, wdev.as<uint8_t>(), (int)N, (int)K, dx.as<float>(), (int)T, dy.as<float>(), (int)N, nullptr, nullptr);
        if (e != hipSuccess) return hip_fail(e, "mmv_float");
    }
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "mul_mat");
    return dy.down(y, (size_t)N * T * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_rms_norm_mul(const float *x, const float *w, int64_t n, int64_t T, float eps, float *y) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if (n % 256) { fail("n must be a multiple of 256"); return MI355_ERR_ARG; }
    DevBuf dx((size_t)n * T * 4), dw((size_t)n * 4), dy((size_t)n * T * 4);
    if (!dx.up(x, (size_t)n * T * 4) || !dw.up(w, (size_t)n * 4) || !dy.p) return MI355_ERR_OOM;
    hipError_t e = launch_rmsnorm_quant(dx.as<float>(), dw.as<float>(), (int)n, (int)T, eps, dy.as<float>(), nullptr, false, false, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "rms_norm");
    return dy.down(y, (size_t)n * T * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_rope(float *x, int32_t n_head, int32_t head_dim, int32_t n_rot, const int32_t *pos, int64_t T,
                  float freq_base, float freq_scale, const float *freq_factors, int32_t neox) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t nb = (size_t)T * n_head * head_dim * 4;
    DevBuf dx(nb), dp((size_t)T * 4), dff(freq_factors ? (size_t)n_rot * 2 : 16);
    if (!dx.up(x, nb) || !dp.up(pos, (size_t)T * 4)) return MI355_ERR_OOM;
    if (freq_factors) dff.up(freq_factors, (size_t)(n_rot / 2) * 4);
    RopeArgs ra{n_rot, freq_base, freq_scale, freq_factors ? dff.as<float>() : nullptr, neox};
    hipError_t e = launch_rope_inplace(dx.as<float>(), (int)T, n_head, head_dim, dp.as<int32_t>(), ra, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "rope");
    return dx.down(x, nb) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_rope_yarn(float *x, int32_t n_head, int32_t head_dim, int32_t n_rot, const int32_t *pos, int64_t T,
                       float freq_base, float freq_scale, const float *freq_factors, int32_t neox,
                       float ext_factor, float attn_factor, float corr_lo, float corr_hi) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t nb = (size_t)T * n_head * head_dim * 4;
    DevBuf dx(nb), dp((size_t)T * 4), dff(freq_factors ? (size_t)n_rot * 2 : 16);
    if (!dx.up(x, nb) || !dp.up(pos, (size_t)T * 4)) return MI355_ERR_OOM;
    if (freq_factors) dff.up(freq_factors, (size_t)(n_rot / 2) * 4);
    RopeArgs ra{n_rot, freq_base, freq_scale, freq_factors ? dff.as<float>() : nullptr, neox};
    ra.ext_factor = ext_factor; ra.attn_factor = attn_factor; ra.corr_lo = corr_lo; ra.corr_hi = corr_hi;
    hipError_t e = launch_rope_inplace(dx.as<float>(), (int)T, n_head, head_dim, dp.as<int32_t>(), ra, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "rope_yarn");
    return dx.down(x, nb) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_get_rows(int32_t type, const void *table, int64_t K, int64_t n_rows, const int32_t *ids, int64_t n_ids, float *dst) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t grow = ggml_row_bytes(type, K), drow = dev_row_bytes(type, K);
    if (!grow) return MI355_ERR_ARG;
    DevBuf src(grow * n_rows), dev(drow * n_rows), di((size_t)n_ids * 4), dd((size_t)n_ids * K * 4);
    if (!src.up(table, grow * n_rows) || !dev.p || !di.up(ids, (size_t)n_ids * 4) || !dd.p) return MI355_ERR_OOM;
    hipError_t e = launch_repack_rows(type, src.as<uint8_t>(), dev.as<uint8_t>(), K, n_rows, nullptr);
    if (e == hipSuccess) e = launch_get_rows(type, dev.as<uint8_t>(), K, di.as<int32_t>(), (int)n_ids, dd.as<float>(), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "get_rows");
    return dd.down(dst, (size_t)n_ids * K * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_swiglu(const float *gate, const float *up, int64_t n, float *y) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    DevBuf g((size_t)n * 4), u((size_t)n * 4), o((size_t)n * 4);
    if (!g.up(gate, (size_t)n * 4) || !u.up(up, (size_t)n * 4) || !o.p) return MI355_ERR_OOM;
    hipError_t e = launch_swiglu(g.as<float>(), u.as<float>(), o.as<float>(), n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "swiglu");
    return o.down(y, (size_t)n * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_soft_max(const float *x, const float *mask, int64_t n, int64_t rows, float scale, float *y) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    const size_t nb = (size_t)n * rows * 4;
    DevBuf dx(nb), dm(nb), dy(nb);
    if (!dx.up(x, nb) || !dy.p) return MI355_ERR_OOM;
    if (mask) dm.up(mask, nb);
    hipError_t e = launch_soft_max(dx.as<float>(), mask ? dm.as<float>() : nullptr, dy.as<float>(), (int)n, (int)rows, scale, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "soft_max");
    return dy.down(y, nb) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_moe_route(const float *logits, int64_t T, int32_t n_expert, int32_t k, int32_t *ids, float *w) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if (T <= 0 || n_expert <= 0 || n_expert > 64 || k <= 0 || k > n_expert) { fail("moe_route: bad shape"); return MI355_ERR_ARG; }
    DevBuf dl((size_t)T * n_expert * 4), di((size_t)T * k * 4), dw((size_t)T * k * 4);
    if (!dl.up(logits, (size_t)T * n_expert * 4) || !di.p || !dw.p) return MI355_ERR_OOM;
    hipError_t e = launch_moe_route(dl.as<float>(), (int)T, n_expert, k, di.as<int32_t>(), dw.as<float>(), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "moe_route");
    return di.down(ids, (size_t)T * k * 4) && dw.down(w, (size_t)T * k * 4) ? MI355_OK : MI355_ERR_HIP;
}

int mi355_op_flash_attn(const float *q, int64_t T, int32_t H, int32_t G, int32_t D, int32_t type_k, const void *k, int32_t type_v,
                        const void *v, int32_t n_cells, const int32_t *cell_pos, const int32_t *q_pos, float scale, float *out) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if ((D != 64 && D != 128) || H % G) { fail("unsupported head geometry"); return MI355_ERR_ARG; }
    // Build cache planes by running the store kernel on dequantised rows is not possible bit-exactly, so the
    // planes are filled from the ggml-layout rows directly on the host.
    const size_t kv_dim = (size_t)G * D;
    auto fill = [&](int type, const void *rows, std::vector<uint8_t> &codes, std::vector<uint16_t> &scales) {
        const uint8_t *src = (const uint8_t *)rows;
        const size_t rb = ggml_row_bytes(type, (int64_t)kv_dim);
        if (type == T_F16) {
            codes.resize((size_t)G * n_cells * D * 2);
            for (int c = 0; c < n_cells; c++)
                for (int g = 0; g < G; g++)
                    memcpy(&codes[(((size_t)g * n_cells + c) * D) * 2], src + (size_t)c * rb + (size_t)g * D * 2, (size_t)D * 2);
        } else if (type == T_Q8_0) {
            codes.resize((size_t)G * n_cells * D);
            scales.resize((size_t)G * n_cells * (D / 32));
            for (int c = 0; c < n_cells; c++)
                for (int g = 0; g < G; g++)
                    for (int b = 0; b < D / 32; b++) {
                        const uint8_t *blk = src + (size_t)c * rb + ((size_t)g * (D / 32) + b) * 34;
                        memcpy(&scales[((size_t)g * n_cells + c) * (D / 32) + b], blk, 2);
                        memcpy(&codes[((size_t)g * n_cells + c) * D + (size_t)b * 32], blk + 2, 32);
                    }
        } else {
            codes.resize((size_t)G * n_cells * D / 2);
            scales.resize((size_t)G * n_cells * (D / 32));
            for (int c = 0; c < n_cells; c++)
                for (int g = 0; g < G; g++)
                    for (int b = 0; b < D / 32; b++) {
                        const uint8_t *blk = src + (size_t)c * rb + ((size_t)g * (D / 32) + b) * 18;
                        memcpy(&scales[((size_t)g * n_cells + c) * (D / 32) + b], blk, 2);
                        memcpy(&codes[((size_t)g * n_cells + c) * (D / 2) + (size_t)b * 16], blk + 2, 16);
                    }
        }
    };
    std::vector<uint8_t> kc, vc;
    std::vector<uint16_t> ks, vs;
    fill(type_k, k, kc, ks);
    fill(type_v, v, vc, vs);
    DevBuf dk(kc.size()), dks(ks.size() * 2 + 16), dv(vc.size()), dvs(vs.size() * 2 + 16);
    DevBuf dq((size_t)T * H * D * 4), dout((size_t)T * H * D * 4), dcp((size_t)n_cells * 4), dcs((size_t)n_cells * 8), dtp((size_t)T * 4), dts((size_t)T * 4), dn(16);
    if (!dk.up(kc.data(), kc.size()) || !dv.up(vc.data(), vc.size()) || !dq.up(q, (size_t)T * H * D * 4)) return MI355_ERR_OOM;
    if (!ks.empty()) dks.up(ks.data(), ks.size() * 2);
    if (!vs.empty()) dvs.up(vs.data(), vs.size() * 2);
    std::vector<uint64_t> seqm((size_t)n_cells, 1ull);
    std::vector<int32_t> tseq((size_t)T, 0);
    int32_t nkv = n_cells;
    dcp.up(cell_pos, (size_t)n_cells * 4); dcs.up(seqm.data(), (size_t)n_cells * 8); dtp.up(q_pos, (size_t)T * 4); dts.up(tseq.data(), (size_t)T * 4);
    dn.up(&nkv, 4);
    AttnArgs a{};
    a.q = dq.as<float>(); a.out = dout.as<float>();
    a.kv.k = dk.as<uint8_t>(); a.kv.kd = dks.as<uint16_t>(); a.kv.v = dv.as<uint8_t>(); a.kv.vd = dvs.as<uint16_t>();
    a.type_k = type_k; a.type_v = type_v; a.T = (int)T; a.H = H; a.G = G; a.D = D; a.n_ctx = n_cells;
    a.cell_pos = dcp.as<int32_t>(); a.cell_seq = dcs.as<uint64_t>(); a.tok_pos = dtp.as<int32_t>(); a.tok_seq = dts.as<int32_t>();
    a.n_kv_dev = dn.as<int32_t>(); a.n_kv_max = n_cells; a.scale = scale;
    a.splits = flash_attn_pick_splits((int)T, G, n_cells);
    a.pf_splits = flash_attn_prefill_splits((int)T, H, G, D, n_cells);
    DevBuf part(flash_attn_workspace_floats((int)T, H, D, std::max(a.splits, a.pf_splits)) * 4);
    a.part = part.as<float>();
    hipError_t e = launch_flash_attn(a, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return hip_fail(e, "flash_attn");
    return dout.down(out, (size_t)T * H * D * 4) ? MI355_OK : MI355_ERR_HIP;
}

// The attention block of ONE single-token step exactly as the decode path launches it (test entry): rope of q
// and of the token's K row, the K / V row quantised into the cache, attention over the visible cells, merge of the chunk partials, Q8_K quantisation and the
// attn_output mat-vec with its residual.  fused = 1: attn_out.hip (one launch); 0: the single-launch decode attention + the weight-stream mat-vec.
// k / v: the cache BEFORE the step as ggml-layout rows [n_cells][G * D] (the row at tok_cell is overwritten); cell_pos[tok_cell] must already be tok_pos.
int mi355_op_attn_step(const float *q, const float *k_new, const float *v_new, int32_t H, int32_t G, int32_t D, int32_t type_k, const void *k, int32_t type_v,
                       const void *v, int32_t n_cells, const int32_t *cell_pos, int32_t tok_pos, int32_t tok_cell, float rope_base, int32_t n_rot, float scale,
                       int32_t type_o, const void *W_o, int64_t E, const float *resid, int32_t fused, float *att_out, float *out, void *k_row_out, void *v_row_out) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if ((D != 64 && D != 128) || G < 1 || H % G || n_cells < 1 || tok_cell < 0 || tok_cell >= n_cells) { fail("bad geometry"); return MI355_ERR_ARG; }
    const int64_t K = (int64_t)H * D;
    const size_t kv_dim = (size_t)G * D;
    auto fill = [&](int type, const void *rows, std::vector<uint8_t> &codes, std::vector<uint16_t> &scales) {
        const uint8_t *src = (const uint8_t *)rows;
        const size_t rb = ggml_row_bytes(type, (int64_t)kv_dim);
        if (type == T_F16) {
            codes.resize((size_t)G * n_cells * D * 2);
            for (int c = 0; c < n_cells; c++)
                for (int g = 0; g < G; g++) memcpy(&codes[(((size_t)g * n_cells + c) * D) * 2], src + (size_t)c * rb + (size_t)g * D * 2, (size_t)D * 2);
        } else {
            const int bb = type == T_Q8_0 ? 34 : 18, cb = type == T_Q8_0 ? 32 : 16;
            codes.resize((size_t)G * n_cells * (D / 32) * cb);
            scales.resize((size_t)G * n_cells * (D / 32));
            for (int c = 0; c < n_cells; c++)
                for (int g = 0; g < G; g++)
                    for (int b = 0; b < D / 32; b++) {
                        const uint8_t *blk = src + (size_t)c * rb + ((size_t)g * (D / 32) + b) * bb;
                        memcpy(&scales[((size_t)g * n_cells + c) * (D / 32) + b], blk, 2);
                        memcpy(&codes[(((size_t)g * n_cells + c) * (D / 32) + b) * cb], blk + 2, (size_t)cb);
                    }
        }
    };
    if ((type_k != T_F16 && type_k != T_Q8_0 && type_k != T_Q4_0) || (type_v != T_F16 && type_v != T_Q8_0 && type_v != T_Q4_0)) { fail("bad cache type"); return MI355_ERR_ARG; }
    std::vector<uint8_t> kc, vc;
    std::vector<uint16_t> ks, vs;
    fill(type_k, k, kc, ks);
    fill(type_v, v, vc, vs);
    const size_t grow = ggml_row_bytes(type_o, K), drow = dev_row_bytes(type_o, K);
    if (!grow || K % 256) { fail("bad attn_output type / K"); return MI355_ERR_ARG; }
    DevBuf dk(kc.size()), dks(ks.size() * 2 + 16), dv(vc.size()), dvs(vs.size() * 2 + 16);
    DevBuf dq((size_t)K * 4), dkn(kv_dim * 4), dvn(kv_dim * 4), datt((size_t)K * 4), dcp((size_t)n_cells * 4), dcs((size_t)n_cells * 8), dtp(16), dts(16), dn(16), dcell(16);
    DevBuf wsrc(grow * E), wdev(drow * E), dres((size_t)E * 4), dout((size_t)E * 4), dcnt(64 * ATT_SYNC_STRIDE * 4), dflags(64 * ATT_SYNC_STRIDE * 4), dserial(64), dcsb((size_t)std::max(n_rot, 4) * 4 + 64);
    ActBufs ab((size_t)K, 1);
    if (!dk.up(kc.data(), kc.size()) || !dv.up(vc.data(), vc.size()) || !dq.up(q, (size_t)K * 4) || !dkn.up(k_new, kv_dim * 4) || !dvn.up(v_new, kv_dim * 4) ||
        !wsrc.up(W_o, grow * E) || !wdev.p || !dres.up(resid, (size_t)E * 4) || !dout.p || !ab.ok() || !datt.p) { fail("device alloc/copy failed"); return MI355_ERR_OOM; }
    if (!ks.empty()) dks.up(ks.data(), ks.size() * 2);
    if (!vs.empty()) dvs.up(vs.data(), vs.size() * 2);
    std::vector<uint64_t> seqm((size_t)n_cells, 1ull);
    const int32_t tseq = 0, nkv = n_cells;
    const unsigned one = 1u;
    dcp.up(cell_pos, (size_t)n_cells * 4); dcs.up(seqm.data(), (size_t)n_cells * 8); dtp.up(&tok_pos, 4); dts.up(&tseq, 4); dn.up(&nkv, 4); dcell.up(&tok_cell, 4);
    dserial.up(&one, 4);
    hipError_t e = launch_repack_rows(type_o, wsrc.as<uint8_t>(), wdev.as<uint8_t>(), K, E, nullptr);
    if (e != hipSuccess) return hip_fail(e, "repack");
    RopeArgs ra{};
    ra.n_rot = n_rot; ra.freq_base = rope_base; ra.freq_scale = 1.0f; ra.freq_factors = nullptr; ra.neox = 0;
    e = launch_rope_table(dtp.as<int32_t>(), 1, ra, dcsb.as<float>(), nullptr);
    if (e != hipSuccess) return hip_fail(e, "rope_table");
    AttnArgs a{};
    a.q = dq.as<float>(); a.out = datt.as<float>();
    a.kv.k = dk.as<uint8_t>(); a.kv.kd = dks.as<uint16_t>(); a.kv.v = dv.as<uint8_t>(); a.kv.vd = dvs.as<uint16_t>();
    a.type_k = type_k; a.type_v = type_v; a.T = 1; a.H = H; a.G = G; a.D = D; a.n_ctx = n_cells;
    a.cell_pos = dcp.as<int32_t>(); a.cell_seq = dcs.as<uint64_t>(); a.tok_pos = dtp.as<int32_t>(); a.tok_seq = dts.as<int32_t>();
    a.n_kv_dev = dn.as<int32_t>(); a.n_kv_max = n_cells; a.scale = scale;
    a.out_q = &ab.q; a.out_q8k = true; a.out_q80 = false;
    MMVQSeg so{};
    so.W = wdev.as<uint8_t>(); so.out = dout.as<float>(); so.resid = dres.as<float>(); so.type = type_o; so.n_rows = (int)E; so.ld_out = (int)E; so.row_bytes = drow;
    if (fused) {
        a.splits = attn_out_fused_splits(a);
        DevBuf part(flash_attn_workspace_floats(1, H, D, a.splits) * 4);
        a.part = part.as<float>();
        if (!attn_out_fused_applicable(a, ra, so, (int)K, EPI_ADD)) { fail("attn_out.hip has no form for this shape"); return MI355_ERR_ARG; }
        DevBuf dgran(attn_out_granule_words((int)K) * 8);
        if (!dgran.p) return MI355_ERR_OOM;
        e = launch_attn_out_fused(a, dcsb.as<float>(), ra, dkn.as<float>(), dvn.as<float>(), dcell.as<int32_t>(), dcnt.as<unsigned>(), dflags.as<unsigned>(),
                                  dgran.as<unsigned long long>(), 3, dserial.as<unsigned>(), so, (int)K, EPI_ADD, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return hip_fail(e, "attn_out_fused");
    } else {
        a.splits = flash_attn_decode_splits(n_cells);
        DevBuf part(flash_attn_workspace_floats(1, H, D, a.splits) * 4);
        a.part = part.as<float>();
        if (!flash_attn_decode_fused_applicable(a, ra)) { fail("the single-launch decode attention has no form for this shape"); return MI355_ERR_ARG; }
        e = launch_flash_attn_decode_fused(a, dcsb.as<float>(), ra, dkn.as<float>(), dvn.as<float>(), dcell.as<int32_t>(), dcnt.as<unsigned>(), nullptr);
        if (e != hipSuccess) return hip_fail(e, "flash_attn_decode_fused");
        MMVQArgs m{};
        m.n_seg = 1; m.K = (int)K; m.T = 1; m.epi = EPI_ADD; m.seg[0] = so;
        m.aq = ab.q.qs; m.ad = ab.q.d; m.abs = ab.q.bsums; m.aq0 = ab.q.qs0; m.ad0 = ab.q.d0;
        e = launch_mmvq(m, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) return hip_fail(e, "attn_output mat-vec");
    }
    if (att_out && !datt.down(att_out, (size_t)K * 4)) return MI355_ERR_HIP;
    if (out && !dout.down(out, (size_t)E * 4)) return MI355_ERR_HIP;
    // the cache row the step wrote, back in ggml block layout
    auto row_back = [&](int type, DevBuf &codes, DevBuf &scales, void *dst) -> bool {
        if (!dst) return true;
        uint8_t *o8 = (uint8_t *)dst;
        for (int g = 0; g < G; g++) {
            const size_t rowi = (size_t)g * n_cells + tok_cell;
            if (type == T_F16) {
                if (hipMemcpy(o8 + (size_t)g * D * 2, (uint8_t *)codes.p + rowi * D * 2, (size_t)D * 2, hipMemcpyDeviceToHost) != hipSuccess) return false;
            } else {
                const int bb = type == T_Q8_0 ? 34 : 18, cb = type == T_Q8_0 ? 32 : 16;
                std::vector<uint8_t> c((size_t)(D / 32) * cb);
                std::vector<uint16_t> sc((size_t)D / 32);
                if (hipMemcpy(c.data(), (uint8_t *)codes.p + rowi * (D / 32) * cb, c.size(), hipMemcpyDeviceToHost) != hipSuccess) return false;
                if (hipMemcpy(sc.data(), (uint8_t *)scales.p + rowi * (D / 32) * 2, sc.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return false;
                for (int b = 0; b < D / 32; b++) {
                    uint8_t *blk = o8 + ((size_t)g * (D / 32) + b) * bb;
                    memcpy(blk, &sc[(size_t)b], 2);
                    memcpy(blk + 2, &c[(size_t)b * cb], (size_t)cb);
                }
            }
        }
        return true;
    };
    if (!row_back(type_k, dk, dks, k_row_out) || !row_back(type_v, dv, dvs, v_row_out)) return MI355_ERR_HIP;
    return MI355_OK;
}

int mi355_debug_set_option(const char *name, int32_t value) {
    if (!name) return MI355_ERR_ARG;
    if (!strcmp(name, "mmq_planes")) { g_op_mmq_planes = value != 0; return MI355_OK; }
    if (!strcmp(name, "mmq_tiles")) { mmq_set_tiles(value); return MI355_OK; }
    if (!strcmp(name, "mmq_split")) { mmq_set_split(value); return MI355_OK; }
    if (!strcmp(name, "mmq_lds_form")) { mmq_set_lds_form(value); return MI355_OK; }
    if (!strcmp(name, "mmq_ksplit")) { g_op_mmq_ksplit = value != 0; return MI355_OK; }
    if (!strcmp(name, "attn_store_fuse")) { set_attn_store_fuse(value != 0); return MI355_OK; }
    if (!strcmp(name, "rope_fast")) { set_rope_fast(value != 0); return MI355_OK; }
    if (!strcmp(name, "decode_mega") || !strcmp(name, "decode_engine")) {
        // the whole-step kernel and the layer engine are experiments that lost their A/Bs: only a library built with MI355_BUILD_EXPERIMENTS=1 holds them
        if (value > 0 && !experiments_built()) { fail(std::string(name) + ": the experiment kernels are not in this build (MI355_BUILD_EXPERIMENTS=1 python cortex.llamacpp_amd/build.py --force)"); return MI355_ERR_ARG; }
        if (name[7] == 'm') set_decode_mega(value != 0); else set_decode_engine(value);
        return MI355_OK;
    }
    if (!strcmp(name, "tp_p2p")) { tp_p2p_use(value != 0); return MI355_OK; }
    if (!strcmp(name, "tp_p2p_prompt")) { tp_p2p_use_prompt(value != 0); return MI355_OK; }
    if (!strcmp(name, "mmvq_stream")) { mmvq_set_stream(value != 0); return MI355_OK; }
    if (!strcmp(name, "attn_out_fused")) { set_attn_out_fused(value); return MI355_OK; }     // (contexts created afterwards)
    if (!strcmp(name, "qkv_attn_fused")) { set_qkv_attn_fused(value); return MI355_OK; }     // Q | K | V inside the attention launch (applies from the next step's launches on; graphs captured earlier keep their form)
    if (!strcmp(name, "moe_group_min")) { set_moe_group_min(value); return MI355_OK; }
    if (!strcmp(name, "fa_v_acc_f16")) { set_fa_v_acc_f16(value); return MI355_OK; }              // f16 cache: the CPU path's fp16 V accumulation (parity mode)
    if (!strcmp(name, "raise_stream_error")) { debug_raise_stream_error((unsigned)value); return MI355_OK; }   // tests: what a timed-out in-kernel wait does
    if (!strcmp(name, "tp_null_group")) { tp_set_null_group(0, value); return MI355_OK; }
    fail(std::string("unknown option ") + name);
    return MI355_ERR_ARG;
}

// simple read-bandwidth probe: sum-reduce `bytes` of device memory
// ---------------------------------------------------------------- row split group
int mi355_tp_unique_id(void *id_out, size_t cap) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    std::string err;
    const int n = tp_unique_id(id_out, cap, err);
    if (n < 0) { fail(err); return MI355_ERR_ARG; }
    return n;
}
int mi355_tp_init(int32_t device, int32_t rank, int32_t size, const void *id, size_t id_len) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) { fail("hipSetDevice failed"); return MI355_ERR_ARG; }
    std::string err;
    if (tp_init(rank, size, id, id_len, err) != 0) { fail(err); return MI355_ERR_ARG; }
    return MI355_OK;
}
void mi355_tp_shutdown(void) { tp_shutdown(); }
int mi355_tp_worker_main(int sock_fd) {
    try {
        if (!g_backend_ok && mi355_backend_init() != MI355_OK) return 66;
        return tp_split_worker_main(sock_fd);
    } catch (const std::exception &e) {
        fprintf(stderr, "mi355_tp_worker: %s\n", e.what());
        return 70;
    } catch (...) { return 70; }
}
int32_t mi355_tp_rank(void) { return tp_rank(); }
int32_t mi355_tp_size(void) { return tp_size(); }
int mi355_tp_set_host_exchange(mi355_tp_host_exchange fn, void *user, int32_t rank, int32_t size) {
    if (fn && (size < 1 || rank < 0 || rank >= size)) { fail("bad rank / size"); return MI355_ERR_ARG; }
    tp_set_host_exchange(reinterpret_cast<tp_host_exchange_fn>(fn), user, rank, size);
    return MI355_OK;
}

int mi355_tp_p2p_local_handle2(void *out, size_t cap, size_t max_floats, size_t prompt_floats) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    std::string err;
    const int n = tp_p2p_local_handle(out, cap, max_floats, prompt_floats, err);
    if (n < 0) { fail(err); return MI355_ERR_ARG; }
    return n;
}
int mi355_tp_p2p_local_handle(void *out, size_t cap, size_t max_floats) { return mi355_tp_p2p_local_handle2(out, cap, max_floats, 0); }
int mi355_tp_p2p_enable(const void *handles, size_t len) {
    if (!need_device()) return MI355_ERR_NO_DEVICE;
    std::string err;
    if (tp_p2p_enable(handles, len, err) != 0) { fail(err); return MI355_ERR_HIP; }
    return MI355_OK;
}
int64_t mi355_tp_p2p_exchanges(void) { return tp_p2p_exchanges(); }
int64_t mi355_tp_p2p_prompt_exchanges(void) { return tp_p2p_prompt_exchanges(); }

double mi355_bench_hbm_read(size_t bytes, int iters) {
    if (!need_device()) return -1.0;
    return hbm_read_probe(bytes, iters);
}

// ---------------------------------------------------------------- tokenizer
static Vocab *model_vocab(mi355_model *m) {
    if (!m) { fail("null model"); return nullptr; }
    if (!m->vocab_tried) {
        m->vocab_tried = true;
        std::string err;
        m->vocab_ok = m->vocab.load(*m->m->file, err);
        if (!m->vocab_ok) fail("tokenizer: " + err);
    }
    return m->vocab_ok ? &m->vocab : nullptr;
}
int32_t mi355_tokenize(mi355_model *m, const char *text, int32_t text_len, mi355_token *out, int32_t cap, int32_t add_special, int32_t parse_special) {
    MI355_GUARD(return INT32_MIN,
        Vocab *v = model_vocab(m);
        if (!v || !text || text_len < 0) return INT32_MIN;
        const std::vector<int32_t> ids = v->tokenize(std::string(text, (size_t)text_len), add_special != 0, parse_special != 0);
        if ((int64_t)ids.size() > cap || !out) return -(int32_t)ids.size();
        memcpy(out, ids.data(), ids.size() * sizeof(int32_t));
        return (int32_t)ids.size();
    )
}
int32_t mi355_token_to_piece(mi355_model *m, mi355_token tok, char *buf, int32_t cap, int32_t special) {
    MI355_GUARD(return INT32_MIN,
        Vocab *v = model_vocab(m);
        if (!v) return INT32_MIN;
        const std::string p = v->token_to_piece(tok, special != 0);
        if ((int64_t)p.size() > cap || !buf) return -(int32_t)p.size();
        memcpy(buf, p.data(), p.size());
        return (int32_t)p.size();
    )
}
mi355_token mi355_token_bos(mi355_model *m) { Vocab *v = model_vocab(m); return v ? v->bos() : -1; }
mi355_token mi355_token_eos(mi355_model *m) { Vocab *v = model_vocab(m); return v ? v->eos() : -1; }
int32_t mi355_token_is_eog(mi355_model *m, mi355_token tok) { Vocab *v = model_vocab(m); return v && v->is_eog(tok) ? 1 : 0; }

// ---------------------------------------------------------------- engine
static bool parse_body(const char *txt, Json &j, mi355_engine_callback cb, void *user) {
    std::string err;
    if (txt && Json::parse(txt, j, &err)) return true;
    if (cb) {
        Json st = Json::object(), body = Json::object();
        st["is_done"] = true; st["has_error"] = true; st["is_stream"] = false; st["status_code"] = 400;
        body["message"] = "Malformed JSON body: " + err;
        cb(st.dump().c_str(), body.dump().c_str(), user);
    }
    return false;
}
// The engine copies the callback into whatever outlives the call (a queued stream task); every copy shares `guard`, so the host's release function runs
// exactly once per request, after the last callback the request will ever make - also for a request that ends without a terminal callback (a stream
// cancelled by StopInferencing: src/llama_engine.cc:950-955 breaks out of the loop without one)
static LlamaEngine::Callback wrap_cb(mi355_engine *e, mi355_engine_callback cb, void *user) {
    const mi355_release_fn rel = e->release.load();
    std::shared_ptr<void> guard(user, [rel](void *u) { if (rel) rel(u); });
    return [cb, user, guard](Json &&st, Json &&body) {
        if (!cb) return;
        const std::string s = st.dump(), b = body.dump();
        cb(s.c_str(), b.c_str(), user);
    };
}
// an exception below an engine entry point: answered in the reference's error shape (status 500), never thrown at the host
static void engine_exception(mi355_engine_callback cb, void *user, const char *what) {
    fail(std::string("internal error: ") + what);
    if (!cb) return;
    try {
        Json st = Json::object(), body = Json::object();
        st["is_done"] = true; st["has_error"] = true; st["is_stream"] = false; st["status_code"] = 500;
        body["message"] = std::string("Internal error: ") + what;
        cb(st.dump().c_str(), body.dump().c_str(), user);
    } catch (...) {}
}
mi355_engine *mi355_engine_create(void) {
    if (!g_backend_ok && mi355_backend_init() != MI355_OK) return nullptr;
    MI355_GUARD(return nullptr, return new mi355_engine;)
}
void mi355_engine_destroy(mi355_engine *e) { delete e; }
#define MI355_ENGINE_FWD(cname, Method)                                                                     \
    void cname(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user) {              \
        try {                                                                                               \
            Json j;                                                                                         \
            if (!e) return;                                                                                 \
            LlamaEngine::Callback w = wrap_cb(e, cb, user);   /* from here on `user` is released with w */   \
            if (!parse_body(body_json, j, cb, user)) return;                                                \
            e->eng.Method(j, std::move(w));                                                                 \
        } catch (const std::exception &ex) {                                                                \
            engine_exception(cb, user, ex.what());                                                          \
        } catch (...) {                                                                                     \
            engine_exception(cb, user, "unknown exception");                                                \
        }                                                                                                   \
    }
MI355_ENGINE_FWD(mi355_engine_load_model, LoadModel)
MI355_ENGINE_FWD(mi355_engine_unload_model, UnloadModel)
MI355_ENGINE_FWD(mi355_engine_get_model_status, GetModelStatus)
MI355_ENGINE_FWD(mi355_engine_get_models, GetModels)
MI355_ENGINE_FWD(mi355_engine_handle_chat_completion, HandleChatCompletion)
MI355_ENGINE_FWD(mi355_engine_handle_embedding, HandleEmbedding)
#undef MI355_ENGINE_FWD
void mi355_engine_set_release_callback(mi355_engine *e, void (*release)(void *user)) { if (e) e->release.store(release); }
int32_t mi355_engine_is_supported(mi355_engine *e, const char *feature) { return e && feature && e->eng.IsSupported(feature) ? 1 : 0; }
void mi355_engine_stop_inferencing(mi355_engine *e, const char *model_id) { if (e && model_id) e->eng.StopInferencing(model_id); }
void mi355_engine_load(mi355_engine *e, const char *engine_path, const char *deps_path, int32_t is_custom_engine_path, const char *log_path, int32_t max_log_lines,
                       int32_t log_level) {
    if (!e) return;
    MI355_GUARD(return, {
        LlamaEngine::EngineLoadOption o;
        o.engine_path = engine_path ? engine_path : ""; o.deps_path = deps_path ? deps_path : ""; o.is_custom_engine_path = is_custom_engine_path != 0;
        o.log_path = log_path ? log_path : ""; o.max_log_lines = max_log_lines; o.log_level = log_level;
        e->eng.Load(o);
        return;
    })
}
void mi355_engine_unload(mi355_engine *e) { if (e) { MI355_GUARD(return, { e->eng.Unload(); return; }) } }
void mi355_engine_set_file_logger(mi355_engine *e, int32_t max_log_lines, const char *log_path) {
    if (e) { MI355_GUARD(return, { e->eng.SetFileLogger(max_log_lines, log_path ? log_path : ""); return; }) }
}
void mi355_engine_set_log_level(mi355_engine *e, int32_t log_level) { if (e) e->eng.SetLogLevel(log_level); }
void mi355_engine_set_log_callback(mi355_engine *e, mi355_log_callback cb, void *user) { (void)e; log_set_callback(cb, user); }

}  // extern "C"
