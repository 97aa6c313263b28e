// runtime.cc — GGUF -> HBM residency, KV cells, and the per-batch executor for the `llama` architecture
// (op order of upstream llm_build_llama / build_attn / build_ffn / build_moe_ffn).
// Reference caller: LlamaServerContext::UpdateSlots -> llama_decode (src/llama_server_context.cc:1628-1635).
#include "runtime.h"

#include <dlfcn.h>
#include "tp_comm.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace mi355 {

unsigned stream_error_epoch();           // error epochs opened so far in this process (Context::stream_check)
static unsigned *stream_error_word();   // pinned, one per process: raised by a weight-stream / engine kernel whose bounded wait gave up

static thread_local std::string g_err;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}
const std::string &last_error_string() { return g_err; }

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_;                                                                    \
        }                                                                                 \
    } while (0)

// ------------------------------------------------------------------------------------------ model
Model::~Model() {
    for (uint8_t *p : arenas) (void)hipFree(p);
}

// the rotary parameters every kernel that rotates takes (with_ff: the per-pair frequency factors of rope_freqs.weight, when the file has them)
static RopeArgs rope_args(const Model &m, bool with_ff) {
    const HParams &hp = m.hp;
    RopeArgs ra{hp.n_rot, hp.rope_base, hp.rope_scale, with_ff && m.rope_freqs.valid() ? (const float *)m.rope_freqs.data : nullptr, hp.rope_neox};
    ra.ext_factor = hp.yarn_ext; ra.attn_factor = hp.yarn_attn; ra.corr_lo = hp.yarn_lo; ra.corr_hi = hp.yarn_hi;
    return ra;
}

static bool type_supported(int t) {
    return t == T_F32 || t == T_F16 || t == T_Q8_0 || t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q2_K || t == T_Q3_K || t == T_Q4_0 || t == T_Q5_0 || t == T_IQ4_NL;
}

Model *model_load(const std::string &path, int main_gpu, std::string &err, int &status, int prefill_planes, int tp_rank, int tp_size) {
    status = 0;
    std::unique_ptr<Model> m(new Model());
    m->file.reset(new GGUFFile());
    m->path = path;
    err = m->file->open(path);
    if (!err.empty()) { status = err.rfind("cannot", 0) == 0 ? -101 : -102; return nullptr; }
    GGUFFile &f = *m->file;
    HParams &hp = m->hp;
    hp.arch = f.get_s("general.architecture", "");
    if (hp.arch.empty()) { err = "general.architecture missing"; status = -102; return nullptr; }
    // the graph built here is llm_build_llama's: "llama" files (Llama, Mistral, TinyLlama, Mixtral ... all carry that name) and "qwen2"
    // (the same op order with NEOX rope pairing and Q / K / V biases).  Gemma, Phi-3, BERT-type encoders etc. are other graphs: refused, never run as llama
    // "nomic-bert" (the reference's embedding smoke model, Makefile:6) is the one encoder graph: llm_build_bert's NOMIC_BERT branches (run_layers_encoder)
    if (hp.arch != "llama" && hp.arch != "qwen2" && hp.arch != "nomic-bert") { err = "unsupported general.architecture '" + hp.arch + "' (this backend builds the llama graph - llama, qwen2 - and the nomic-bert encoder)"; status = -102; return nullptr; }
    hp.encoder = hp.arch == "nomic-bert";
    const std::string a = hp.arch + ".";
    hp.n_embd = (int)f.get_u(a + "embedding_length", 0);
    hp.n_layer = (int)f.get_u(a + "block_count", 0);
    hp.n_ff = (int)f.get_u(a + "feed_forward_length", 0);
    hp.n_head = (int)f.get_u(a + "attention.head_count", 0);
    hp.n_head_kv = (int)f.get_u(a + "attention.head_count_kv", (uint64_t)hp.n_head);
    hp.eps = (float)f.get_f(a + (hp.encoder ? "attention.layer_norm_epsilon" : "attention.layer_norm_rms_epsilon"), hp.encoder ? 1e-12 : 1e-5);
    hp.rope_base = (float)f.get_f(a + "rope.freq_base", 10000.0);
    hp.n_expert = (int)f.get_u(a + "expert_count", 0);
    hp.n_expert_used = (int)f.get_u(a + "expert_used_count", 0);
    hp.n_ctx_train = (int)f.get_u(a + "context_length", 0);
    hp.pooling_type = (int)f.get_u(a + "pooling_type", 0);
    if (hp.n_embd <= 0 || hp.n_layer <= 0 || hp.n_head <= 0) { err = "missing hyper-parameters for arch " + hp.arch; status = -102; return nullptr; }
    if (hp.n_layer > 1024 || hp.n_embd > (1 << 20) || hp.n_head > 4096) { err = "implausible hyper-parameters for arch " + hp.arch; status = -102; return nullptr; }
    // the head counts size buffers and pick kernels: check them here, not at the first decode
    if (hp.n_head_kv <= 0 || hp.n_head % hp.n_head_kv) { err = "attention.head_count_kv (" + std::to_string(hp.n_head_kv) + ") must be positive and divide attention.head_count (" + std::to_string(hp.n_head) + ")"; status = -102; return nullptr; }
    {
        const int ratio = hp.n_head / hp.n_head_kv;
        // (1, 2, 4, 8 have the tuned single-launch decode attention and the matrix-core prompt attention; 3, 5, 6, 7 take the general split kernel)
        if (ratio < 1 || ratio > 8) { err = "unsupported query / kv head ratio " + std::to_string(ratio) + " (the attention kernels are built for 1 .. 8)"; status = -102; return nullptr; }
    }
    if (hp.n_embd % hp.n_head) { err = "embedding_length is not a multiple of attention.head_count"; status = -102; return nullptr; }
    if (hp.n_expert < 0 || hp.n_expert > 256 || hp.n_expert_used < 0 || hp.n_expert_used > hp.n_expert || (hp.n_expert > 0 && hp.n_expert_used == 0)) {
        err = "bad expert_count / expert_used_count"; status = -102; return nullptr;
    }
    hp.head_dim = hp.n_embd / hp.n_head;
    hp.n_rot = (int)f.get_u(a + "rope.dimension_count", (uint64_t)hp.head_dim);
    hp.rope_neox = hp.arch != "llama";
    const std::string scaling = f.get_s(a + "rope.scaling.type", "none");
    if (scaling == "linear") hp.rope_scale = 1.0f / (float)f.get_f(a + "rope.scaling.factor", 1.0);
    if (scaling == "yarn") {
        // YaRN (llama.cpp: rope_yarn / ggml_rope_yarn_corr_dims with the context defaults beta_fast 32, beta_slow 1, ext_factor 1): pairs that turn more than
        // beta_fast times over the ORIGINAL context keep their angle, pairs that turn less than beta_slow times are interpolated by 1/factor, a linear ramp between
        const float factor = (float)f.get_f(a + "rope.scaling.factor", 1.0);
        if (!(factor > 0.0f)) { err = "rope.scaling.factor must be positive"; status = -102; return nullptr; }
        const float n_orig = (float)f.get_u(a + "rope.scaling.original_context_length", f.get_u(a + "context_length", 4096));
        hp.rope_scale = 1.0f / factor;
        hp.yarn_ext = 1.0f;
        hp.yarn_attn = (float)f.get_f(a + "rope.scaling.attn_factor", 1.0);
        const float two_log_base = 2.0f * logf(hp.rope_base);
        const float lo = floorf((float)hp.n_rot * logf(n_orig / (32.0f * 2.0f * 3.14159265358979323846f)) / two_log_base);
        const float hi = ceilf((float)hp.n_rot * logf(n_orig / (1.0f * 2.0f * 3.14159265358979323846f)) / two_log_base);
        hp.yarn_lo = lo > 0.0f ? lo : 0.0f;
        hp.yarn_hi = hi < (float)(hp.n_rot - 1) ? hi : (float)(hp.n_rot - 1);
    } else if (scaling != "none" && scaling != "linear") { err = "unsupported rope.scaling.type " + scaling; status = -102; return nullptr; }
    if (hp.head_dim != 64 && hp.head_dim != 128) { err = "unsupported head_dim " + std::to_string(hp.head_dim); status = -102; return nullptr; }
    if (hp.n_embd % 256) { err = "n_embd must be a multiple of 256"; status = -102; return nullptr; }
    // ---- row split: this rank's share of the heads and of the feed-forward width
    const int P = tp_size > 1 ? tp_size : 1, R = tp_size > 1 ? tp_rank : 0;
    hp.n_head_full = hp.n_head; hp.n_head_kv_full = hp.n_head_kv; hp.n_ff_full = hp.n_ff;
    hp.tp_rank = R; hp.tp_size = P;
    // the exchange steps run whenever the process has a group of that size — also a group of ONE rank, which is how the
    // RCCL calls (and their capture into graphs) are exercised on a single GPU
    hp.tp_exchange = tp_active() && mi355::tp_size() == P && mi355::tp_rank() == R;
    if (P > 1 && !hp.tp_exchange) { err = "tp_size > 1 needs the process's row-split group first (mi355_tp_init with the same rank / size)"; status = -102; return nullptr; }
    if (P > 1) {
        if (R < 0 || R >= P) { err = "tp_rank out of range"; status = -102; return nullptr; }
        if (hp.n_expert > 0) { err = "row split of mixture-of-experts files is not supported"; status = -102; return nullptr; }
        if (hp.n_head % P || hp.n_head_kv % P) { err = "tp_size must divide the head counts (" + std::to_string(hp.n_head) + " / " + std::to_string(hp.n_head_kv) + ")"; status = -102; return nullptr; }
        if (((hp.n_head / P) * hp.head_dim) % 256) { err = "a rank's attention width must be a multiple of 256"; status = -102; return nullptr; }
        hp.n_head /= P; hp.n_head_kv /= P;
    }

    // plan the arena
    // where a tensor's bytes come from: the whole tensor, or this rank's rows (a contiguous range), or this rank's
    // columns (the same block range of every row: a strided copy)
    enum { SPLIT_NONE = 0, SPLIT_ROWS = 1, SPLIT_COLS = 2 };
    struct Plan { const GGUFTensorInfo *ti; DevTensor *dst; size_t off; size_t src_off, src_pitch, src_width; int64_t src_rows; size_t src_bytes; bool extra_copy; };
    std::vector<Plan> plan;
    size_t total = 0, max_stage = 0;
    bool fail = false;
    // cpart / cparts: column part cpart of cparts of THIS RANK's tensor as a tensor of its own (a second copy for the single-token steps, see LayerWeights::down_lo)
    auto want = [&](const std::string &name, DevTensor &dst, bool required, int split = SPLIT_NONE, int cpart = 0, int cparts = 1) {
        const GGUFTensorInfo *ti = f.tensor(name);
        if (!ti) {
            if (required) { err = "missing tensor " + name; fail = true; }
            return;
        }
        if (!type_supported(ti->type)) {
            err = "tensor " + name + " has unsupported type " + ggml_type_name(ti->type);
            fail = true;
            return;
        }
        dst.name = name;
        dst.type = ti->type;
        dst.K = ti->ne[0];
        dst.N = ti->ne[1];
        dst.n_expert = ti->ne[2];
        if (ggml_block_elems(dst.type) > 1 && dst.K % ggml_block_elems(dst.type)) { err = "tensor " + name + ": row length not a block multiple"; fail = true; return; }
        const size_t full_row = ggml_row_bytes(dst.type, dst.K);
        Plan pl{ti, &dst, total, 0, full_row, full_row, ti->n_dims == 1 ? 1 : dst.N * dst.n_expert, (size_t)ti->bytes, cparts > 1};
        if (P > 1 && split != SPLIT_NONE) {
            const int64_t blk = std::max<int64_t>(ggml_block_elems(dst.type), 1);
            if (ti->n_dims == 1 || split == SPLIT_COLS) {          // a bias vector is cut like the rows it is added to
                const int64_t unit = split == SPLIT_COLS && blk > 1 ? std::max<int64_t>(blk, 256) : blk;
                if (dst.K % P || (dst.K / P) % unit) { err = "tensor " + name + ": row length " + std::to_string(dst.K) + " cannot be cut " + std::to_string(P) + " ways on block boundaries"; fail = true; return; }
                dst.K /= P;
                pl.src_width = ggml_row_bytes(dst.type, dst.K);
                pl.src_off = (size_t)R * pl.src_width;
            } else {
                if (dst.N % P) { err = "tensor " + name + ": " + std::to_string(dst.N) + " rows cannot be cut " + std::to_string(P) + " ways"; fail = true; return; }
                dst.N /= P;
                pl.src_rows = dst.N;
                pl.src_off = (size_t)R * (size_t)dst.N * full_row;
            }
            pl.src_bytes = pl.src_width * (size_t)pl.src_rows;
        }
        if (cparts > 1) {                                          // (on 256-element boundaries, checked by the caller)
            dst.name = name + "[cols " + std::to_string(cpart) + "/" + std::to_string(cparts) + "]";
            dst.K /= cparts;
            pl.src_width = ggml_row_bytes(dst.type, dst.K);
            pl.src_off += (size_t)cpart * pl.src_width;
            pl.src_bytes = pl.src_width * (size_t)pl.src_rows;
        }
        dst.row_bytes = ti->n_dims == 1 ? ggml_row_bytes(dst.type, dst.K) : dev_row_bytes(dst.type, dst.K);
        const int64_t rows = ti->n_dims == 1 ? 1 : dst.N * dst.n_expert;
        dst.bytes = dst.row_bytes * (size_t)rows;
        dst.ggml_bytes = pl.src_bytes;
        plan.push_back(pl);
        total += (dst.bytes + 255) & ~(size_t)255;
        if (dst.type == T_Q6_K || dst.type == T_Q8_0 || dst.type == T_Q2_K || dst.type == T_Q3_K || dst.type == T_Q4_0 || dst.type == T_Q5_0 || dst.type == T_IQ4_NL || dst.row_bytes != ggml_row_bytes(dst.type, dst.K)) max_stage = std::max(max_stage, pl.src_bytes);
    };
    want("token_embd.weight", m->tok_embd, true);
    if (hp.encoder) {
        if (P > 1) { err = "row split of encoder files is not supported"; status = -102; return nullptr; }
        if (hp.n_expert > 0) { err = "mixture-of-experts encoder files are not supported"; status = -102; return nullptr; }
        want("token_types.weight", m->tok_types, false);
        want("token_embd_norm.weight", m->tok_norm, true);
        want("token_embd_norm.bias", m->tok_norm_b, true);
        m->layers.resize((size_t)hp.n_layer);
        for (int il = 0; il < hp.n_layer && !fail; il++) {
            LayerWeights &L = m->layers[(size_t)il];
            const std::string p = "blk." + std::to_string(il) + ".";
            want(p + "attn_qkv.weight", L.wqkv, true);
            want(p + "attn_output.weight", L.wo, true);
            want(p + "attn_output.bias", L.bo, false);
            want(p + "attn_output_norm.weight", L.attn_out_norm, true);
            want(p + "attn_output_norm.bias", L.attn_out_norm_b, true);
            want(p + "ffn_gate.weight", L.gate, true);
            want(p + "ffn_up.weight", L.up, true);
            want(p + "ffn_down.weight", L.down, true);
            want(p + "layer_output_norm.weight", L.layer_out_norm, true);
            want(p + "layer_output_norm.bias", L.layer_out_norm_b, true);
        }
    } else {
    want("output_norm.weight", m->out_norm, true);
    {   // the output projection is cut by vocabulary rows when they divide evenly (logits are gathered), else every rank keeps it whole
        const GGUFTensorInfo *ot = f.tensor("output.weight");
        want("output.weight", m->output, false, ot && ot->ne[1] % P == 0 ? SPLIT_ROWS : SPLIT_NONE);
    }
    want("rope_freqs.weight", m->rope_freqs, false);
    m->layers.resize((size_t)hp.n_layer);
    for (int il = 0; il < hp.n_layer && !fail; il++) {
        LayerWeights &L = m->layers[(size_t)il];
        const std::string p = "blk." + std::to_string(il) + ".";
        want(p + "attn_norm.weight", L.attn_norm, true);
        want(p + "attn_q.weight", L.wq, true, SPLIT_ROWS);
        want(p + "attn_k.weight", L.wk, true, SPLIT_ROWS);
        want(p + "attn_v.weight", L.wv, true, SPLIT_ROWS);
        want(p + "attn_output.weight", L.wo, true, SPLIT_COLS);
        want(p + "attn_q.bias", L.bq, false, SPLIT_ROWS);
        want(p + "attn_k.bias", L.bk, false, SPLIT_ROWS);
        want(p + "attn_v.bias", L.bv, false, SPLIT_ROWS);
        want(p + "ffn_norm.weight", L.ffn_norm, true);
        if (hp.n_expert > 0) {
            want(p + "ffn_gate_inp.weight", L.gate_inp, true);
            want(p + "ffn_gate_exps.weight", L.gate_exps, true);
            want(p + "ffn_up_exps.weight", L.up_exps, true);
            want(p + "ffn_down_exps.weight", L.down_exps, true);
        } else {
            want(p + "ffn_gate.weight", L.gate, true, SPLIT_ROWS);
            want(p + "ffn_up.weight", L.up, true, SPLIT_ROWS);
            want(p + "ffn_down.weight", L.down, true, SPLIT_COLS);
            // a contraction length without a weight-stream form whose half has one (mmvq_stream_applicable: 1, 2, 3, 4, 6, 7 or 10 passes of 2048):
            // Llama-3-70B's 28672 -> 2 x 14336
            static const bool halves_on = !(getenv("MI355_DOWN_HALVES") && getenv("MI355_DOWN_HALVES")[0] == '0');
            auto kb_ok = [](int64_t K) { const int64_t kb = (K + 2047) >> 11; return kb == 1 || kb == 2 || kb == 3 || kb == 4 || kb == 6 || kb == 7 || kb == 10; };
            const int64_t Kd = L.down.K;
            if (halves_on && !fail && (L.down.type == T_Q4_K || L.down.type == T_Q5_K || L.down.type == T_Q6_K) && !kb_ok(Kd) && Kd % 512 == 0 && kb_ok(Kd / 2)) {
                want(p + "ffn_down.weight", L.down_lo, true, SPLIT_COLS, 0, 2);
                want(p + "ffn_down.weight", L.down_hi, true, SPLIT_COLS, 1, 2);
            }
        }
    }
    }
    if (fail) { status = -102; return nullptr; }
    // ---- every tensor against the shape the hyper-parameters imply (per rank under a row split).  The activation buffers
    // are sized from the hyper-parameters and the kernels write one value per weight ROW: a file whose tensors disagree
    // with its own metadata must fail here, not write out of bounds at the first decode (upstream create_tensor does the
    // same).  Per-rank sizes: hp.n_head / n_head_kv are already this rank's.
    {
        const int64_t E = hp.n_embd, D = hp.head_dim, QW = (int64_t)hp.n_head * D, KVW = (int64_t)hp.n_head_kv * D;
        auto shape = [&](const DevTensor &t, int64_t K, int64_t N, int64_t NE, bool vec) {
            if (fail || t.name.empty()) return;                    // (absent optional tensor)
            const bool ok = vec ? (t.K == K && t.N == 1 && t.n_expert == 1) : (t.K == K && t.N == N && t.n_expert == NE);
            if (!ok) {
                err = "tensor " + t.name + " has shape [" + std::to_string(t.K) + ", " + std::to_string(t.N) + ", " + std::to_string(t.n_expert) + "], expected [" +
                      std::to_string(K) + (vec ? "]" : ", " + std::to_string(N) + ", " + std::to_string(NE) + "]");
                fail = true;
            }
        };
        if (m->tok_embd.K != E || m->tok_embd.N <= 0 || m->tok_embd.n_expert != 1) { err = "token_embd.weight does not have embedding_length columns"; fail = true; }
        const int64_t V = m->tok_embd.N;
        if (hp.encoder) {
            if (!m->tok_types.name.empty() && (m->tok_types.K != E || m->tok_types.N < 1 || m->tok_types.type != T_F32)) { err = "token_types.weight must hold f32 rows of embedding_length"; fail = true; }
            shape(m->tok_norm, E, 0, 0, true); shape(m->tok_norm_b, E, 0, 0, true);
            if (hp.n_rot != D) { err = "encoder files rotate whole heads (rope.dimension_count must equal the head size)"; fail = true; }
            int64_t FFe = 0;
            for (int il = 0; il < hp.n_layer && !fail; il++) {
                const LayerWeights &L = m->layers[(size_t)il];
                if (il == 0) FFe = L.gate.N;
                shape(L.wqkv, E, QW + 2 * KVW, 1, false); shape(L.wo, QW, E, 1, false); shape(L.bo, E, 0, 0, true);
                shape(L.attn_out_norm, E, 0, 0, true); shape(L.attn_out_norm_b, E, 0, 0, true);
                shape(L.layer_out_norm, E, 0, 0, true); shape(L.layer_out_norm_b, E, 0, 0, true);
                shape(L.gate, E, FFe, 1, false); shape(L.up, E, FFe, 1, false); shape(L.down, FFe, E, 1, false);
                if (!fail && (FFe <= 0 || (hp.n_ff_full > 0 && FFe != hp.n_ff_full))) { err = "feed-forward tensors do not match feed_forward_length"; fail = true; }
                for (const DevTensor *t : {&L.bo, &L.attn_out_norm, &L.attn_out_norm_b, &L.layer_out_norm, &L.layer_out_norm_b})
                    if (!fail && !t->name.empty() && t->type != T_F32) { err = "tensor " + t->name + " must be f32"; fail = true; }
            }
            if (!fail && (m->tok_norm.type != T_F32 || m->tok_norm_b.type != T_F32)) { err = "token_embd_norm must be f32"; fail = true; }
            if (fail) { status = -102; return nullptr; }
        } else {
        shape(m->out_norm, E, 0, 0, true);
        if (!m->output.name.empty()) shape(m->output, E, f.tensor("output.weight")->ne[1] % P == 0 ? V / P : V, 1, false);
        if (!m->rope_freqs.name.empty() && (m->rope_freqs.K != hp.n_rot / 2 || m->rope_freqs.type != T_F32)) { err = "rope_freqs.weight must hold rope.dimension_count / 2 f32 factors"; fail = true; }
        if (hp.n_rot <= 0 || hp.n_rot > D || (hp.n_rot & 1)) { err = "bad rope.dimension_count"; fail = true; }
        int64_t FF = 0;
        for (int il = 0; il < hp.n_layer && !fail; il++) {
            const LayerWeights &L = m->layers[(size_t)il];
            shape(L.attn_norm, E, 0, 0, true); shape(L.ffn_norm, E, 0, 0, true);
            shape(L.wq, E, QW, 1, false); shape(L.wk, E, KVW, 1, false); shape(L.wv, E, KVW, 1, false);
            shape(L.wo, QW, E, 1, false);
            shape(L.bq, QW, 0, 0, true); shape(L.bk, KVW, 0, 0, true); shape(L.bv, KVW, 0, 0, true);
            if (hp.n_expert > 0) {
                if (il == 0) FF = L.gate_exps.N;
                shape(L.gate_inp, E, hp.n_expert, 1, false);
                shape(L.gate_exps, E, FF, hp.n_expert, false); shape(L.up_exps, E, FF, hp.n_expert, false);
                shape(L.down_exps, FF, E, hp.n_expert, false);
            } else {
                if (il == 0) FF = L.gate.N;
                shape(L.gate, E, FF, 1, false); shape(L.up, E, FF, 1, false);
                shape(L.down, FF, E, 1, false);
            }
            if (!fail && (FF <= 0 || (hp.n_ff_full > 0 && FF * P != hp.n_ff_full))) { err = "feed-forward tensors do not match feed_forward_length"; fail = true; }
            // norms and biases are read as f32 vectors by the kernels
            for (const DevTensor *t : {&L.attn_norm, &L.ffn_norm, &L.bq, &L.bk, &L.bv})
                if (!fail && !t->name.empty() && t->type != T_F32) { err = "tensor " + t->name + " must be f32"; fail = true; }
        }
        if (!fail && m->out_norm.type != T_F32) { err = "output_norm.weight must be f32"; fail = true; }
        if (fail) { status = -102; return nullptr; }
        }
    }
    hp.n_vocab = (int)m->tok_embd.N;
    hp.n_ff = (int)(hp.n_expert ? m->layers[0].gate_exps.N : m->layers[0].gate.N);     // this rank's width under a row split
    if (!hp.n_ff_full) hp.n_ff_full = hp.n_ff * P;
    hp.n_vocab_local = !m->output.name.empty() ? (int)m->output.N : hp.n_vocab;      // (planned, not uploaded yet)

    // (the file has been validated without touching the device: a malformed file fails the same way with and without a GPU)
    if (hipSetDevice(main_gpu) != hipSuccess) { err = "hipSetDevice failed"; status = -100; return nullptr; }
    m->device = main_gpu;
    uint8_t *arena = nullptr, *stage = nullptr;
    if (hipMalloc(&arena, total) != hipSuccess) { err = "hipMalloc of " + std::to_string(total) + " weight bytes failed"; status = -104; return nullptr; }
    m->arenas.push_back(arena);
    if (max_stage && hipMalloc(&stage, max_stage) != hipSuccess) { err = "hipMalloc of staging buffer failed"; status = -104; return nullptr; }
    hipStream_t st = nullptr;
    (void)hipStreamCreate(&st);
    for (const Plan &pl : plan) {
        DevTensor &d = *pl.dst;
        d.data = arena + pl.off;
        const bool direct = !(d.type == T_Q6_K || d.type == T_Q8_0 || d.type == T_Q2_K || d.type == T_Q3_K || d.type == T_Q4_0 || d.type == T_Q5_0 || d.type == T_IQ4_NL) && (pl.ti->n_dims == 1 || d.row_bytes == ggml_row_bytes(d.type, d.K));
        hipError_t e;
        const uint8_t *src = (const uint8_t *)pl.ti->data + pl.src_off;
        uint8_t *to = direct ? d.data : stage;
        if (pl.src_width == pl.src_pitch) e = hipMemcpyAsync(to, src, pl.src_bytes, hipMemcpyHostToDevice, st);
        else e = hipMemcpy2DAsync(to, pl.src_width, src, pl.src_pitch, pl.src_width, (size_t)pl.src_rows, hipMemcpyHostToDevice, st);
        if (!direct && e == hipSuccess) e = launch_repack_rows(d.type, stage, d.data, d.K, d.N * d.n_expert, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            err = std::string("upload of ") + d.name + " failed: " + hipGetErrorString(e);
            status = -105;
            if (stage) (void)hipFree(stage);
            (void)hipStreamDestroy(st);
            return nullptr;
        }
        if (!pl.extra_copy) m->file_tensor_bytes += pl.src_bytes;
    }
    if (stage) (void)hipFree(stage);
    (void)hipStreamDestroy(st);
    if (!m->output.valid()) m->output = m->tok_embd;   // tied embeddings
    if (hp.encoder) {
        // Q, K and V are row ranges of the fused projection: views into its device rows (one launch each; the rows are contiguous per output)
        for (auto &L : m->layers) {
            const int64_t qw = (int64_t)hp.n_head * hp.head_dim, kvw = (int64_t)hp.n_head_kv * hp.head_dim;
            auto view = [&](DevTensor &v, int64_t row0, int64_t rows, const char *what) {
                v = L.wqkv;
                v.name = L.wqkv.name + "[" + what + "]";
                v.N = rows;
                v.data = L.wqkv.data + (size_t)row0 * L.wqkv.row_bytes;
                v.bytes = (size_t)rows * L.wqkv.row_bytes;
                v.planes = nullptr; v.planes_bytes = 0;
            };
            view(L.wq, 0, qw, "q"); view(L.wk, qw, kvw, "k"); view(L.wv, qw + kvw, kvw, "v");
        }
    }
    m->device_bytes = total;
    m->host_bytes = 0;

    // prompt-processing copy of the per-layer projection weights: both int8 MFMA operand planes of every K-step,
    // expanded once here (2 B per weight) so that the prefill kernel spends nothing per weight (mmq.hip).  288 GB of HBM
    // is what makes this the default; it is skipped (the prefill kernel then expands on the fly) when memory is short.
    if (const char *e = getenv("MI355_PREFILL_PLANES")) prefill_planes = atoi(e);
    if (prefill_planes != 0) {
        std::vector<DevTensor *> want;
        for (auto &L : m->layers)
            for (DevTensor *d : {&L.wq, &L.wk, &L.wv, &L.wo, &L.gate, &L.up, &L.down, &L.gate_exps, &L.up_exps, &L.down_exps})
                if (d->valid() && (mmq_planes_bytes(d->type, d->N, (int)d->K) || mmq_q80_copy_bytes(d->type, d->N, (int)d->K))) want.push_back(d);
        // (an *_exps tensor holds one plane set per expert, back to back: a prompt batch runs one contraction per expert)
        // (K-quants: the two int8 MFMA planes; Q4_0 / Q5_0 / IQ4_NL: an exact Q8_0-layout copy for the Q8_0 prompt kernel)
        auto planes_of = [](const DevTensor *d) {
            const size_t b = mmq_planes_bytes(d->type, d->N, (int)d->K);
            return ((b ? b : mmq_q80_copy_bytes(d->type, d->N, (int)d->K)) + 255) & ~(size_t)255;
        };
        size_t need = 0;
        for (DevTensor *d : want) need += planes_of(d) * (size_t)d->n_expert;
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        const size_t reserve = (size_t)24 << 30;             // KV cache, activations, other contexts
        const bool fits = need > 0 && free_b > need + reserve;
        if (need > 0 && !fits && prefill_planes == 1) { err = "not enough device memory for the prefill planes (" + std::to_string(need >> 20) + " MiB)"; status = -104; return nullptr; }
        if (fits) {
            uint8_t *parena = nullptr;
            if (hipMalloc(&parena, need) != hipSuccess) { err = "hipMalloc of the prefill planes failed"; status = -104; return nullptr; }
            m->arenas.push_back(parena);
            size_t off = 0;
            for (DevTensor *d : want) {
                d->planes = parena + off;
                d->planes_bytes = planes_of(d) * (size_t)d->n_expert;
                off += d->planes_bytes;
                for (int64_t x = 0; x < d->n_expert; x++) {
                    const uint8_t *src = d->data + (size_t)x * d->row_bytes * (size_t)d->N;
                    uint8_t *dst = d->planes + (size_t)x * planes_of(d);
                    const hipError_t e = mmq_planes_bytes(d->type, d->N, (int)d->K) ? launch_mmq_expand(d->type, src, d->row_bytes, (int)d->N, (int)d->K, dst, nullptr)
                                                                                   : launch_expand_q80_copy(d->type, src, d->row_bytes, (int)d->N, (int)d->K, dst, nullptr);
                    if (e != hipSuccess) { err = std::string("plane expansion of ") + d->name + " failed: " + hipGetErrorString(e); status = -105; return nullptr; }
                }
            }
            if (hipDeviceSynchronize() != hipSuccess) { err = "plane expansion failed"; status = -105; return nullptr; }
            m->planes_bytes = need;
            m->device_bytes += need;
        }
    }

    // algorithmic bytes per decoded token: each tensor once, one embedding row, used experts only
    uint64_t bpt = 0;
    for (const Plan &pl : plan) {
        if (pl.extra_copy) continue;                               // (the column halves of ffn_down: the same bytes a second time)
        const DevTensor &d = *pl.dst;
        uint64_t b = pl.src_bytes;
        if (&d == &m->tok_embd) b = ggml_row_bytes(d.type, d.K);
        else if (d.n_expert > 1 && hp.n_expert_used > 0) b = b / (uint64_t)d.n_expert * (uint64_t)hp.n_expert_used;
        bpt += b;
    }
    if (m->output.data == m->tok_embd.data) bpt += m->tok_embd.ggml_bytes;
    m->bytes_per_token = bpt;
    char desc[256];
    snprintf(desc, sizeof desc, "%s %dL E%d H%d/%d FF%d V%d%s", hp.arch.c_str(), hp.n_layer, hp.n_embd, hp.n_head, hp.n_head_kv, hp.n_ff,
             hp.n_vocab, hp.n_expert ? " MoE" : "");
    m->desc = desc;
    return m.release();
}

// ------------------------------------------------------------------------------------------ context
Context::Context(Model *m, const ContextParams &p) : model(m), cp(p) {}

// the dispatches' own begin / end timestamps, by role (kernels.h KernelTimer): events are created on demand and reused from step to step
struct Context::KTimer : KernelTimer {
    struct Rec { const char *role; hipEvent_t a, b; };
    std::vector<Rec> pool;
    size_t used = 0;
    bool next(const char *role, hipEvent_t *start, hipEvent_t *stop) override {
        if (used == pool.size()) {
            Rec r{role, nullptr, nullptr};
            if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
            pool.push_back(r);
        }
        Rec &r = pool[used++];
        r.role = role; *start = r.a; *stop = r.b;
        return true;
    }
    ~KTimer() override { for (auto &r : pool) { if (r.a) (void)hipEventDestroy(r.a); if (r.b) (void)hipEventDestroy(r.b); } }
};
Context::~Context() {
    if (holds_fused_) { (void)hipStreamSynchronize(stream_); fused_release(); }
    if (ktimer_) { if (kernel_timer() == ktimer_) set_kernel_timer(nullptr); delete ktimer_; ktimer_ = nullptr; }
    attn_probe_report();
    attn_out_probe_report();
    if (d_engine_probe_) {                                     // diagnosis: time line of the probed layer's last engine launch
        const int ncu = num_cu(), NW = 10, NS = 48;
        std::vector<unsigned long long> t((size_t)ncu * NW * NS);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(t.data(), d_engine_probe_, t.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (const char *pf = getenv("MI355_ENGINE_PROBE_FILE")) {          // raw stamps [workgroup][wave][32] for tools/engine_probe.py
                if (FILE *f = fopen(pf, "wb")) { fwrite(t.data(), 8, t.size(), f); fclose(f); }
            }
            unsigned long long base = ~0ull;
            for (int w = 0; w < ncu * NW; w++) { const unsigned long long v = t[(size_t)w * NS]; if (v && v < base) base = v; }
            // kind 0: wall-clock stamp relative to the first wave; 1: a duration (10 ns ticks); 2: a count
            auto stat = [&](bool loader, int idx, const char *name, int kind = 0) {
                std::vector<double> v;
                for (int w = 0; w < ncu * NW; w++) {
                    if (loader != ((w % NW) < 2)) continue;
                    const unsigned long long x = t[(size_t)w * NS + idx];
                    if (kind == 0) { if (x) v.push_back((double)(long long)(x - base) * 0.01); }
                    else v.push_back(kind == 1 ? (double)x * 0.01 : (double)x);
                }
                if (v.empty()) return;
                std::sort(v.begin(), v.end());
                fprintf(stderr, "  %-52s n=%4zu  min %8.2f  med %8.2f  max %8.2f %s\n", name, v.size(), v.front(), v[v.size() / 2], v.back(), kind == 2 ? "" : "us");
            };
            fprintf(stderr, "engine probe, layer %d (us since the first wave entered):\n", engine_probe_layer_);
            stat(true, 0, "loader: enter"); stat(true, 1, "loader: go (consumers' requests queued)"); stat(true, 2, "loader: attn_output issued");
            stat(true, 3, "loader: gate|up issued"); stat(true, 4, "loader: down issued"); stat(true, 5, "loader: qkv issued"); stat(true, 6, "loader: all landed");
            stat(true, 12, "loader: attn_output polls waiting for landings", 2); stat(true, 13, "loader: attn_output polls waiting for ring space", 2);
            stat(true, 14, "loader: gate|up polls waiting for landings", 2); stat(true, 15, "loader: gate|up polls waiting for ring space", 2);
            stat(true, 16, "loader: down polls waiting for landings", 2); stat(true, 17, "loader: down polls waiting for ring space", 2);
            stat(true, 18, "loader: qkv polls waiting for landings", 2); stat(true, 19, "loader: qkv polls waiting for ring space", 2);
            stat(false, 0, "consumer: enter"); stat(false, 20, "consumer: attn_output activation ready"); stat(false, 21, "consumer: attn_output waiting for slots", 1);
            stat(false, 22, "consumer: attn_output decoding", 1); stat(false, 1, "consumer: attn_output decoded");
            stat(false, 32, "consumer: next described, norm weights requested");
            stat(false, 2, "consumer: x' gather starts"); stat(false, 3, "consumer: x' in LDS");
            stat(false, 34, "consumer: sum of squares done"); stat(false, 35, "consumer: rendezvous 1 passed"); stat(false, 36, "consumer: scale known");
            stat(false, 37, "consumer: blocks quantised");
            stat(false, 23, "consumer: gate|up activation ready"); stat(false, 24, "consumer: gate|up waiting for slots", 1); stat(false, 25, "consumer: gate|up decoding", 1);
            stat(false, 4, "consumer: gate|up decoded"); stat(false, 5, "last arriver: swiglu hand-over starts"); stat(false, 6, "last arriver: own blocks quantised + published");
            stat(false, 7, "consumer: codes in LDS"); stat(false, 26, "consumer: down activation ready"); stat(false, 27, "consumer: down waiting for slots", 1);
            stat(false, 28, "consumer: down decoding", 1); stat(false, 8, "consumer: down decoded"); stat(false, 9, "consumer: x'' gather starts");
            stat(false, 10, "consumer: x'' in LDS"); stat(false, 29, "consumer: qkv activation ready"); stat(false, 30, "consumer: qkv waiting for slots", 1);
            stat(false, 31, "consumer: qkv decoding", 1); stat(false, 11, "consumer: qkv decoded");
        }
    }
    if (d_mega_probe_) {                                       // diagnosis: where the last whole-step launch spent its time
        const int nl = model->hp.n_layer, np = 1 + MEGA_PROBES_PER_LAYER * nl;
        std::vector<unsigned long long> t((size_t)np);
        (void)hipDeviceSynchronize();
        if (hipMemcpy(t.data(), d_mega_probe_, (size_t)np * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            const char *names[MEGA_PROBES_PER_LAYER] = {"wait", "qkv", "wait", "attention", "wait", "wo", "wait", "gate_up", "wait", "down"};
            double sum[MEGA_PROBES_PER_LAYER] = {0};
            for (int il = 0; il < nl; il++)
                for (int k = 0; k < MEGA_PROBES_PER_LAYER; k++)
                    sum[k] += (double)(t[(size_t)(1 + il * MEGA_PROBES_PER_LAYER + k)] - t[(size_t)(il * MEGA_PROBES_PER_LAYER + k)]) * 0.01;
            fprintf(stderr, "mega probe (workgroup 0, us per layer):");
            for (int k = 0; k < MEGA_PROBES_PER_LAYER; k++) fprintf(stderr, " %s %.2f", names[k], sum[k] / nl);
            fprintf(stderr, " | total %.1f us\n", (double)(t[(size_t)np - 1] - t[0]) * 0.01);
        }
    }
    for (auto &ge : graphs_) (void)hipGraphExecDestroy(ge.second);
    if (stage_event_) (void)hipEventDestroy(stage_event_);
    for (auto &pe : prof_events_) (void)hipEventDestroy(pe.second);
    for (void *p : allocs_) (void)hipFree(p);
    if (h_stage_) (void)hipHostFree(h_stage_);
    if (h_logits_) (void)hipHostFree(h_logits_);
    if (h_argmax_) (void)hipHostFree(h_argmax_);
    if (h_embd_) (void)hipHostFree(h_embd_);
    if (h_chunks_) (void)hipHostFree(h_chunks_);
    if (h_mega_flag_) (void)hipHostFree(h_mega_flag_);
    if (h_topk_) (void)hipHostFree(h_topk_);
    if (h_topk_adj_) (void)hipHostFree(h_topk_adj_);
    if (d_topk_adj_) (void)hipFree(d_topk_adj_);
    if (topk_scratch_) { (void)hipFree(topk_scratch_); topk_scratch_ = nullptr; }
    if (h_moe_meta_) (void)hipHostFree(h_moe_meta_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void *Context::dalloc(size_t bytes) {
    void *p = nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, bytes);
    allocs_.push_back(p);
    device_bytes += bytes;
    return p;
}

static void alloc_actq(ActQuant &q, size_t K, size_t T, bool k, bool z, std::vector<void *> &track, uint64_t &tot, bool &ok) {
    auto al = [&](size_t b) -> void * {
        void *p = nullptr;
        b = (b + 255) & ~(size_t)255;
        if (hipMalloc(&p, b) != hipSuccess) { ok = false; return nullptr; }
        track.push_back(p);
        tot += b;
        return p;
    };
    if (k) {
        q.qs = (int8_t *)al(T * K);
        q.d = (float *)al(T * (K / 256) * 4);
        q.bsums = (int16_t *)al(T * (K / 16) * 2);
    }
    if (z) {
        q.qs0 = (int8_t *)al(T * K);
        q.d0 = (uint16_t *)al(T * (K / 32) * 2);
    }
}

bool Context::init(std::string &err) {
    const HParams &hp = model->hp;
    if (hipSetDevice(model->device) != hipSuccess) { err = "hipSetDevice failed"; return false; }
    if (cp.n_ubatch > cp.n_batch) cp.n_ubatch = cp.n_batch;
    if (const char *ng = getenv("MI355_NO_GRAPHS")) { if (ng[0] == '1') cp.use_graphs = false; }   // e.g. under rocprofv3
    if (cp.n_ubatch == 0 || cp.n_ctx == 0) { err = "n_ctx / n_ubatch must be > 0"; return false; }
    if (cp.n_seq_max > 64) { err = "n_seq_max > 64 unsupported"; return false; }
    auto kv_ok = [](int t) { return t == T_F16 || t == T_Q8_0 || t == T_Q4_0; };
    if (!kv_ok(cp.type_k) || !kv_ok(cp.type_v)) { err = "unsupported KV cache type"; return false; }
    if (hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, model->device) == hipSuccess) set_num_cu(prop.multiProcessorCount);
    err_epoch_seen_ = stream_error_epoch();
    if (unsigned *ew = stream_error_word()) { mmvq_stream_set_error_word(ew); decode_engine_set_error_word(ew); attn_out_set_error_word(ew); tp_p2p_set_error_word(ew); }   // (per device: the pointer lives in device globals)

    const size_t T = cp.n_ubatch, E = hp.n_embd, FF = hp.n_ff, G = hp.n_head_kv, D = hp.head_dim, NC = cp.n_ctx;
    cells_.assign(NC, KVCell());
    kv_.resize((size_t)hp.n_layer);
    auto plane_bytes = [&](int type, size_t &codes, size_t &scales) {
        if (type == T_F16) { codes = G * NC * D * 2; scales = 0; }
        else if (type == T_Q8_0) { codes = G * NC * D; scales = G * NC * (D / 32) * 2; }
        else { codes = G * NC * D / 2; scales = G * NC * (D / 32) * 2; }
    };
    for (int il = 0; il < hp.n_layer; il++) {
        size_t c, s;
        plane_bytes(cp.type_k, c, s);
        kv_[(size_t)il].k = (uint8_t *)dalloc(c);
        kv_[(size_t)il].kd = s ? (uint16_t *)dalloc(s) : nullptr;
        plane_bytes(cp.type_v, c, s);
        kv_[(size_t)il].v = (uint8_t *)dalloc(c);
        kv_[(size_t)il].vd = s ? (uint16_t *)dalloc(s) : nullptr;
        if (!kv_[(size_t)il].k || !kv_[(size_t)il].v) { err = "KV cache allocation failed"; return false; }
    }
    d_cell_pos_ = (int32_t *)dalloc(NC * 4);
    d_cell_seq_ = (uint64_t *)dalloc(NC * 8);
    d_delta_ = (int32_t *)dalloc(NC * 4);
    // token staging block: [nkv x4][tok][pos][seq][cell][outrow][seqmask]
    const size_t Tp = (T + 1) & ~(size_t)1;   // keeps the u64 seqmask array 8-byte aligned
    stage_bytes_ = 16 + Tp * 4 * 5 + Tp * 8;
    stage_bytes_ = (stage_bytes_ + 15) & ~(size_t)15;
    d_stage_ = (uint8_t *)dalloc(stage_bytes_);
    if (hipHostMalloc((void **)&h_stage_, stage_bytes_, hipHostMallocDefault) != hipSuccess) { err = "pinned alloc failed"; return false; }
    d_nkv_ = (int32_t *)d_stage_;
    d_tok_ = (int32_t *)(d_stage_ + 16);
    d_pos_ = d_tok_ + Tp;
    d_seq_ = d_pos_ + Tp;
    d_cell_ = d_seq_ + Tp;
    d_outrow_ = d_cell_ + Tp;
    d_seqmask_ = (uint64_t *)(d_outrow_ + Tp);

    x_ = (float *)dalloc(T * E * 4);
    xn_ = (float *)dalloc(T * E * 4);
    q_ = (float *)dalloc(T * E * 4);
    k_ = (float *)dalloc(T * G * D * 4);
    v_ = (float *)dalloc(T * G * D * 4);
    att_ = (float *)dalloc(T * E * 4);
    ffn_ = (float *)dalloc(T * FF * 4);
    ffn_u_ = (float *)dalloc(T * FF * 4);
    xo_ = (float *)dalloc(T * E * 4);
    if (hp.tp_exchange) {
        if (!tp_active() || tp_size() != hp.tp_size || tp_rank() != hp.tp_rank) { err = "model was loaded as rank " + std::to_string(hp.tp_rank) + " of " + std::to_string(hp.tp_size) + " but the process has no matching row-split group (mi355_tp_init)"; return false; }
        tp_part_ = (float *)dalloc(T * E * 4);
        if (tp_uses_host()) cp.use_graphs = false;     // the host transport drains the stream inside the step
        if (const char *tg = getenv("MI355_TP_GRAPHS")) { if (tg[0] == '0') cp.use_graphs = false; }
    }
    if (hp.n_expert > 0) {
        router_ = (float *)dalloc(T * hp.n_expert * 4);
        moe_ids_ = (int32_t *)dalloc(T * hp.n_expert_used * 4);
        moe_w_ = (float *)dalloc(T * hp.n_expert_used * 4);
        moe_out_ = (float *)dalloc((size_t)hp.n_expert_used * T * E * 4);
        if (T >= 8) {                                              // grouped-by-expert batches of a prompt (run_layers)
            const size_t GR = (size_t)hp.n_expert_used * T;
            bool okg = true;
            alloc_actq(aq_eg_, E, GR, true, true, allocs_, device_bytes, okg);
            alloc_actq(aq_ffg_, FF, GR, true, true, allocs_, device_bytes, okg);
            ffn_g_ = (float *)dalloc(GR * FF * 4);
            ffn_ug_ = (float *)dalloc(GR * FF * 4);
            y_g_ = (float *)dalloc(GR * E * 4);
            moe_meta_ = (int32_t *)dalloc((size_t)(2 * hp.n_expert + 1) * 4);
            moe_slot_ = (int32_t *)dalloc(GR * 4);
            moe_tok_ = (int32_t *)dalloc(GR * 4);
            if (!okg || !ffn_g_ || !ffn_ug_ || !y_g_ || !moe_tok_ ||
                hipHostMalloc((void **)&h_moe_meta_, (size_t)(2 * hp.n_expert + 1) * 4, hipHostMallocDefault) != hipSuccess) { err = "MoE batch buffers allocation failed"; return false; }
        }
    }
    bool ok = true;
    alloc_actq(aq_e_, E, T, true, true, allocs_, device_bytes, ok);
    alloc_actq(aq_o_, E, T, true, true, allocs_, device_bytes, ok);
    alloc_actq(aq_ff_, FF, T, true, true, allocs_, device_bytes, ok);
    const int prep_rows = (int)T * (hp.n_expert > 0 ? std::max(1, (int)hp.n_expert_used) : 1);   // (expert batches: every (token, rank) pair is a row)
    mmq_bh_ = (int8_t *)dalloc(mmq_prep_bytes((int)std::max(E, FF), prep_rows));
    mmq_bl_ = (int8_t *)dalloc(mmq_prep_bytes((int)std::max(E, FF), prep_rows));
    // partial sums of the K-split prompt contraction: only tensors with few rows split (Q | K | V, attention output, FFN down),
    // up to four ways; tensors that do not fit fall back to an unsplit kernel
    // A launch only splits while its 128 x 256 tiles number fewer than 3/4 of the CUs, and then into ceil(CUs / tiles) <= 4 parts (mmq.hip planes2_split):
    // n_split * tiles < CUs + tiles < 7/4 CUs, i.e. never more than 7/4 * CUs tiles' worth of partial sums whatever the model (58.7 MB on 256 CUs;
    // sized by the widest tensor instead it was 335 MB per context for Llama-3-70B at n_ubatch 2048)
    mmq_ws_.bytes = std::min((size_t)4 * T * std::max<size_t>(E, (size_t)(hp.n_head + 2 * hp.n_head_kv) * D) * sizeof(float),
                             (size_t)(num_cu() * 7 / 4) * 128 * 256 * sizeof(float));
    mmq_ws_.p = T >= 128 ? (float *)dalloc(mmq_ws_.bytes) : nullptr;
    if (!mmq_ws_.p) mmq_ws_.bytes = 0;
    if (!ok || !x_ || !ffn_u_) { err = "activation buffer allocation failed"; return false; }

    size_t ws = 0;
    for (int t = 1; t <= (int)T; t++) {
        const int sp = flash_attn_pick_splits(t, (int)G, (int)NC);
        ws = std::max(ws, flash_attn_workspace_floats(t, hp.n_head, (int)D, sp));
    }
    ws = std::max(ws, flash_attn_workspace_floats(std::min<int>((int)T, 64), hp.n_head, (int)D, flash_attn_decode_splits((int)NC)));
    for (int t = 32; t <= (int)T; t += 32)                       // key splits of a prompt batch (fewer tokens take more splits)
        ws = std::max(ws, flash_attn_workspace_floats(t, hp.n_head, (int)D, flash_attn_prefill_splits(t, hp.n_head, (int)G, (int)D, (int)NC)));
    att_part_ = (float *)dalloc(ws * 4);
    att_part_floats_ = ws;
    att_counters_ = (unsigned *)dalloc(64 * ATT_SYNC_STRIDE * sizeof(unsigned));
    if (!att_part_ || !att_counters_) { err = "attention workspace allocation failed"; return false; }
    if (hipMemset(att_counters_, 0, 64 * ATT_SYNC_STRIDE * sizeof(unsigned)) != hipSuccess) { err = "hipMemset failed"; return false; }
    d_step_serial_ = (unsigned *)dalloc(64);
    d_ao_flags_ = (unsigned *)dalloc((size_t)std::max(1, hp.n_layer) * 64 * ATT_SYNC_STRIDE * sizeof(unsigned));
    d_ao_gran_ = (unsigned long long *)dalloc(attn_out_granule_words((int)(hp.n_head * D)) * 8);
    d_qkv_gran_ = (unsigned long long *)dalloc(qkv_attn_granule_words((int)(hp.n_head * D), (int)(hp.n_head_kv * D)) * 8);
    if (!d_step_serial_ || !d_ao_flags_ || !d_ao_gran_ || !d_qkv_gran_) { err = "step serial / flag allocation failed"; return false; }
    d_argmax_ = (int32_t *)dalloc(T * 4);
    argmax_scratch_ = (float *)dalloc(T * 129 * 4);     // T ticket words at the head, then per row 64 part values and 64 part indices (zero-filled: dalloc)
    rope_cs_ = (float *)dalloc(T * (size_t)hp.n_rot * 4);
    chunk_stride_ = (int)((NC + 63) / 64);
    d_chunks_ = (int32_t *)dalloc((size_t)64 * (chunk_stride_ + 1) * 4);
    if (hipHostMalloc((void **)&h_chunks_, (size_t)64 * (chunk_stride_ + 1) * 4, hipHostMallocDefault) != hipSuccess) { err = "pinned alloc failed"; return false; }
    d_embd_ = (float *)dalloc(T * E * 4);
    if (hipHostMalloc((void **)&h_embd_, T * E * 4, hipHostMallocDefault) != hipSuccess) { err = "pinned alloc failed"; return false; }
    if (hipHostMalloc((void **)&h_argmax_, T * 4, hipHostMallocDefault) != hipSuccess) { err = "pinned alloc failed"; return false; }
    embeddings_enabled = cp.embeddings || model->hp.encoder;      // an encoder has nothing but embeddings to give
    if (model->hp.encoder) {
        d_pos_open_ = (int32_t *)dalloc(T * 4);
        std::vector<int32_t> open((size_t)T, 0x7fffffff);
        if (!d_pos_open_ || hipMemcpy(d_pos_open_, open.data(), T * 4, hipMemcpyHostToDevice) != hipSuccess) { err = "encoder position buffer allocation failed"; return false; }
    }
    kv_clear();
    // the zero-fills above ran on the null stream, which the context's non-blocking stream does not wait for: drain them
    // before any kernel can touch these buffers (a late fill would wipe live KV rows; seen under rocprofv3 --pmc)
    if (hipDeviceSynchronize() != hipSuccess) { err = "device synchronise failed"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------ KV cells
void Context::kv_clear() {
    for (auto &c : cells_) c = KVCell();
    head_ = 0;
    has_shift_ = false;
    meta_dirty_ = true;
    region_next_.clear();
}
bool Context::kv_seq_rm(int seq, int p0, int p1) {
    if (seq >= 64) return false;                                   // (the cell masks are 64 bits wide)
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    int new_head = (int)cells_.size();
    for (int i = 0; i < (int)cells_.size(); i++) {
        KVCell &c = cells_[(size_t)i];
        if (c.pos < p0 || c.pos >= p1) continue;
        if (seq < 0) c.seqs = 0;
        else if (c.seqs & (1ull << seq)) c.seqs &= ~(1ull << seq);
        else continue;
        if (!c.seqs) { c.pos = -1; c.delta = 0; if (i < new_head) new_head = i; }
    }
    if (new_head < (int)cells_.size() && new_head < head_) head_ = new_head;
    if (new_head < (int)cells_.size()) region_next_.clear();   // cells came free: the next allocation starts at its region's lowest free cell
    meta_dirty_ = true;
    return true;
}
void Context::kv_seq_cp(int src, int dst, int p0, int p1) {
    if (src == dst || src < 0 || dst < 0 || src >= 64 || dst >= 64) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    for (auto &c : cells_)
        if ((c.seqs & (1ull << src)) && c.pos >= p0 && c.pos < p1) c.seqs |= 1ull << dst;
    meta_dirty_ = true;
}
void Context::kv_seq_add(int seq, int p0, int p1, int delta) {
    if (seq < 0 || seq >= 64) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    if (p0 == p1 || delta == 0) return;
    for (auto &c : cells_) {
        if (!(c.seqs & (1ull << seq)) || c.pos < p0 || c.pos >= p1) continue;
        has_shift_ = true;
        c.pos += delta;
        c.delta += delta;
        if (c.pos < 0) { c.pos = -1; c.seqs = 0; c.delta = 0; region_next_.clear(); }
    }
    meta_dirty_ = true;
}
int Context::kv_used_cells() const {
    int n = 0;
    for (const auto &c : cells_) n += c.pos >= 0;
    return n;
}

int Context::find_slot(int n) {
    const int NC = (int)cells_.size();
    if (n > NC) return -1;
    int head = head_, tested = 0;
    while (true) {
        if (head + n > NC) { tested += NC - head; head = 0; if (tested >= NC) return -1; continue; }
        bool ok = true;
        for (int i = 0; i < n; i++)
            if (cells_[(size_t)(head + i)].pos >= 0) { ok = false; head += i + 1; tested += i + 1; break; }
        if (ok) return head;
        if (tested >= NC) return -1;
    }
}

bool Context::alloc_cells(int n, const uint64_t *seqmask, std::vector<int> &out) {
    out.assign((size_t)n, -1);
    const int NC = (int)cells_.size();
    const int NS = (int)cp.n_seq_max;
    if (NS <= 1) {                                              // one sequence: a contiguous run, as before
        const int slot = find_slot(n);
        if (slot < 0) return false;
        for (int i = 0; i < n; i++) out[(size_t)i] = slot + i;
        head_ = slot + n;
        if (head_ >= NC) head_ = 0;
        return true;
    }
    // every sequence owns a region of the cache, a whole number of 64-cell attention chunks where the cache allows it: a
    // sequence then sees the same chunk partition whichever region it lives in (results do not depend on the slot), and
    // the hint below always points at the region's lowest free cell, so allocation is a pure function of the cell table
    int R = std::max(1, NC / NS);
    if (R >= 64) R &= ~63;
    if ((int)region_next_.size() != NS) { region_next_.assign((size_t)NS, 0); for (int s = 0; s < NS; s++) region_next_[(size_t)s] = s * R; }
    std::vector<char> taken((size_t)NC, 0);                    // cells handed out within this call
    auto is_free = [&](int c) { return cells_[(size_t)c].pos < 0 && !taken[(size_t)c]; };
    for (int i = 0; i < n; i++) {
        int s = seqmask[i] ? __builtin_ctzll(seqmask[i]) : 0;
        if (s >= NS) s = NS - 1;
        const int lo = s * R, hi = (s == NS - 1) ? NC : lo + R;
        int c = -1;
        int start = region_next_[(size_t)s];
        if (start < lo || start >= hi) start = lo;
        for (int k = 0; k < hi - lo; k++) {                    // own region first, from where the last cell went
            int cand = start + k;
            if (cand >= hi) cand -= hi - lo;
            if (is_free(cand)) { c = cand; break; }
        }
        if (c < 0) for (int cand = 0; cand < NC; cand++) if (is_free(cand)) { c = cand; break; }   // region full: anywhere
        if (c < 0) return false;
        taken[(size_t)c] = 1;
        out[(size_t)i] = c;
        if (c >= lo && c < hi) region_next_[(size_t)s] = c + 1;
    }
    return true;
}

void Context::apply_k_shift() {
    const HParams &hp = model->hp;
    std::vector<int32_t> delta(cells_.size());
    for (size_t i = 0; i < cells_.size(); i++) { delta[i] = cells_[i].delta; cells_[i].delta = 0; }
    (void)hipMemcpyAsync(d_delta_, delta.data(), delta.size() * 4, hipMemcpyHostToDevice, stream_);
    (void)hipStreamSynchronize(stream_);
    RopeArgs ra = rope_args(*model, true);
    for (int il = 0; il < hp.n_layer; il++)
        (void)launch_k_shift(kv_[(size_t)il], cp.type_k, hp.n_head_kv, hp.head_dim, (int)cp.n_ctx, d_delta_, ra, stream_);
    has_shift_ = false;
}

// ------------------------------------------------------------------------------------------ profiling
void Context::prof_begin() {
    if (!profile_) return;
    for (auto &pe : prof_events_) (void)hipEventDestroy(pe.second);
    prof_events_.clear();
    if (!ktimer_) ktimer_ = new KTimer;
    ktimer_->used = 0;
    set_kernel_timer(ktimer_);
    prof_mark("begin");
}
void Context::prof_mark(const char *name) {
    if (!profile_) return;
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, stream_);
    prof_events_.emplace_back(name, ev);
}
void Context::prof_end() {
    if (!profile_) return;
    set_kernel_timer(nullptr);
    (void)hipStreamSynchronize(stream_);
    last_profile_.clear();
    if (ktimer_) {     // "k:<role>" = sum of the role's kernel durations in this step, "n:<role>" = its launches
        for (size_t i = 0; i < ktimer_->used; i++) {
            const KTimer::Rec &r = ktimer_->pool[i];
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
            const std::string kn = std::string("k:") + r.role, nn = std::string("n:") + r.role;
            bool fk = false, fn = false;
            for (auto &e : last_profile_) { if (e.name == kn) { e.us += ms * 1000.0f; fk = true; } else if (e.name == nn) { e.us += 1.0f; fn = true; } }
            if (!fk) last_profile_.push_back({kn, ms * 1000.0f});
            if (!fn) last_profile_.push_back({nn, 1.0f});
        }
    }
    for (size_t i = 1; i < prof_events_.size(); i++) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, prof_events_[i - 1].second, prof_events_[i].second);
        bool found = false;
        for (auto &e : last_profile_)
            if (e.name == prof_events_[i].first) { e.us += ms * 1000.0f; found = true; break; }
        if (!found) last_profile_.push_back({prof_events_[i].first, ms * 1000.0f});
    }
}

// ------------------------------------------------------------------------------------------ linear layers
static bool is_quant(int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K || t == T_Q8_0 || t == T_Q2_K || t == T_Q3_K || t == T_Q4_0 || t == T_Q5_0 || t == T_IQ4_NL; }

static MMVQSeg make_seg(const DevTensor &w, float *out, int ld_out, const float *resid, const int32_t *esel) {
    MMVQSeg s{};
    s.W = w.data; s.out = out; s.resid = resid; s.type = w.type; s.n_rows = (int)w.N; s.ld_out = ld_out; s.row_bytes = w.row_bytes;
    s.expert_sel = esel; s.expert_stride = w.row_bytes * (size_t)w.N;
    return s;
}

static void chunk_act(MMVQArgs &a, const ActQuant &aq, int K, int t0) {
    a.aq = aq.qs ? aq.qs + (size_t)t0 * K : nullptr;
    a.ad = aq.d ? aq.d + (size_t)t0 * (K / 256) : nullptr;
    a.abs = aq.bsums ? aq.bsums + (size_t)t0 * (K / 16) : nullptr;
    a.aq0 = aq.qs0 ? aq.qs0 + (size_t)t0 * K : nullptr;
    a.ad0 = aq.d0 ? aq.d0 + (size_t)t0 * (K / 32) : nullptr;
}

// up to 3 quantised weight tensors sharing one activation (fused Q/K/V), or one tensor with an epilogue
// (any whole number of super-blocks: the register-ring kernel's prologue stops at the row's end inside its last pass; the weight stream wants K % 1024 == 0
// and leaves the other hidden sizes - Qwen2-7B's 3584, Llama-30B's 6656 - to the register ring)
static bool can_fuse(int K, int T) { return T == 1 && K <= 8192 && (K & 255) == 0; }

static hipError_t mmvq_tokens(MMVQSeg *segs, int n_seg, int K, int T, int epi, const ActQuant &aq, hipStream_t st, const Fuse &fz = Fuse()) {
    for (int t0 = 0; t0 < T;) {
        const int rem = T - t0, nt = rem >= 16 ? 16 : rem >= 8 ? 8 : rem >= 4 ? 4 : rem >= 2 ? 2 : 1;
        MMVQArgs a{};
        a.n_seg = n_seg; a.K = K; a.T = nt; a.epi = epi;
        a.fuse_mode = fz.mode; a.nx = fz.x; a.nw = fz.w; a.neps = fz.eps;
        a.out_host = nt == 1 && T == 1 ? fz.out_host : nullptr;
        for (int s = 0; s < n_seg; s++) {
            a.seg[s] = segs[s];
            a.seg[s].out = segs[s].out + (size_t)t0 * segs[s].ld_out;
            if (segs[s].resid) a.seg[s].resid = segs[s].resid + (size_t)t0 * segs[s].ld_out;
        }
        chunk_act(a, aq, K, t0);
        hipError_t e = launch_mmvq(a, st);
        if (e != hipSuccess) return e;
        t0 += nt;
    }
    return hipSuccess;
}

hipError_t Context::ensure_prep(const ActQuant &aq, int K, int T) {
    if (prep_owner_ == aq.qs && prep_K_ == K && prep_T_ == T && aq.qs) return hipSuccess;
    HIP_TRY(launch_mmq_prep(aq, K, T, mmq_bh_, mmq_bl_, stream_));
    prep_written(aq, K, T);
    return hipSuccess;
}

// Q2_K / Q3_K tensors reach the matrix cores only through their plane sets (no expand-on-the-fly kernel): prompt batches of 32 tokens and more
static bool q80_copy(const DevTensor &w, int K, int T) {
    return (w.type == T_Q4_0 || w.type == T_Q5_0 || w.type == T_IQ4_NL) && w.planes && w.n_expert == 1 && mmq_q80_applicable(T_Q8_0, K, T);
}
static bool planes_small(const DevTensor &w, int K, int T) { return (w.type == T_Q2_K || w.type == T_Q3_K) && w.planes && T >= 32 && (K % 256) == 0; }

hipError_t Context::linear(const DevTensor &w, const ActQuant &aq, const float *x_f32, int K, int T, float *out, int ld_out,
                           const float *resid, int epi) {
    if (is_quant(w.type)) {
        if (mmq_q80_applicable(w.type, K, T) && pending_fuse_.mode == 0 && epi != EPI_SWIGLU && aq.qs0)   // prompt processing, Q8_0 weights
            return launch_mmq_q80(w.data, w.row_bytes, (int)w.N, K, T, aq, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
        // (bh_over_ / bl_over_: the caller prepared the block-sum planes of a larger batch that aq's rows are a slice of - the expert loop)
        const int8_t *bh = bh_over_ ? bh_over_ : mmq_bh_, *bl = bh_over_ ? bl_over_ : mmq_bl_;
        if (mmq_ksplit_preferred(w.type, (int)w.N, K, T, w.planes != nullptr) && pending_fuse_.mode == 0 && epi != EPI_SWIGLU) {   // batched decode steps, short prompts, expert batches: MFMA, K split
            if (w.type != T_Q6_K && !bh_over_) HIP_TRY(ensure_prep(aq, K, T));
            return launch_mmq_ksplit(w.type, w.data, w.row_bytes, (int)w.N, K, T, aq, bh, bl, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
        }
        if (mmq_applicable(w.type, K, T) && pending_fuse_.mode == 0 && epi != EPI_SWIGLU) {   // prompt processing: MFMA path
            if ((w.type != T_Q6_K || !w.planes) && !bh_over_) HIP_TRY(ensure_prep(aq, K, T));
            if (w.planes) return launch_mmq_planes(w.type, w.planes, (int)w.N, K, T, aq, bh, bl, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_, mmq_ws_);
            return launch_mmq(w.type, w.data, w.row_bytes, (int)w.N, K, T, aq, bh, bl, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
        }
        if (q80_copy(w, K, T) && pending_fuse_.mode == 0 && epi != EPI_SWIGLU && aq.qs0)      // prompt processing of Q4_0 / Q5_0 / IQ4_NL tensors: their exact Q8_0-layout copy
            return launch_mmq_q80(w.planes, dev_row_bytes(T_Q8_0, K), (int)w.N, K, T, aq, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
        if (planes_small(w, K, T) && pending_fuse_.mode == 0 && epi != EPI_SWIGLU && aq.qs) {
            // prompt processing of Q2_K / Q3_K tensors: their plane sets (expanded at load in the Q4_K / Q6_K plane formats, mmq.hip) on the same kernels
            if (w.type == T_Q2_K && !bh_over_) HIP_TRY(ensure_prep(aq, K, T));
            return launch_mmq_planes(w.type, w.planes, (int)w.N, K, T, aq, bh, bl, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_, mmq_ws_);
        }
        MMVQSeg s = make_seg(w, out, ld_out, resid, nullptr);
        return mmvq_tokens(&s, 1, K, T, epi, aq, stream_, pending_fuse_);
    }
    if (mmf16_applicable(w.type, (int)w.N, K, T, w.data, x_f32, out) && w.n_expert == 1 && (ld_out & 3) == 0)      // a batch against an f16 tensor: matrix cores
        return launch_mmf16(w.data, (int)w.N, K, x_f32, T, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
    return launch_mmv_float(w.type, w.data, (int)w.N, K, x_f32, T, out, ld_out, epi == EPI_ADD ? resid : nullptr, stream_);
}

hipError_t Context::linear_multi(const DevTensor *const *ws, float *const *outs, int n, const ActQuant &aq, const float *x_f32, int T) {
    bool all_q = true;
    for (int i = 0; i < n; i++) all_q &= is_quant(ws[i]->type);
    const int K = (int)ws[0]->K;
    bool all_mmq = true, all_ks = true;
    for (int i = 0; i < n; i++) { all_mmq &= mmq_applicable(ws[i]->type, K, T); all_ks &= mmq_ksplit_applicable(ws[i]->type, K, T); }
    if (all_ks && pending_fuse_.mode == 0 && n <= 3) {         // batched decode step: Q, K, V in one launch
        HIP_TRY(ensure_prep(aq, K, T));
        MMQSeg sg[3];
        for (int i = 0; i < n; i++) sg[i] = MMQSeg{ws[i]->data, ws[i]->row_bytes, (int)ws[i]->N, ws[i]->type, outs[i], (int)ws[i]->N, nullptr, 0};
        return launch_mmq_ksplit_multi(sg, n, K, T, aq, mmq_bh_, mmq_bl_, false, stream_);
    }
    if (all_mmq && pending_fuse_.mode == 0) {
        HIP_TRY(ensure_prep(aq, K, T));
        // Q | K | V (or Q | K) as one launch over the concatenated rows where their plane sets are adjacent in the arena and
        // of one plane format (Q4_K and Q5_K share it; the Q6_K attn_v of the "more bits" layers runs on its own)
        auto mins = [](int t) { return t != T_Q6_K; };
        auto adjacent = [&](int i) { return ws[i]->planes && ws[i - 1]->planes && ws[i]->planes == ws[i - 1]->planes + ws[i - 1]->planes_bytes &&
                                            (ws[i - 1]->N % 32) == 0 && ws[i]->n_expert == 1; };
        int nf = 1;
        while (nf < n && nf < 3 && adjacent(nf) && mins(ws[nf]->type) == mins(ws[0]->type)) nf++;
        // the "more bits" layers (Q6_K attn_v beside Q4_K / Q5_K attn_q, attn_k): still one launch where the 128 x 256 kernel takes it
        int nm = nf;
        while (nm < n && nm < 3 && adjacent(nm)) nm++;
        if (nm > nf) {
            int rows[3];
            for (int i = 0; i < nm; i++) rows[i] = (int)ws[i]->N;
            if (mmq_planes_mixed_ok(rows, nm, K, T, mmq_ws_)) nf = nm;
        }
        if (nf >= 2) {
            int rows[3], ldo[3], types[3];
            float *o[3];
            for (int i = 0; i < nf; i++) { rows[i] = (int)ws[i]->N; ldo[i] = (int)ws[i]->N; o[i] = outs[i]; types[i] = ws[i]->type; }
            HIP_TRY(launch_mmq_planes_multi(ws[0]->type, ws[0]->planes, rows, o, ldo, nf, K, T, aq, mmq_bh_, mmq_bl_, nullptr, stream_, mmq_ws_, types));
        } else nf = 0;
        for (int i = nf; i < n; i++) {
            if (ws[i]->planes) HIP_TRY(launch_mmq_planes(ws[i]->type, ws[i]->planes, (int)ws[i]->N, K, T, aq, mmq_bh_, mmq_bl_, outs[i], (int)ws[i]->N, nullptr, stream_, mmq_ws_));
            else HIP_TRY(launch_mmq(ws[i]->type, ws[i]->data, ws[i]->row_bytes, (int)ws[i]->N, K, T, aq, mmq_bh_, mmq_bl_, outs[i], (int)ws[i]->N, nullptr, stream_));
        }
        return hipSuccess;
    }
    bool any_mmq = false;
    for (int i = 0; i < n; i++) any_mmq |= mmq_applicable(ws[i]->type, K, T) || mmq_q80_applicable(ws[i]->type, K, T) || planes_small(*ws[i], K, T) || q80_copy(*ws[i], K, T);
    if (all_q && n <= 3 && !(any_mmq && pending_fuse_.mode == 0)) {
        MMVQSeg segs[3];
        for (int i = 0; i < n; i++) segs[i] = make_seg(*ws[i], outs[i], (int)ws[i]->N, nullptr, nullptr);
        return mmvq_tokens(segs, n, K, T, EPI_STORE, aq, stream_, pending_fuse_);
    }
    // a prompt batch with mixed types (8-expert files keep attn_k / attn_v in Q8_0): every tensor takes its own best path
    for (int i = 0; i < n; i++) {
        hipError_t e = linear(*ws[i], aq, x_f32, K, T, outs[i], (int)ws[i]->N, nullptr, EPI_STORE);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------ whole-step kernel
// Off by default: measured on MI355X (tools/bench_gridbar.hip) a device-wide barrier costs ~4 us where a
// launch boundary inside a graph costs ~2.7 us, so the whole-step kernel is ~10 % slower than one launch per operation.
// MI355_MEGA=1 or mi355_debug_set_option("decode_mega", 1) turns it on for contexts created afterwards.
// Batches of at least this many tokens run a mixture-of-experts feed-forward grouped by expert (ggml_mul_mat_id as one
// batched contraction per expert); fewer tokens loop over (token, expert) with the mat-vec.  Tests move it to compare both.
static int g_moe_group_min = getenv("MI355_MOE_GROUP_MIN") ? atoi(getenv("MI355_MOE_GROUP_MIN")) : 8;
void set_moe_group_min(int t) { g_moe_group_min = t < 1 ? 1 : t; }
static int g_decode_mega = -1;           // -1: take the environment
static bool g_store_fuse = true;
void set_attn_store_fuse(bool on) { g_store_fuse = on; }
static bool store_fuse_enabled() {
    static const bool env_off = getenv("MI355_ATTN_STORE_FUSE") && getenv("MI355_ATTN_STORE_FUSE")[0] == '0';
    return g_store_fuse && !env_off;
}
static bool g_rope_fast = true;
void set_rope_fast(bool on) { g_rope_fast = on; }
void set_decode_mega(bool on) { g_decode_mega = on ? 1 : 0; }

// Builds the per-layer phase descriptors of decode_mega.hip: the same MMVQArgs the per-launch path passes to
// launch_mmvq_fast for Q/K/V, attn_output, gate/up and down of a single token, planned for the mega grid.
bool Context::mega_prepare() {
    if (mega_state_ != 0) return mega_state_ > 0;
    mega_state_ = -1;
    static const bool mega_env_on = getenv("MI355_MEGA") && getenv("MI355_MEGA")[0] == '1';
    const bool mega_env = g_decode_mega < 0 ? mega_env_on : g_decode_mega > 0;
    const HParams &hp = model->hp;
    const int E = hp.n_embd, FF = hp.n_ff, H = hp.n_head, G = hp.n_head_kv, D = hp.head_dim;
    if (!mega_env || hp.n_expert > 0 || G <= 0 || H % G != 0 || H * D != E) return false;
    const int kb_e = (E + 2047) >> 11, kb_ff = (FF + 2047) >> 11;
    if ((E % 2048) != 0 || (FF % 256) != 0 || !decode_mega_applicable(kb_e, kb_ff, H / G, cp.type_k, cp.type_v)) return false;
    RopeArgs ra = rope_args(*model, false);
    if (ra.neox || (ra.n_rot % 4) != 0 || D != 128 || !kv_store_fast_applicable(G, D, cp.type_k, cp.type_v, ra)) return false;
    auto kq = [](int t) { return t == T_Q4_K || t == T_Q5_K || t == T_Q6_K; };
    std::vector<MegaLayer> ml((size_t)hp.n_layer);
    const int blocks = mega_blocks();
    size_t lds = 0;
    for (int il = 0; il < hp.n_layer; il++) {
        const LayerWeights &L = model->layers[(size_t)il];
        if (!kq(L.wq.type) || !kq(L.wk.type) || !kq(L.wv.type) || !kq(L.wo.type) || !kq(L.gate.type) || !kq(L.up.type) || !kq(L.down.type)) return false;
        if (L.gate.type != L.up.type || L.gate.N != L.up.N || L.bq.valid() || L.bk.valid() || L.bv.valid()) return false;
        if ((int)L.wq.K != E || (int)L.wo.K != E || (int)L.gate.K != E || (int)L.down.K != FF) return false;
        MegaLayer &m = ml[(size_t)il];
        auto base = [&](MMVQArgs &a, int n_seg, int K, int epi, int fuse, const float *nx, const float *nw, const ActQuant &aq) {
            a = MMVQArgs{};
            a.n_seg = n_seg; a.K = K; a.T = 1; a.epi = epi;
            a.fuse_mode = fuse; a.nx = nx; a.nw = nw; a.neps = hp.eps;
            chunk_act(a, aq, K, 0);
        };
        base(m.qkv, 3, E, EPI_STORE, 1, x_, (const float *)L.attn_norm.data, aq_e_);
        m.qkv.seg[0] = make_seg(L.wq, q_, (int)L.wq.N, nullptr, nullptr);
        m.qkv.seg[1] = make_seg(L.wk, k_, (int)L.wk.N, nullptr, nullptr);
        m.qkv.seg[2] = make_seg(L.wv, v_, (int)L.wv.N, nullptr, nullptr);
        base(m.wo, 1, E, EPI_ADD, 0, nullptr, nullptr, aq_o_);
        m.wo.seg[0] = make_seg(L.wo, x_, E, x_, nullptr);
        base(m.gate_up, 2, E, EPI_SWIGLU, 1, x_, (const float *)L.ffn_norm.data, aq_e_);
        m.gate_up.seg[0] = make_seg(L.gate, ffn_, FF, nullptr, nullptr);
        m.gate_up.seg[1] = make_seg(L.up, ffn_u_, FF, nullptr, nullptr);
        base(m.down, 1, FF, EPI_ADD, 2, ffn_, nullptr, aq_ff_);
        m.down.seg[0] = make_seg(L.down, x_, E, x_, nullptr);
        for (MMVQArgs *a : {&m.qkv, &m.wo, &m.gate_up, &m.down}) {
            if (!mmvq_fast_applicable(*a)) return false;
            const size_t l = mmvq_fast_plan(*a, blocks, 4);
            if (!l) return false;
            lds = std::max(lds, l);
        }
        m.kv = kv_[(size_t)il];
    }
    d_mega_layers_ = (MegaLayer *)dalloc(ml.size() * sizeof(MegaLayer));
    // the barrier words are polled through the scalar cache path: they must never be cached in an XCD's L2
    if (hipExtMallocWithFlags((void **)&d_mega_sync_, MEGA_SYNC_WORDS * sizeof(unsigned), hipDeviceMallocUncached) != hipSuccess) { d_mega_sync_ = nullptr; return false; }
    allocs_.push_back(d_mega_sync_);
    if (!d_mega_layers_ || !d_mega_sync_) return false;
    if (!h_mega_flag_ && hipHostMalloc((void **)&h_mega_flag_, sizeof(int), hipHostMallocDefault) != hipSuccess) return false;
    *h_mega_flag_ = 0;
    if (hipMemcpy(d_mega_layers_, ml.data(), ml.size() * sizeof(MegaLayer), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemset(d_mega_sync_, 0, MEGA_SYNC_WORDS * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
    if (getenv("MI355_MEGA_PROBE") && getenv("MI355_MEGA_PROBE")[0] == '1')
        d_mega_probe_ = (unsigned long long *)dalloc((size_t)(1 + MEGA_PROBES_PER_LAYER * hp.n_layer) * 8);
    mega_lds_ = lds;
    mega_state_ = 1;
    return true;
}

bool Context::mega_check() {
    if (mega_state_ <= 0 || !h_mega_flag_ || *h_mega_flag_ == 0) return true;
    // a device-wide barrier gave up: the step's results are not valid.  Fall back to one launch per operation from here on.
    *h_mega_flag_ = 0;
    (void)hipMemset(d_mega_sync_, 0, MEGA_SYNC_WORDS * sizeof(unsigned));
    (void)hipDeviceSynchronize();
    mega_state_ = -1;
    for (auto &ge : graphs_) (void)hipGraphExecDestroy(ge.second);
    graphs_.clear();
    graph_exec_ = nullptr;
    last_error = "whole-step kernel: device-wide barrier timed out (workgroups not co-resident?); results of the step discarded";
    return false;
}

// ------------------------------------------------------------------------------------------ layer engine
// Opt-in (MI355_ENGINE=1 or mi355_debug_set_option("decode_engine", 1)): measured on MI355X the layer launch takes 42 us against 38.7 us for the four
// launches it replaces (545 vs 570 tok/s on Llama-3-8B Q4_K_M; DESIGN.md 4.8 has the per-phase time line and what bounds it), so one launch per mat-vec
// stays the default; the bitwise test turns the engine on explicitly.
static int g_decode_engine = -1;         // -1: take the environment
void set_decode_engine(int on) { g_decode_engine = on < 0 ? -1 : on ? 1 : 0; }

// the sticky error word of the weight-stream / engine kernels: one pinned word per process, raised by a bounded wait that gave up
static unsigned *stream_error_word() {
    static unsigned *w = [] { unsigned *p = nullptr; if (hipHostMalloc((void **)&p, 64, hipHostMallocDefault) != hipSuccess) p = nullptr; if (p) *p = 0u; return p; }();
    return w;
}

// The word is one per process (the kernels find it through a device global), but a process may hold several contexts (the engine's server_map_): the
// context that happens to read a raised word first is not necessarily the one whose kernel raised it.  So a raised word opens a new ERROR EPOCH, and every
// live context fails its next check once per epoch: the step that really timed out is never returned as valid, at the price of one discarded step in the
// bystanders.
static std::atomic<unsigned> g_stream_err_epoch{0}, g_stream_err_code{0};
// who can have raised it: every step launch of the process takes a serial; an epoch remembers the last serial handed out when it was opened; a context is a
// suspect if it has an unchecked launch from before that moment
static std::atomic<unsigned long long> g_launch_serial{0}, g_epoch_launch_serial{0};
unsigned long long next_launch_serial() { return g_launch_serial.fetch_add(1) + 1; }
unsigned stream_error_epoch() { return g_stream_err_epoch.load(); }
// tests (mi355_debug_set_option("raise_stream_error", code)): what a kernel's bounded wait does when it gives up
void debug_raise_stream_error(unsigned code) {
    unsigned *w = stream_error_word();
    if (w && code) __atomic_fetch_or(w, code, __ATOMIC_RELAXED);
}

static std::atomic<int> g_fused_owner_count[64];      // per device: 1 while a context holds the cross-workgroup-wait launches
static bool fused_gate_on() { static const bool off = getenv("MI355_FUSED_GATE") && getenv("MI355_FUSED_GATE")[0] == '0'; return !off; }
bool Context::fused_acquire() {
    if (!fused_gate_on() || holds_fused_) return true;
    int expected = 0;
    if (g_fused_owner_count[model->device & 63].compare_exchange_strong(expected, 1)) { holds_fused_ = true; return true; }
    return false;
}
void Context::fused_release() {
    if (holds_fused_) { holds_fused_ = false; g_fused_owner_count[model->device & 63].store(0); }
}

bool Context::stream_check() {
    fused_release();                                     // (every caller has just synchronised the context's stream)
    unsigned *w = stream_error_word();
    if (!w) return true;
    const unsigned raised = __atomic_exchange_n(w, 0u, __ATOMIC_RELAXED);
    if (raised) { g_stream_err_code.store(raised); g_epoch_launch_serial.store(g_launch_serial.load()); g_stream_err_epoch.fetch_add(1); }
    const unsigned epoch = g_stream_err_epoch.load();
    const unsigned long long first = first_unchecked_launch_;
    first_unchecked_launch_ = 0;                         // (every caller has just synchronised the context's stream: nothing of this context is in flight)
    if (epoch == err_epoch_seen_) return true;
    err_epoch_seen_ = epoch;
    // The word is raised by SOME kernel of the process.  A context with no unchecked launch from before the epoch was opened cannot be the one: it takes note
    // of the epoch and carries on - a time-out in one model's kernel then does not abort the requests of the other models a server holds, unless they were in
    // flight at the same moment (those discard one step each: the word cannot say whose kernel raised it).
    if (first == 0 || first > g_epoch_launch_serial.load()) return true;
    const unsigned code = g_stream_err_code.load();
    // a wait inside a kernel that waits for OTHER workgroups gave up (debugger, time-slicing, a workgroup that was not resident): the step's results are not
    // valid.  Two kernels wait that way - the layer engine and the one-launch attention + attn_output (attn_out.hip) - and both are left from here on: this
    // context takes one launch per mat-vec and the wait-free attention launch + linear(attn_output), so that a cause that persists (CU oversubscription from a
    // second stream, a debugger) cannot make every later step time out the same way.  The weight-stream kernels' own waits are inside one workgroup.
    engine_state_ = -1;
    attn_out_off_ = true;
    for (auto &ge : graphs_) (void)hipGraphExecDestroy(ge.second);
    graphs_.clear();
    graph_exec_ = nullptr;
    char buf[240];
    snprintf(buf, sizeof buf, "weight-stream / attention kernel: a bounded wait gave up in this process (code 0x%x); results of the step discarded, this context "
                              "continues on the wait-free launches", code);
    last_error = buf;
    return false;
}

// Builds the per-layer descriptors of decode_engine.hip: the same MMVQArgs the per-launch path passes to launch_mmvq_stream for attn_output, gate | up,
// down and the next layer's Q | K | V of a single token, planned for one workgroup per CU.
bool Context::engine_prepare() {
    if (engine_state_ != 0) return engine_state_ > 0;
    engine_state_ = -1;
    static const bool env_on = getenv("MI355_ENGINE") && getenv("MI355_ENGINE")[0] == '1';
    const bool on = g_decode_engine < 0 ? env_on : g_decode_engine > 0;
    const HParams &hp = model->hp;
    const int E = hp.n_embd, FF = hp.n_ff;
    if (!on || hp.n_expert > 0 || hp.tp_exchange || cp.n_ubatch < 1) return false;
    std::vector<EngineLayer> el((size_t)hp.n_layer);
    for (int il = 0; il < hp.n_layer; il++) {
        const LayerWeights &L = model->layers[(size_t)il];
        if (L.bq.valid() || L.bk.valid() || L.bv.valid()) return false;
        if (!is_quant(L.wo.type) || act_is_q80(L.wo.type)) return false;
        EngineLayer &m = el[(size_t)il];
        auto base = [&](MMVQArgs &a, int n_seg, int K, int epi, int fuse, const float *nx, const float *nw, const ActQuant &aq) {
            a = MMVQArgs{};
            a.n_seg = n_seg; a.K = K; a.T = 1; a.epi = epi;
            a.fuse_mode = fuse; a.nx = nx; a.nw = nw; a.neps = hp.eps;
            chunk_act(a, aq, K, 0);
        };
        base(m.wo, 1, E, EPI_ADD, 0, nullptr, nullptr, aq_o_);
        m.wo.seg[0] = make_seg(L.wo, x_, E, x_, nullptr);
        base(m.gu, 2, E, EPI_SWIGLU, 1, x_, (const float *)L.ffn_norm.data, aq_e_);
        m.gu.seg[0] = make_seg(L.gate, ffn_, FF, nullptr, nullptr);
        m.gu.seg[1] = make_seg(L.up, ffn_u_, FF, nullptr, nullptr);
        base(m.dn, 1, FF, EPI_ADD, 2, ffn_, nullptr, aq_ff_);
        m.dn.seg[0] = make_seg(L.down, x_, E, x_, nullptr);
        m.has_qkv = il + 1 < hp.n_layer ? 1 : 0;
        m.qkv = MMVQArgs{};
        if (m.has_qkv) {
            const LayerWeights &N = model->layers[(size_t)il + 1];
            base(m.qkv, 3, E, EPI_STORE, 1, x_, (const float *)N.attn_norm.data, aq_e_);
            m.qkv.seg[0] = make_seg(N.wq, q_, (int)N.wq.N, nullptr, nullptr);
            m.qkv.seg[1] = make_seg(N.wk, k_, (int)N.wk.N, nullptr, nullptr);
            m.qkv.seg[2] = make_seg(N.wv, v_, (int)N.wv.N, nullptr, nullptr);
        }
        if ((int)L.wo.K != E || (int)L.gate.K != E || (int)L.down.K != FF || (int)L.gate.N != FF || (int)L.up.N != FF) return false;
        if (!decode_engine_applicable(m, E, FF)) return false;
        decode_engine_plan(m);
    }
    unsigned *ew = stream_error_word();
    if (!ew) return false;
    mmvq_stream_set_error_word(ew);
    decode_engine_set_error_word(ew);
    d_engine_layers_ = (EngineLayer *)dalloc(el.size() * sizeof(EngineLayer));
    const size_t gw = decode_engine_granule_words(E, FF);
    d_engine_gran_ = (unsigned long long *)dalloc(gw * 8);
    d_engine_epoch_ = d_step_serial_;     // the step serial every set-up launch increments
    if (!d_engine_layers_ || !d_engine_gran_ || !d_engine_epoch_) return false;
    if (hipMemcpy(d_engine_layers_, el.data(), el.size() * sizeof(EngineLayer), hipMemcpyHostToDevice) != hipSuccess) return false;
    if (hipMemset(d_engine_gran_, 0, gw * 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return false;
    if (const char *pl = getenv("MI355_ENGINE_PROBE")) {
        engine_probe_layer_ = atoi(pl);
        const size_t np = (size_t)num_cu() * 10 * 48;
        d_engine_probe_ = (unsigned long long *)dalloc(np * 8);
        if (d_engine_probe_) (void)hipMemset(d_engine_probe_, 0, np * 8);
        (void)hipDeviceSynchronize();
    }
    engine_state_ = 1;
    return true;
}

// ------------------------------------------------------------------------------------------ the forward pass
// nomic-bert (llm_build_bert, the NOMIC_BERT branches): token + type-0 embeddings -> LayerNorm; per layer the fused Q | K | V projection, NEOX rope on Q and K,
// attention over ALL cells of the token's sequence (bidirectional: the attention kernels get INT_MAX as every token's position, so their causal test always
// passes), attn_output (+ bias), residual, LayerNorm, SwiGLU feed-forward, residual, LayerNorm.  The output is the last layer's hidden state; there is no head.
// K / V rows go through the context's cache like a prompt batch's (the whole sequence is one micro-batch: Context::decode checks that).
hipError_t Context::run_layers_encoder(int T, int n_kv_cap) {
    cur_T_ = T;
    const HParams &hp = model->hp;
    const int E = hp.n_embd, FF = hp.n_ff, H = hp.n_head, G = hp.n_head_kv, D = hp.head_dim;
    RopeArgs ra = rope_args(*model, true);
    const float kq_scale = 1.0f / sqrtf((float)D);
    const int n_kv_max = std::max(n_kv_cap, 1);
    att_splits_ = flash_attn_pick_splits(T, G, n_kv_max);
    last_layers_mega_ = false; last_layers_engine_ = false;
    HIP_TRY(launch_step_setup_embed(d_pos_, T, ra, rope_cs_, d_cell_pos_, d_cell_seq_, d_cell_, d_seqmask_, nullptr, nullptr, model->tok_embd.type, model->tok_embd.data, E,
                                    d_tok_, x_, stream_));
    if (model->tok_types.valid())      // token types are all zero: row 0 of the table on every token
        HIP_TRY(launch_add_qkv_bias(x_, nullptr, nullptr, (const float *)model->tok_types.data, nullptr, nullptr, E, 0, T, stream_));
    HIP_TRY(launch_layer_norm(x_, (const float *)model->tok_norm.data, (const float *)model->tok_norm_b.data, E, T, hp.eps, x_, stream_));
    prof_mark("embed");
    // quantised weight tensors contract against a quantised copy of their input (Q8_K for K-quants, Q8_0 otherwise), float tensors against the f32 rows
    auto quantise_for = [&](const float *x, int K, ActQuant &aq, std::initializer_list<const DevTensor *> ws) -> hipError_t {
        bool k = false, z = false;
        for (const DevTensor *w : ws) if (is_quant(w->type)) { if (act_is_q80(w->type)) z = true; else k = true; }
        if (!k && !z) return hipSuccess;
        prep_owner_ = nullptr;
        return launch_quantize(x, K, T, aq, k, z, stream_);
    };
    for (int il = 0; il < hp.n_layer; il++) {
        const LayerWeights &L = model->layers[(size_t)il];
        HIP_TRY(quantise_for(x_, E, aq_e_, {&L.wq, &L.wk, &L.wv}));
        const DevTensor *ws[3] = {&L.wq, &L.wk, &L.wv};
        float *outs[3] = {q_, k_, v_};
        HIP_TRY(linear_multi(ws, outs, 3, aq_e_, x_, T));
        prof_mark("qkv");
        HIP_TRY(launch_rope_kv_store(q_, k_, v_, T, H, G, D, d_pos_, d_cell_, ra, kv_[(size_t)il], cp.type_k, cp.type_v, (int)cp.n_ctx, rope_cs_, stream_));
        prof_mark("rope_kv");
        AttnArgs aa{};
        aa.q = q_; aa.out = att_; aa.kv = kv_[(size_t)il]; aa.type_k = cp.type_k; aa.type_v = cp.type_v;
        aa.T = T; aa.H = H; aa.G = G; aa.D = D; aa.n_ctx = (int)cp.n_ctx;
        aa.cell_pos = d_cell_pos_; aa.cell_seq = d_cell_seq_; aa.tok_pos = d_pos_open_; aa.tok_seq = d_seq_;
        aa.n_kv_dev = d_nkv_; aa.n_kv_max = n_kv_max; aa.scale = kq_scale; aa.part = att_part_;
        aa.out_q = nullptr; aa.out_q8k = false; aa.out_q80 = false;
        aa.splits = att_splits_;
        aa.pf_splits = flash_attn_prefill_splits(T, H, G, D, n_kv_max);
        while (aa.pf_splits > 1 && flash_attn_workspace_floats(T, H, D, aa.pf_splits) > att_part_floats_) aa.pf_splits >>= 1;
        HIP_TRY(launch_flash_attn(aa, stream_));
        prof_mark("attn");
        HIP_TRY(quantise_for(att_, H * D, aq_o_, {&L.wo}));
        HIP_TRY(linear(L.wo, aq_o_, att_, H * D, T, q_, E, nullptr, EPI_STORE));
        if (L.bo.valid()) HIP_TRY(launch_add_qkv_bias(q_, nullptr, nullptr, (const float *)L.bo.data, nullptr, nullptr, E, 0, T, stream_));
        HIP_TRY(launch_add(q_, x_, xn_, (int64_t)T * E, stream_));                       // re-add the layer input
        HIP_TRY(launch_layer_norm(xn_, (const float *)L.attn_out_norm.data, (const float *)L.attn_out_norm_b.data, E, T, hp.eps, xn_, stream_));
        prof_mark("attn_out");
        HIP_TRY(quantise_for(xn_, E, aq_e_, {&L.gate, &L.up}));
        HIP_TRY(linear(L.gate, aq_e_, xn_, E, T, ffn_, FF, nullptr, EPI_STORE));
        HIP_TRY(linear(L.up, aq_e_, xn_, E, T, ffn_u_, FF, nullptr, EPI_STORE));
        HIP_TRY(launch_swiglu(ffn_, ffn_u_, ffn_, (int64_t)T * FF, stream_));
        prof_mark("ffn_gate_up");
        HIP_TRY(quantise_for(ffn_, FF, aq_ff_, {&L.down}));
        HIP_TRY(linear(L.down, aq_ff_, ffn_, FF, T, q_, E, nullptr, EPI_STORE));
        HIP_TRY(launch_add(q_, xn_, x_, (int64_t)T * E, stream_));                       // the attention output bypasses the feed-forward block
        HIP_TRY(launch_layer_norm(x_, (const float *)L.layer_out_norm.data, (const float *)L.layer_out_norm_b.data, E, T, hp.eps, x_, stream_));
        prof_mark("ffn_down");
        if (debug_taps_ && dbg_) HIP_TRY(hipMemcpyAsync(dbg_ + (size_t)il * cp.n_ubatch * E, x_, (size_t)T * E * 4, hipMemcpyDeviceToDevice, stream_));
    }
    return hipSuccess;
}

hipError_t Context::run_layers(int T, int n_kv_cap) {
    if (model->hp.encoder) return run_layers_encoder(T, n_kv_cap);
    cur_T_ = T;
    const HParams &hp = model->hp;
    const int E = hp.n_embd, FF = hp.n_ff, H = hp.n_head, G = hp.n_head_kv, D = hp.head_dim;
    RopeArgs ra = rope_args(*model, true);
    const float kq_scale = 1.0f / sqrtf((float)D);
    const int n_kv_max = std::max(n_kv_cap, 1);   // upper bound of occupied cells the kernels are sized for
    att_splits_ = flash_attn_pick_splits(T, G, n_kv_max);

    // single-token step on a dense K-quant model: all layers in one launch (decode_mega.hip)
    bool mega = T == 1 && !profile_ && !debug_taps_ && !ub_embd_ && mega_prepare();
    AttnArgs ma{};
    if (mega) {
        ma.q = q_; ma.out = att_; ma.type_k = cp.type_k; ma.type_v = cp.type_v;
        ma.T = 1; ma.H = H; ma.G = G; ma.D = D; ma.n_ctx = (int)cp.n_ctx;
        ma.cell_pos = d_cell_pos_; ma.cell_seq = d_cell_seq_; ma.tok_pos = d_pos_; ma.tok_seq = d_seq_;
        ma.n_kv_dev = d_nkv_; ma.n_kv_max = n_kv_max; ma.scale = kq_scale; ma.part = att_part_;
        const DevTensor &wo0 = model->layers[0].wo;
        ma.out_q = &aq_o_; ma.out_q8k = !act_is_q80(wo0.type); ma.out_q80 = false;
        ma.splits = flash_attn_decode_splits(n_kv_max);
        if (chunk_lmax_ > 0) {
            ma.tok_chunks = d_chunks_; ma.tok_nchunks = d_chunks_ + (size_t)64 * chunk_stride_; ma.chunk_stride = chunk_stride_;
            ma.splits = std::max(chunk_cap_, chunk_lmax_);
        }
        mega = flash_attn_decode_fused_applicable(ma, ra);      // (more than 64 chunks: the per-launch path merges them)
    }
    // ... or one persistent launch per layer for the mat-vecs between two attention calls (decode_engine.hip)
    const bool engine = T == 1 && !mega && !profile_ && !debug_taps_ && !ub_embd_ && engine_prepare();
    last_layers_engine_ = engine;
    // cos / sin table, cell metadata and the tokens' embedding rows: one launch
    HIP_TRY(launch_step_setup_embed(d_pos_, T, ra, rope_cs_, d_cell_pos_, d_cell_seq_, d_cell_, d_seqmask_, mega ? d_mega_sync_ : nullptr,
                                    d_step_serial_, model->tok_embd.type, model->tok_embd.data, E, d_tok_, x_, stream_));
    // an embeddings batch: the caller's rows take the place of the looked-up ones (never inside a captured graph: decode_ubatch)
    if (ub_embd_) HIP_TRY(hipMemcpyAsync(x_, ub_embd_, (size_t)T * E * sizeof(float), hipMemcpyHostToDevice, stream_));
    prof_mark("embed");
    last_layers_mega_ = mega;
    if (mega)
        return launch_decode_mega(d_mega_layers_, hp.n_layer, (E + 2047) >> 11, (FF + 2047) >> 11, ma, rope_cs_, ra.n_rot, k_, v_, d_cell_,
                                  att_counters_, d_mega_sync_, h_mega_flag_, d_mega_probe_, mega_lds_, stream_);

    const bool tp = hp.tp_exchange;
    for (int il = 0; il < hp.n_layer; il++) {
        const LayerWeights &L = model->layers[(size_t)il];
        // --- attention block
        const bool any_f = !is_quant(L.wq.type) || !is_quant(L.wk.type) || !is_quant(L.wv.type);
        const bool need_k = is_quant(L.wq.type) && !act_is_q80(L.wq.type) || is_quant(L.wk.type) && !act_is_q80(L.wk.type) || is_quant(L.wv.type) && !act_is_q80(L.wv.type);
        const bool need_0 = act_is_q80(L.wq.type) || act_is_q80(L.wk.type) || act_is_q80(L.wv.type);
        const bool fuse_attn = !any_f && can_fuse(E, T);
        // the attention launch's description first: where attn_out.hip takes Q | K | V as well (round 6), no projection launch is made at all
        AttnArgs aa{};
        aa.q = q_; aa.out = att_; aa.kv = kv_[(size_t)il]; aa.type_k = cp.type_k; aa.type_v = cp.type_v;
        aa.T = T; aa.H = H; aa.G = G; aa.D = D; aa.n_ctx = (int)cp.n_ctx;
        aa.cell_pos = d_cell_pos_; aa.cell_seq = d_cell_seq_; aa.tok_pos = d_pos_; aa.tok_seq = d_seq_;
        aa.n_kv_dev = d_nkv_; aa.n_kv_max = n_kv_max; aa.scale = kq_scale; aa.part = att_part_;
        const bool o_q = is_quant(L.wo.type);
        aa.out_q = o_q ? &aq_o_ : nullptr; aa.out_q8k = !act_is_q80(L.wo.type); aa.out_q80 = act_is_q80(L.wo.type);   // merged + quantised in one pass
        const bool o_pl = o_q && aa.out_q8k && T >= 3;         // the batched kernels will want the block-sum planes: the merge writes them too
        if (o_pl) { aa.out_bh = mmq_bh_; aa.out_bl = mmq_bl_; }
        // (the single-launch step and the store-fused batched step write the new K / V rows themselves; only the third branch needs the fast store kernel)
        const bool fast_store = kv_store_fast_applicable(G, D, cp.type_k, cp.type_v, ra);
        static const int attn_mode = getenv("MI355_ATTN_MODE") ? atoi(getenv("MI355_ATTN_MODE")) : 2;
        bool fused_step = false, attn_out_done = false;
        // (parity mode for an f16 cache: the generic launch below dispatches to the cell-by-cell kernel with fp16 V accumulation)
        const bool v16 = fa_v_acc_f16_enabled() && cp.type_k == T_F16 && cp.type_v == T_F16;
        const bool decode_attn = !v16 && flash_attn_decode_applicable(aa, ra);
        if (decode_attn) {
            aa.splits = flash_attn_decode_splits(n_kv_max);
            if (chunk_lmax_ > 0) {                             // per-token chunk lists (decode_ubatch): batched steps, or regions in use
                aa.tok_chunks = d_chunks_; aa.tok_nchunks = d_chunks_ + (size_t)64 * chunk_stride_; aa.chunk_stride = chunk_stride_;
                aa.splits = std::max(chunk_cap_, chunk_lmax_);
            }
            fused_step = attn_mode > 0 && flash_attn_decode_fused_applicable(aa, ra);
        }
        // single-token step: attention + attn_output in one launch (attn_out.hip) where it has a form for the shape
        // (not where the ranks of a row split exchange through the host callback - the transport of a rig whose ranks SHARE one device: this kernel's
        // workgroups wait for each other (consumers for the item workgroups' flags), and two processes' copies placed on the same CUs at the same time can
        // hold each other's item workgroups out - every wait then runs into its bound (round 6: 0x8 on three of eight ranks behind one MI355X at the first
        // single-token step, profiles/r6_tp_shared_device_trace.txt).  A rank that owns its GPU has the chip to itself.)
        const bool ao_add = !tp || hp.tp_rank == 0;
        const MMVQSeg ao_seg = make_seg(L.wo, tp ? tp_part_ : x_, E, ao_add ? x_ : nullptr, nullptr);
        AttnArgs af = aa;
        bool ao_form = false;
        if (decode_attn && fused_step && attn_mode == 2 && !engine && il < 255 && !attn_out_off_ && !attn_out_skip_step_ && !(tp && tp_uses_host())) {
            af.splits = attn_out_fused_splits(af);
            if (chunk_lmax_ > 0) af.splits = aa.splits;
            ao_form = attn_out_fused_applicable(af, ra, ao_seg, (int)L.wo.K, ao_add ? EPI_ADD : EPI_STORE);
        }
        // ... and the layer's Q | K | V in front of it in that launch: the RMSNorm -> Q8_K prologue, the three mat-vecs, rope, KV store, attention, merge,
        // Q8_K and attn_output + residual are ONE launch per layer (outputs bit-identical to the two launches)
        bool qkv_in_attn = false;
        if (ao_form && fuse_attn && !tp && d_qkv_gran_ && !(L.bq.valid() || L.bk.valid() || L.bv.valid())) {
            QKVFuse qf{};
            qf.seg[0] = make_seg(L.wq, q_, (int)L.wq.N, nullptr, nullptr);
            qf.seg[1] = make_seg(L.wk, k_, (int)L.wk.N, nullptr, nullptr);
            qf.seg[2] = make_seg(L.wv, v_, (int)L.wv.N, nullptr, nullptr);
            qf.nx = x_; qf.nw = (const float *)L.attn_norm.data; qf.neps = hp.eps; qf.K = E; qf.gran = d_qkv_gran_;
            if (qkv_attn_out_applicable(af, ra, ao_seg, (int)L.wo.K, EPI_ADD, qf)) {
                HIP_TRY(launch_qkv_attn_out(af, rope_cs_, ra, d_cell_, att_counters_, d_ao_flags_ + (size_t)il * 64 * ATT_SYNC_STRIDE, d_ao_gran_, il, d_step_serial_, ao_seg,
                                            (int)L.wo.K, EPI_ADD, qf, stream_));
                qkv_in_attn = true; attn_out_done = true; qkv_attn_launches++;
                prof_mark("qkv");
            }
        }
        if (qkv_in_attn) {
            // nothing to launch
        } else if (engine && il > 0) {
            // Q | K | V of this layer were computed at the end of the previous layer's engine launch
        } else if (fuse_attn) {
            pending_fuse_.mode = 1; pending_fuse_.x = x_; pending_fuse_.w = (const float *)L.attn_norm.data; pending_fuse_.eps = hp.eps;
        } else {
            const bool pl = need_k && T >= 3;                  // the batched kernels will want the block-sum planes
            HIP_TRY(launch_rmsnorm_quant(x_, (const float *)L.attn_norm.data, E, T, hp.eps, any_f ? xn_ : nullptr, &aq_e_, need_k, need_0, stream_,
                                         pl ? mmq_bh_ : nullptr, pl ? mmq_bl_ : nullptr));
            prep_owner_ = nullptr;
            if (pl) prep_written(aq_e_, E, T);
            prof_mark("norm_quant");
        }
        const DevTensor *ws[3] = {&L.wq, &L.wk, &L.wv};
        float *outs[3] = {q_, k_, v_};
        if (!(engine && il > 0) && !qkv_in_attn) HIP_TRY(linear_multi(ws, outs, 3, aq_e_, xn_, T));
        pending_fuse_ = Fuse();
        if (!qkv_in_attn && (L.bq.valid() || L.bk.valid() || L.bv.valid()))       // qwen2-style attention biases: all T rows of the three projections in one launch
            HIP_TRY(launch_add_qkv_bias(q_, k_, v_, L.bq.valid() ? (const float *)L.bq.data : nullptr, L.bk.valid() ? (const float *)L.bk.data : nullptr,
                                        L.bv.valid() ? (const float *)L.bv.data : nullptr, H * D, G * D, T, stream_));
        if (!qkv_in_attn) prof_mark("qkv");
        if (qkv_in_attn) {
            // the whole block ran in the launch above
        } else if (decode_attn && (fast_store || fused_step || (batch_distinct_ && store_fuse_enabled()))) {
            if (fused_step) {
                // single-token step: K rope + KV store + attention + split merge + quantise in ONE launch - and, where attn_out.hip has a form for the shape,
                // the attn_output mat-vec with its residual add in that launch too
                if (ao_form) {
                    HIP_TRY(launch_attn_out_fused(af, rope_cs_, ra, k_, v_, d_cell_, att_counters_, d_ao_flags_ + (size_t)il * 64 * ATT_SYNC_STRIDE, d_ao_gran_, il, d_step_serial_, ao_seg,
                                                  (int)L.wo.K, ao_add ? EPI_ADD : EPI_STORE, stream_));
                    attn_out_done = true;
                }
                if (!attn_out_done)
                    HIP_TRY(launch_flash_attn_decode_fused(aa, rope_cs_, ra, k_, v_, d_cell_, attn_mode == 2 ? att_counters_ : nullptr, stream_));
            } else if (batch_distinct_ && store_fuse_enabled()) {
                // batched step, every token of a different sequence: K rope + KV store inside the attention launch (no token reads another's new cell)
                HIP_TRY(launch_flash_attn_decode(aa, rope_cs_, ra, stream_, k_, v_, d_cell_));
            } else {
                // small-batch step: K rope + KV store in one small kernel; q is rotated inside the attention kernel
                HIP_TRY(launch_kv_store_fast(k_, v_, T, G, D, rope_cs_, ra, d_cell_, kv_[(size_t)il], cp.type_k, cp.type_v, (int)cp.n_ctx, stream_));
                prof_mark("rope_kv");
                HIP_TRY(launch_flash_attn_decode(aa, rope_cs_, ra, stream_));
            }
        } else {
            static const bool rope_fast = !(getenv("MI355_ROPE_FAST") && getenv("MI355_ROPE_FAST")[0] == '0');
            if (rope_fast && g_rope_fast && rope_cs_ && rope_q_kv_store_fast_applicable(H, G, D, cp.type_k, cp.type_v, ra))
                HIP_TRY(launch_rope_q_kv_store_fast(q_, k_, v_, T, H, G, D, rope_cs_, ra, d_cell_, kv_[(size_t)il], cp.type_k, cp.type_v, (int)cp.n_ctx, stream_));
            else
                HIP_TRY(launch_rope_kv_store(q_, k_, v_, T, H, G, D, d_pos_, d_cell_, ra, kv_[(size_t)il], cp.type_k, cp.type_v, (int)cp.n_ctx, rope_cs_, stream_));
            prof_mark("rope_kv");
            aa.splits = att_splits_;
            // (splits balance the causal tiles of ONE long sequence: sized by what a query of this batch can see - its position + 1 -
            // not by the cache's high-water mark: 32 sequences of 50-token prompts in a 32000-cell cache need none, and every
            // extra workgroup would scan the whole cell table)
            aa.pf_splits = flash_attn_prefill_splits(T, H, G, D, std::min(n_kv_max, cur_max_pos_ + 1));
            while (aa.pf_splits > 1 && flash_attn_workspace_floats(T, H, D, aa.pf_splits) > att_part_floats_) aa.pf_splits >>= 1;
            HIP_TRY(launch_flash_attn(aa, stream_));
        }
        prof_mark("attn");
        if (o_pl) prep_written(aq_o_, (int)L.wo.K, T);          // the attention just re-quantised its output, planes included
        else if (prep_owner_ == aq_o_.qs) prep_owner_ = nullptr;
        if (engine) {   // attn_output, gate | up, down and the next layer's Q | K | V in one launch
            HIP_TRY(launch_decode_engine(d_engine_layers_ + il, E, FF, d_engine_gran_, d_engine_epoch_, il, il == engine_probe_layer_ ? d_engine_probe_ : nullptr, stream_));
            continue;
        }
        if (attn_out_done) {   // the mat-vec ran inside the attention launch (attn_out.hip)
            if (tp) HIP_TRY(tp_reduce_into_x(T));
        } else if (tp) {   // this rank's partial sum (rank 0 carries the residual), then the exchange: x = sum over ranks
            HIP_TRY(linear(L.wo, aq_o_, att_, (int)L.wo.K, T, tp_part_, E, hp.tp_rank == 0 ? x_ : nullptr, hp.tp_rank == 0 ? EPI_ADD : EPI_STORE));
            HIP_TRY(tp_reduce_into_x(T));
        } else {
            HIP_TRY(linear(L.wo, aq_o_, att_, (int)L.wo.K, T, x_, E, x_, EPI_ADD));
        }
        prof_mark("attn_out");

        // --- feed-forward block
        if (hp.n_expert > 0) {
            HIP_TRY(launch_rmsnorm_quant(x_, (const float *)L.ffn_norm.data, E, T, hp.eps, xn_, &aq_e_,
                                         !act_is_q80(L.gate_exps.type) || !act_is_q80(L.up_exps.type), act_is_q80(L.gate_exps.type) || act_is_q80(L.up_exps.type), stream_));
            prep_owner_ = nullptr;
            prof_mark("norm_quant");
            if (L.gate_inp.type == T_F32 || L.gate_inp.type == T_F16) {
                HIP_TRY(launch_moe_router(L.gate_inp.type, L.gate_inp.data, hp.n_expert, E, xn_, T, hp.n_expert_used, router_, moe_ids_, moe_w_, stream_,
                                          moe_forced_T_ == T ? d_moe_forced_ + (size_t)il * T * hp.n_expert_used : nullptr));
            } else {
                HIP_TRY(launch_mmv_float(L.gate_inp.type, L.gate_inp.data, hp.n_expert, E, xn_, T, router_, hp.n_expert, nullptr, stream_));
                HIP_TRY(launch_moe_route(router_, T, hp.n_expert, hp.n_expert_used, moe_ids_, moe_w_, stream_,
                                         moe_forced_T_ == T ? d_moe_forced_ + (size_t)il * T * hp.n_expert_used : nullptr));
            }
            prof_mark("moe_route");
            const int KU = hp.n_expert_used;
            const bool exps_q = is_quant(L.gate_exps.type) && is_quant(L.up_exps.type) && is_quant(L.down_exps.type);
            const int group_min = g_moe_group_min;
            if (exps_q && T >= group_min && aq_eg_.qs) {
                // ggml_mul_mat_id on a batch: group the (token, rank) pairs by expert, one contiguous activation batch per
                // expert (its weights are read once per batch, through the same batched kernels as the dense projections)
                const int NE = hp.n_expert, GR = T * KU;
                HIP_TRY(launch_moe_group(moe_ids_, T, KU, NE, moe_meta_, moe_slot_, moe_tok_, stream_));
                // round 5: every expert's batch in ONE launch per projection (the workgroups find their expert and token tile from the counts on the device):
                // no host synchronisation in the layer, and ~128-token batches that half-fill a launch each become one launch that fills the chip
                const bool grouped = T >= 32 && L.gate_exps.planes && L.up_exps.planes && L.down_exps.planes && L.gate_exps.type == L.up_exps.type &&
                                     L.gate_exps.N == L.up_exps.N && mmq_planes_moe_ok(L.gate_exps.type, (int)L.gate_exps.N, E) &&
                                     mmq_planes_moe_ok(L.down_exps.type, (int)L.down_exps.N, FF) && (L.gate_exps.N % 64) == 0;
                if (grouped) {
                    HIP_TRY(launch_moe_gather_act(aq_e_, moe_tok_, GR, E, aq_eg_, stream_));
                    prep_owner_ = nullptr;
                    const size_t ps_gu = L.gate_exps.planes_bytes / (size_t)L.gate_exps.n_expert, ps_d = L.down_exps.planes_bytes / (size_t)L.down_exps.n_expert;
                    HIP_TRY(launch_mmq_planes_swiglu_moe(L.gate_exps.type, L.gate_exps.planes, L.up_exps.planes, ps_gu, NE, moe_meta_, (int)L.gate_exps.N, E, GR, aq_eg_,
                                                         ffn_g_, FF, stream_));
                    HIP_TRY(launch_quantize(ffn_g_, FF, GR, aq_ffg_, true, false, stream_));
                    HIP_TRY(launch_mmq_planes_moe(L.down_exps.type, L.down_exps.planes, ps_d, NE, moe_meta_, (int)L.down_exps.N, FF, GR, aq_ffg_, y_g_, E, stream_));
                    HIP_TRY(launch_moe_scatter_combine(x_, y_g_, moe_w_, moe_slot_, T, E, KU, stream_));
                    prof_mark("moe_ffn");
                    if (debug_taps_ && dbg_) HIP_TRY(hipMemcpyAsync(dbg_ + (size_t)il * cp.n_ubatch * E, x_, (size_t)T * E * 4, hipMemcpyDeviceToDevice, stream_));
                    continue;
                }
                HIP_TRY(hipMemcpyAsync(h_moe_meta_, moe_meta_, (size_t)(2 * NE + 1) * 4, hipMemcpyDeviceToHost, stream_));
                HIP_TRY(launch_moe_gather_act(aq_e_, moe_tok_, GR, E, aq_eg_, stream_));
                prep_owner_ = nullptr;                             // the grouped rows were just rewritten
                HIP_TRY(hipStreamSynchronize(stream_));            // the batch sizes size the launches (prompt batches only: never inside a graph)
                auto view = [](const DevTensor &w, int e) {
                    DevTensor v = w;
                    v.data = w.data + (size_t)e * w.row_bytes * (size_t)w.N;
                    v.n_expert = 1;
                    v.planes = w.planes ? w.planes + (size_t)e * (w.planes_bytes / (size_t)w.n_expert) : nullptr;
                    return v;
                };
                auto rows = [](const ActQuant &q, int r0, int K) {
                    ActQuant v;
                    if (q.qs) { v.qs = q.qs + (size_t)r0 * K; v.d = q.d + (size_t)r0 * (K / 256); v.bsums = q.bsums + (size_t)r0 * (K / 16); }
                    if (q.qs0) { v.qs0 = q.qs0 + (size_t)r0 * K; v.d0 = q.d0 + (size_t)r0 * (K / 32); }
                    return v;
                };
                // the block-sum planes of ALL grouped rows in one launch (they were one launch per expert and projection: 16 a layer);
                // an expert's batch takes its slice of them
                const bool pl_gu = !act_is_q80(L.gate_exps.type) || !act_is_q80(L.up_exps.type), pl_d = !act_is_q80(L.down_exps.type);
                if (pl_gu) HIP_TRY(launch_mmq_prep(aq_eg_, E, GR, mmq_bh_, mmq_bl_, stream_));
                hipError_t e_exp = hipSuccess;
                for (int e = 0; e < NE && e_exp == hipSuccess; e++) {
                    const int n_e = h_moe_meta_[e], r0 = h_moe_meta_[NE + e];
                    if (n_e <= 0) continue;
                    if (pl_gu) { bh_over_ = mmq_bh_ + (size_t)r0 * (E >> 4); bl_over_ = mmq_bl_ + (size_t)r0 * (E >> 4); }
                    e_exp = linear(view(L.gate_exps, e), rows(aq_eg_, r0, E), nullptr, E, n_e, ffn_g_ + (size_t)r0 * FF, FF, nullptr, EPI_STORE);
                    if (e_exp == hipSuccess) e_exp = linear(view(L.up_exps, e), rows(aq_eg_, r0, E), nullptr, E, n_e, ffn_ug_ + (size_t)r0 * FF, FF, nullptr, EPI_STORE);
                }
                bh_over_ = bl_over_ = nullptr;
                HIP_TRY(e_exp);
                HIP_TRY(launch_swiglu_quant(ffn_g_, ffn_ug_, FF, GR, aq_ffg_, !act_is_q80(L.down_exps.type), act_is_q80(L.down_exps.type), stream_,
                                            pl_d ? mmq_bh_ : nullptr, pl_d ? mmq_bl_ : nullptr));
                prep_owner_ = nullptr;
                for (int e = 0; e < NE && e_exp == hipSuccess; e++) {
                    const int n_e = h_moe_meta_[e], r0 = h_moe_meta_[NE + e];
                    if (n_e <= 0) continue;
                    if (pl_d) { bh_over_ = mmq_bh_ + (size_t)r0 * (FF >> 4); bl_over_ = mmq_bl_ + (size_t)r0 * (FF >> 4); }
                    e_exp = linear(view(L.down_exps, e), rows(aq_ffg_, r0, FF), nullptr, FF, n_e, y_g_ + (size_t)r0 * E, E, nullptr, EPI_STORE);
                }
                bh_over_ = bl_over_ = nullptr;
                HIP_TRY(e_exp);
                HIP_TRY(launch_moe_scatter_combine(x_, y_g_, moe_w_, moe_slot_, T, E, KU, stream_));
                prof_mark("moe_ffn");
                if (debug_taps_ && dbg_) HIP_TRY(hipMemcpyAsync(dbg_ + (size_t)il * cp.n_ubatch * E, x_, (size_t)T * E * 4, hipMemcpyDeviceToDevice, stream_));
                continue;
            }
            // single-token step: the token's selected experts share one launch per projection (the workgroups are divided
            // among them; each reads its own expert index on the device): gate/up with SwiGLU, then down with the Q8_K
            // quantisation of its own expert's activation in the prologue
            bool experts_done = false;
            if (T == 1 && KU >= 2 && KU <= 8 && (int)cp.n_ubatch >= KU && L.gate_exps.type == L.up_exps.type) {
                MMVQArgs a{};
                a.n_seg = 2; a.K = E; a.T = 1; a.epi = EPI_SWIGLU; a.n_sel = KU; a.sel_out_stride = FF;
                a.seg[0] = make_seg(L.gate_exps, ffn_, FF, nullptr, moe_ids_);
                a.seg[1] = make_seg(L.up_exps, ffn_u_, FF, nullptr, moe_ids_);
                chunk_act(a, aq_e_, E, 0);
                // (RMSNorm + quantise again in the launch's prologue - the same arithmetic as the norm_quant launch that fed the router, so the same codes: the
                // weight stream's gate | up form takes its activation that way, and with it the selected experts stream like the dense feed-forward does)
                if (can_fuse(E, 1) && !act_is_q80(L.gate_exps.type)) { a.fuse_mode = 1; a.nx = x_; a.nw = (const float *)L.ffn_norm.data; a.neps = hp.eps; }
                MMVQArgs d{};
                d.n_seg = 1; d.K = FF; d.T = 1; d.epi = EPI_STORE; d.fuse_mode = 2; d.nx = ffn_; d.n_sel = KU; d.sel_nx_stride = FF; d.sel_out_stride = E;
                d.seg[0] = make_seg(L.down_exps, moe_out_, E, nullptr, moe_ids_);
                chunk_act(d, aq_ff_, FF, 0);
                if (mmvq_fast_applicable(a) && mmvq_fast_applicable(d)) {
                    HIP_TRY(launch_mmvq(a, stream_));          // the weight stream where it has a form for the shape, else the register ring (bit-identical)
                    HIP_TRY(launch_mmvq(d, stream_));
                    experts_done = true;
                }
            }
            for (int t = 0; t < T && !experts_done; t++) {
                for (int j = 0; j < KU; j++) {
                    const int32_t *esel = moe_ids_ + (size_t)t * KU + j;
                    MMVQArgs a{};
                    a.n_seg = 2; a.K = E; a.T = 1; a.epi = EPI_SWIGLU;
                    a.seg[0] = make_seg(L.gate_exps, ffn_ + (size_t)t * FF, FF, nullptr, esel);
                    a.seg[1] = make_seg(L.up_exps, ffn_u_ + (size_t)t * FF, FF, nullptr, esel);
                    chunk_act(a, aq_e_, E, t);
                    if (L.gate_exps.type == L.up_exps.type) {
                        HIP_TRY(launch_mmvq(a, stream_));
                    } else {
                        a.n_seg = 2; a.epi = EPI_STORE;
                        HIP_TRY(launch_mmvq(a, stream_));
                        HIP_TRY(launch_swiglu(ffn_ + (size_t)t * FF, ffn_u_ + (size_t)t * FF, ffn_ + (size_t)t * FF, FF, stream_));
                    }
                    // quantise inside the down projection's prologue where the persistent mat-vec takes this shape (as the
                    // dense feed-forward does), else as its own launch
                    const int kbf = (FF + 2047) / 2048;
                    const bool fuse_q = (L.down_exps.type == T_Q4_K || L.down_exps.type == T_Q5_K || L.down_exps.type == T_Q6_K) && (FF % 256) == 0 &&
                                        mmvq_fast_kb_ok(kbf);
                    if (!fuse_q) { HIP_TRY(launch_quantize(ffn_ + (size_t)t * FF, FF, 1, aq_ff_, !act_is_q80(L.down_exps.type), act_is_q80(L.down_exps.type), stream_)); prep_owner_ = nullptr; }
                    MMVQArgs d{};
                    d.n_seg = 1; d.K = FF; d.T = 1; d.epi = EPI_STORE;
                    if (fuse_q) { d.fuse_mode = 2; d.nx = ffn_ + (size_t)t * FF; }
                    d.seg[0] = make_seg(L.down_exps, moe_out_ + ((size_t)j * T + t) * E, E, nullptr, esel);
                    chunk_act(d, aq_ff_, FF, 0);
                    HIP_TRY(launch_mmvq(d, stream_));
                }
            }
            HIP_TRY(launch_moe_combine(x_, moe_out_, moe_w_, T, E, KU, (size_t)T * E, stream_));
            prof_mark("moe_ffn");
        } else {
            const bool gq = is_quant(L.gate.type), uq = is_quant(L.up.type);
            const bool fk = (gq && !act_is_q80(L.gate.type)) || (uq && !act_is_q80(L.up.type));
            const bool f0 = act_is_q80(L.gate.type) || act_is_q80(L.up.type);
            const bool fuse_ffn = gq && uq && L.gate.type == L.up.type && can_fuse(E, T);
            Fuse fz;
            if (fuse_ffn) {
                fz.mode = 1; fz.x = x_; fz.w = (const float *)L.ffn_norm.data; fz.eps = hp.eps;
            } else {
                const bool pl = fk && T >= 3;
                HIP_TRY(launch_rmsnorm_quant(x_, (const float *)L.ffn_norm.data, E, T, hp.eps, (!gq || !uq) ? xn_ : nullptr, &aq_e_, fk, f0, stream_,
                                             pl ? mmq_bh_ : nullptr, pl ? mmq_bl_ : nullptr));
                prep_owner_ = nullptr;
                if (pl) prep_written(aq_e_, E, T);
                prof_mark("norm_quant");
            }
            bool swiglu_quantised = false;
            const bool ffn_mmq = (mmq_q80_applicable(L.gate.type, E, T) && mmq_q80_applicable(L.up.type, E, T)) ||
                                 (mmq_applicable(L.gate.type, E, T) && mmq_applicable(L.up.type, E, T)) ||
                                 (mmq_ksplit_applicable(L.gate.type, E, T) && mmq_ksplit_applicable(L.up.type, E, T)) ||
                                 (planes_small(L.gate, E, T) && planes_small(L.up, E, T)) || (q80_copy(L.gate, E, T) && q80_copy(L.up, E, T));
            const bool ffn_ks = mmq_ksplit_preferred(L.gate.type, (int)L.gate.N, E, T, L.gate.planes != nullptr, true) &&
                                mmq_ksplit_preferred(L.up.type, (int)L.up.N, E, T, L.up.planes != nullptr, true) && !fuse_ffn;
            if (gq && uq && L.gate.type == L.up.type && !ffn_mmq) {
                MMVQSeg segs[2] = {make_seg(L.gate, ffn_, FF, nullptr, nullptr), make_seg(L.up, ffn_u_, FF, nullptr, nullptr)};
                HIP_TRY(mmvq_tokens(segs, 2, E, T, EPI_SWIGLU, aq_e_, stream_, fz));
            } else if (ffn_ks) {                               // batched decode step: gate and up in one launch
                if (L.gate.type != T_Q6_K || L.up.type != T_Q6_K) HIP_TRY(ensure_prep(aq_e_, E, T));
                MMQSeg sg[2] = {{L.gate.data, L.gate.row_bytes, (int)L.gate.N, L.gate.type, ffn_, FF, nullptr, 0},
                                {L.up.data, L.up.row_bytes, (int)L.up.N, L.up.type, ffn_u_, FF, nullptr, 0}};
                const bool pair = T <= 32 && L.gate.type == L.up.type && L.gate.N == L.up.N;     // SwiGLU in the epilogue
                HIP_TRY(launch_mmq_ksplit_multi(sg, 2, E, T, aq_e_, mmq_bh_, mmq_bl_, pair, stream_));
                if (!pair) {
                    if (is_quant(L.down.type) && (FF % 256) == 0) {
                        const bool pl = !act_is_q80(L.down.type);
                        HIP_TRY(launch_swiglu_quant(ffn_, ffn_u_, FF, T, aq_ff_, !act_is_q80(L.down.type), act_is_q80(L.down.type), stream_,
                                                    pl ? mmq_bh_ : nullptr, pl ? mmq_bl_ : nullptr));
                        prep_owner_ = nullptr;
                        if (pl) prep_written(aq_ff_, FF, T);
                        swiglu_quantised = true;
                    } else {
                        HIP_TRY(launch_swiglu(ffn_, ffn_u_, ffn_, (int64_t)T * FF, stream_));
                    }
                }
            } else if (T > 1 && !fuse_ffn && L.gate.planes && L.up.planes && L.gate.N == L.up.N && (mmq_applicable(L.gate.type, E, T) || planes_small(L.gate, E, T)) &&
                       mmq_planes_swiglu_ok(L.gate.type, L.up.type, (int)L.gate.N, E, T)) {
                // prompt batch on the LDS kernel: gate and up in one launch, 64 rows of each per workgroup, SwiGLU in the epilogue (one
                // f32 result of T x FF instead of two); the quantiser for the down projection then reads half as much
                HIP_TRY(launch_mmq_planes_swiglu(L.gate.type, L.gate.planes, L.up.planes, (int)L.gate.N, E, T, aq_e_, ffn_, FF, stream_));
                if (is_quant(L.down.type) && (FF % 256) == 0) {
                    const bool pl = !act_is_q80(L.down.type) && T >= 3;
                    HIP_TRY(launch_quantize(ffn_, FF, T, aq_ff_, !act_is_q80(L.down.type), act_is_q80(L.down.type), stream_, pl ? mmq_bh_ : nullptr, pl ? mmq_bl_ : nullptr));
                    prep_owner_ = nullptr;
                    if (pl) prep_written(aq_ff_, FF, T);
                    swiglu_quantised = true;
                }
            } else {
                HIP_TRY(linear(L.gate, aq_e_, xn_, E, T, ffn_, FF, nullptr, EPI_STORE));
                HIP_TRY(linear(L.up, aq_e_, xn_, E, T, ffn_u_, FF, nullptr, EPI_STORE));
                // prompt batch: SwiGLU and the quantisation for the down projection in one pass (no f32 round trip of T x FF)
                if (T > 1 && is_quant(L.down.type) && (FF % 256) == 0) {
                    const bool pl = !act_is_q80(L.down.type) && T >= 3;
                    HIP_TRY(launch_swiglu_quant(ffn_, ffn_u_, FF, T, aq_ff_, !act_is_q80(L.down.type), act_is_q80(L.down.type), stream_,
                                                pl ? mmq_bh_ : nullptr, pl ? mmq_bl_ : nullptr));
                    prep_owner_ = nullptr;
                    if (pl) prep_written(aq_ff_, FF, T);
                    swiglu_quantised = true;
                } else {
                    HIP_TRY(launch_swiglu(ffn_, ffn_u_, ffn_, (int64_t)T * FF, stream_));
                }
            }
            prof_mark("ffn_gate_up");
            static const bool fuse_down_env = !(getenv("MI355_FUSE_DOWN") && getenv("MI355_FUSE_DOWN")[0] == '0');
            // quantise inside the down-projection's prologue (once per CU, overlapped with its first weight loads)
            // (the widths listed are the ones the register-ring and weight-stream kernels take; the generic mat-vec's fused prologue would refuse others)
            const bool fuse_down = fuse_down_env && T == 1 && (L.down.type == T_Q4_K || L.down.type == T_Q5_K || L.down.type == T_Q6_K || act_is_q80(L.down.type) ||
                                                               L.down.type == T_Q2_K || L.down.type == T_Q3_K) &&
                                   (FF % 256) == 0 && mmvq_fast_kb_ok((FF + 2047) / 2048);
            if (fuse_down && !tp && L.down_lo.valid() && L.down_hi.valid()) {
                // the column halves of ffn_down, each quantising its half of the SwiGLU output in its prologue: x += W_lo a_lo; x += W_hi a_hi
                const int Kh = (int)L.down_lo.K;
                pending_fuse_.mode = 2; pending_fuse_.x = ffn_;
                HIP_TRY(linear(L.down_lo, aq_ff_, ffn_, Kh, T, x_, E, x_, EPI_ADD));
                pending_fuse_.mode = 2; pending_fuse_.x = ffn_ + Kh;
                HIP_TRY(linear(L.down_hi, aq_ff_, ffn_ + Kh, Kh, T, x_, E, x_, EPI_ADD));
                pending_fuse_ = Fuse();
                prof_mark("ffn_down");
                if (debug_taps_ && dbg_) HIP_TRY(hipMemcpyAsync(dbg_ + (size_t)il * cp.n_ubatch * E, x_, (size_t)T * E * 4, hipMemcpyDeviceToDevice, stream_));
                continue;
            }
            if (fuse_down) {
                pending_fuse_.mode = 2; pending_fuse_.x = ffn_;       // quantise inside the mat-vec prologue
            } else if (is_quant(L.down.type) && !swiglu_quantised) {
                HIP_TRY(launch_quantize(ffn_, FF, T, aq_ff_, !act_is_q80(L.down.type), act_is_q80(L.down.type), stream_));
                prep_owner_ = nullptr;
                prof_mark("quant");
            }
            if (tp) {
                HIP_TRY(linear(L.down, aq_ff_, ffn_, FF, T, tp_part_, E, hp.tp_rank == 0 ? x_ : nullptr, hp.tp_rank == 0 ? EPI_ADD : EPI_STORE));
                pending_fuse_ = Fuse();
                HIP_TRY(tp_reduce_into_x(T));
            } else {
                HIP_TRY(linear(L.down, aq_ff_, ffn_, FF, T, x_, E, x_, EPI_ADD));
            }
            pending_fuse_ = Fuse();
            prof_mark("ffn_down");
        }
        if (debug_taps_ && dbg_) HIP_TRY(hipMemcpyAsync(dbg_ + (size_t)il * cp.n_ubatch * E, x_, (size_t)T * E * 4, hipMemcpyDeviceToDevice, stream_));
    }
    return hipSuccess;
}

hipError_t Context::tp_reduce_into_x(int T) {
    return tp_all_reduce_sum(tp_part_, x_, (size_t)T * model->hp.n_embd, stream_);
}

hipError_t Context::run_output(int n_out, int out_base) {
    logits_on_host_ = false;
    if (n_out <= 0) return hipSuccess;
    const HParams &hp = model->hp;
    const int E = hp.n_embd, V = hp.n_vocab;
    // (a single-token step has one row and it is the flagged one: no gather launch)
    float *const xo = cur_T_ == 1 && n_out == 1 ? x_ : xo_;
    if (xo != x_) HIP_TRY(launch_gather_rows_f32(x_, d_outrow_, n_out, E, xo_, stream_));
    if (hp.encoder) {                                        // the embeddings ARE the last layer's rows (t_embd of llm_build_bert): no final norm, no head
        HIP_TRY(hipMemcpyAsync(d_embd_ + (size_t)out_base * E, xo, (size_t)n_out * E * 4, hipMemcpyDeviceToDevice, stream_));
        prof_mark("embd");
        return hipSuccess;
    }
    if (embeddings_enabled) {
        // embeddings mode (llama_set_embeddings): output = result_norm rows, no lm-head (pooling NONE on this architecture)
        HIP_TRY(launch_rmsnorm_quant(xo, (const float *)model->out_norm.data, E, n_out, hp.eps, d_embd_ + (size_t)out_base * E, nullptr, false, false, stream_));
        prof_mark("embd");
        return hipSuccess;
    }
    const bool oq = is_quant(model->output.type);
    // a single-token step whose logits row is wanted on the host: the head's launch stores it into the pinned row itself (the row crosses PCIe under the
    // launch; a copy node behind it cost 9 - 12 us of every step, tools/host_gap.py)
    static const bool zc_env = !(getenv("MI355_LOGITS_ZERO_COPY") && getenv("MI355_LOGITS_ZERO_COPY")[0] == '0');
    logits_on_host_ = zc_env && oq && cur_T_ == 1 && n_out == 1 && cp.logits_to_host && !hp.tp_exchange && h_logits_ != nullptr;
    if (oq && can_fuse(E, n_out)) {
        pending_fuse_.mode = 1; pending_fuse_.x = xo; pending_fuse_.w = (const float *)model->out_norm.data; pending_fuse_.eps = hp.eps;
    } else {
        prep_owner_ = nullptr;
        HIP_TRY(launch_rmsnorm_quant(xo, (const float *)model->out_norm.data, E, n_out, hp.eps, oq ? nullptr : xn_, &aq_e_,
                                     oq && !act_is_q80(model->output.type), act_is_q80(model->output.type), stream_));
        prof_mark("norm_quant");
    }
    float *lg = d_logits_ + (size_t)out_base * V;
    const int VL = hp.n_vocab_local, P = hp.tp_size;
    if (hp.tp_exchange && VL * P == V) {
        // vocabulary rows are cut across the ranks: each computes its slice of every flagged row, the slices are gathered
        // rank-major and copied into place (for a single row the gathered buffer already is the logits row)
        if (n_out > 1 && (size_t)n_out > tp_logits_rows_) {      // (never inside a capture: graphs carry single-row steps only)
            HIP_TRY(hipStreamSynchronize(stream_));
            tp_logits_ = (float *)dalloc((size_t)n_out * V * 4);
            if (!tp_logits_) return hipErrorOutOfMemory;
            tp_logits_rows_ = (size_t)n_out;
            HIP_TRY(hipDeviceSynchronize());
        }
        float *gathered = n_out == 1 ? lg : tp_logits_;
        float *part = gathered + (size_t)hp.tp_rank * n_out * VL;
        HIP_TRY(linear(model->output, aq_e_, xn_, E, n_out, part, VL, nullptr, EPI_STORE));
        pending_fuse_ = Fuse();
        HIP_TRY(tp_all_gather(part, gathered, (size_t)n_out * VL, stream_));
        for (int r = 0; r < P && n_out > 1; r++)
            HIP_TRY(hipMemcpy2DAsync(lg + (size_t)r * VL, (size_t)V * 4, tp_logits_ + (size_t)r * n_out * VL, (size_t)VL * 4, (size_t)VL * 4, (size_t)n_out,
                                     hipMemcpyDeviceToDevice, stream_));
    } else {
        if (logits_on_host_) pending_fuse_.out_host = h_logits_ + (size_t)out_base * V;
        HIP_TRY(linear(model->output, aq_e_, xn_, E, n_out, lg, V, nullptr, EPI_STORE));
    }
    pending_fuse_ = Fuse();
    prof_mark("lm_head");
    for (int r0 = 0; r0 < n_out; r0 += (int)cp.n_ubatch) {
        const int nr = std::min((int)cp.n_ubatch, n_out - r0);
        // the winner goes straight into pinned host memory (visible once the stream has drained): no copy node
        HIP_TRY(launch_argmax_rows(lg + (size_t)r0 * V, V, nr, h_argmax_ + out_base + r0, argmax_scratch_, (int)cp.n_ubatch, stream_));
    }
    prof_mark("argmax");
    return hipSuccess;
}

int Context::decode_ubatch(int n, const int32_t *tokens, const int32_t *pos, const int32_t *seq, const uint64_t *seqmask,
                           const int8_t *flags, int out_base) {
    { const unsigned long long ls = next_launch_serial(); if (!first_unchecked_launch_) first_unchecked_launch_ = ls; }   // (stream_check: who is a suspect)
    cur_max_pos_ = 0;
    for (int i = 0; i < n; i++) cur_max_pos_ = std::max(cur_max_pos_, (int)pos[i]);
    std::vector<int> tcell;
    if (!alloc_cells(n, seqmask, tcell)) return 1;
    for (int i = 0; i < n; i++) {
        KVCell &c = cells_[(size_t)tcell[(size_t)i]];
        c.pos = pos[i]; c.seqs = seqmask[i]; c.delta = 0;
    }
    int hi = 0;
    for (int i = (int)cells_.size() - 1; i >= 0; i--) if (cells_[(size_t)i].pos >= 0) { hi = i + 1; break; }
    n_kv_ = std::min((int)cp.n_ctx, (hi + 31) & ~31);

    if (meta_dirty_) {   // sequence ops since the last batch: re-upload the whole cell table
        std::vector<int32_t> cpos(cells_.size());
        std::vector<uint64_t> cseq(cells_.size());
        for (size_t i = 0; i < cells_.size(); i++) { cpos[i] = cells_[i].pos; cseq[i] = cells_[i].seqs; }
        if (hipMemcpyAsync(d_cell_pos_, cpos.data(), cpos.size() * 4, hipMemcpyHostToDevice, stream_) != hipSuccess ||
            hipMemcpyAsync(d_cell_seq_, cseq.data(), cseq.size() * 8, hipMemcpyHostToDevice, stream_) != hipSuccess ||
            hipStreamSynchronize(stream_) != hipSuccess) { last_error = "cell table upload failed"; return -1; }
        meta_dirty_ = false;
    }
    // stage token arrays (the pinned block may still be in flight from the previous micro-batch)
    if (stage_event_) (void)hipEventSynchronize(stage_event_);
    const size_t T = ((size_t)cp.n_ubatch + 1) & ~(size_t)1;
    int32_t *h_nkv = (int32_t *)h_stage_;
    int32_t *h_tok = (int32_t *)(h_stage_ + 16), *h_pos = h_tok + T, *h_seq = h_pos + T, *h_cell = h_seq + T, *h_out = h_cell + T;
    uint64_t *h_mask = (uint64_t *)(h_out + T);
    h_nkv[0] = n_kv_;
    int n_out = 0;
    for (int i = 0; i < n; i++) {
        h_tok[i] = tokens[i]; h_pos[i] = pos[i]; h_seq[i] = seq[i]; h_cell[i] = tcell[(size_t)i]; h_mask[i] = seqmask[i];
        if (flags[i]) h_out[n_out++] = i;
    }
    if (hipMemcpyAsync(d_stage_, h_stage_, stage_bytes_, hipMemcpyHostToDevice, stream_) != hipSuccess) { last_error = "token upload failed"; return -1; }
    // continuous-batching steps (a few tokens, usually one per sequence): each token scans only the 64-cell chunks that
    // hold cells of its own sequence instead of the whole cache
    chunk_lmax_ = 0;
    batch_distinct_ = n >= 2 && n <= 64;                       // every token in exactly one sequence, no two in the same one (a continuous-batching step)
    {
        uint64_t seen = 0;
        for (int i = 0; i < n && batch_distinct_; i++) {
            const uint64_t mk = seqmask[i];
            if (mk == 0 || (mk & (mk - 1)) != 0 || (seen & mk) != 0) batch_distinct_ = false;
            seen |= mk;
        }
    }
    bool lists_usable = false;                                 // only the split-per-chunk attention kernels walk the lists
    {
        const HParams &hp = model->hp;
        RopeArgs ra = rope_args(*model, false);
        AttnArgs aa{};
        aa.T = n; aa.H = hp.n_head; aa.G = hp.n_head_kv; aa.D = hp.head_dim; aa.n_kv_max = n_kv_; aa.type_k = cp.type_k; aa.type_v = cp.type_v;
        lists_usable = n <= 64 && flash_attn_decode_applicable(aa, ra) && kv_store_fast_applicable(aa.G, aa.D, cp.type_k, cp.type_v, ra);
    }
    if ((n >= 2 || cp.n_seq_max > 1) && lists_usable) {
        const int nch = (n_kv_ + 63) / 64;
        std::vector<uint64_t> seq_of_chunk((size_t)nch, 0ull);
        for (int i = 0; i < n_kv_ && i < (int)cells_.size(); i++)
            if (cells_[(size_t)i].pos >= 0) seq_of_chunk[(size_t)(i >> 6)] |= cells_[(size_t)i].seqs;
        int32_t *cnt = h_chunks_ + (size_t)64 * chunk_stride_;
        for (int t = 0; t < n; t++) {
            int32_t *lst = h_chunks_ + (size_t)t * chunk_stride_;
            int c = 0;
            const uint64_t bit = 1ull << (seq[t] & 63);
            for (int ch = 0; ch < nch; ch++) if (seq_of_chunk[(size_t)ch] & bit) lst[c++] = ch;
            cnt[t] = c;
            chunk_lmax_ = std::max(chunk_lmax_, c);
        }
        if (chunk_lmax_ < 1) chunk_lmax_ = 1;
        if (hipMemcpyAsync(d_chunks_, h_chunks_, (size_t)64 * (chunk_stride_ + 1) * 4, hipMemcpyHostToDevice, stream_) != hipSuccess) { last_error = "chunk list upload failed"; return -1; }
    }
    // rocprofv3 --pmc does not keep copy-engine transfers ordered with the kernels it serialises (observed: kernels reading
    // the previous step's token / position block); MI355_PROFILER_SAFE=1 drains the stream around the transfers
    static const bool profiler_safe = getenv("MI355_PROFILER_SAFE") && getenv("MI355_PROFILER_SAFE")[0] == '1';
    if (profiler_safe) (void)hipStreamSynchronize(stream_);
    if (!stage_event_) (void)hipEventCreateWithFlags(&stage_event_, hipEventDisableTiming);
    if (stage_event_) (void)hipEventRecord(stage_event_, stream_);

    const int V = model->hp.n_vocab;
    if (n == 1 && !model->hp.encoder && !ub_embd_) { (void)mega_prepare(); (void)engine_prepare(); }   // allocate and upload on first use: must not happen inside a stream capture
    // a single-token step would launch the attention + attn_output kernel whose workgroups wait for each other: only while no other context of this device has
    // such a step in flight (runtime.h fused_acquire); otherwise this step takes the wait-free launches, eagerly (the captured graphs hold the fused kernel)
    attn_out_skip_step_ = n == 1 && !attn_out_off_ && !model->hp.encoder && !fused_acquire();
    if (attn_out_skip_step_) fused_skipped_steps++;
    bool graph_ok = cp.use_graphs && n == 1 && n_out == 1 && out_base == 0 && !profile_ && !debug_taps_ && !embeddings_enabled && moe_forced_T_ == 0 && !ub_embd_ && !attn_out_skip_step_;
    hipError_t e = hipSuccess;
    if (graph_ok) {
        // the attention grid is sized for an upper bound of occupied cells; one captured graph per 256-cell bucket
        int bucket = std::min((int)cp.n_ctx, (n_kv_ + 255) & ~255);
        chunk_cap_ = 0;
        if (chunk_lmax_ > 0) {                                 // chunk lists: the grid is sized by the list length, in steps of 4
            chunk_cap_ = std::min(chunk_stride_, (chunk_lmax_ + 3) & ~3);
            bucket = -chunk_cap_;                              // separate key space from the cell-count buckets
        }
        auto git = graphs_.find(bucket);
        graph_exec_ = git == graphs_.end() ? nullptr : git->second;
        if (!graph_exec_) {
            hipGraph_t g = nullptr;
            std::string inner;                                  // which launch refused, when one did (HIP_TRY's message)
            e = hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                last_error.clear();
                hipError_t e2 = run_layers(1, bucket > 0 ? bucket : std::min((int)cp.n_ctx, chunk_cap_ * 64));
                if (e2 == hipSuccess) e2 = run_output(1, 0);
                if (e2 == hipSuccess && cp.logits_to_host && !logits_on_host_) e2 = hipMemcpyAsync(h_logits_, d_logits_, (size_t)V * 4, hipMemcpyDeviceToHost, stream_);
                if (e2 != hipSuccess) inner = last_error;
                e = hipStreamEndCapture(stream_, &g);
                if (e2 != hipSuccess) e = e2;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess && model->hp.tp_exchange) {
                // a collective that cannot be captured on this RCCL / driver pair: every rank fails the same way, so all
                // of them continue with eager launches from here on (nothing was executed during the failed capture)
                (void)hipGetLastError();
                cp.use_graphs = false;
                graph_ok = false;
                graph_exec_ = nullptr;
                e = hipSuccess;
            } else {
                if (e != hipSuccess) { last_error = std::string("graph capture failed: ") + hipGetErrorString(e) + (inner.empty() ? "" : " / " + inner); return -1; }
                graphs_[bucket] = graph_exec_;
                graph_is_mega_[graph_exec_] = last_layers_mega_;
                graph_is_engine_[graph_exec_] = last_layers_engine_;
            }
        }
        if (graph_ok) e = hipGraphLaunch(graph_exec_, stream_);
    }
    if (!graph_ok) {
        prof_begin();
        chunk_cap_ = chunk_lmax_;
        e = run_layers(n, n_kv_);
        if (e == hipSuccess) e = run_output(n_out, out_base);
        if (profiler_safe) (void)hipStreamSynchronize(stream_);
        if (e == hipSuccess && n_out > 0 && cp.logits_to_host && !embeddings_enabled && !logits_on_host_)
            e = hipMemcpyAsync(h_logits_ + (size_t)out_base * V, d_logits_ + (size_t)out_base * V, (size_t)n_out * V * 4, hipMemcpyDeviceToHost, stream_);
        prof_end();
    }
    if (e != hipSuccess) { last_error = std::string("decode failed: ") + hipGetErrorString(e) + " / " + last_error_string(); return -1; }
    if (n == 1 && (graph_ok ? graph_is_mega_[graph_exec_] : last_layers_mega_)) mega_steps++;
    if (n == 1 && (graph_ok ? graph_is_engine_[graph_exec_] : last_layers_engine_)) engine_steps++;
    dbg_tokens_ = n;
    return 0;
}

// GPU-side trace ranges (tracing row): with MI355_ROCTX=1 every decode call is a roctx range - "mi355_decode prompt n=512" / "mi355_decode step
// n=1" - so a rocprofv3 --marker-trace run shows where the batches begin and end between the kernels.  The marker library is looked up at run time
// (librocprofiler-sdk-roctx, then the legacy libroctx64); absent or switched off, a range costs one branch.
namespace {
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char *ev = getenv("MI355_ROCTX");
        if (!ev || ev[0] != '1') return;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
struct TraceRange {
    static RoctxApi &api() { static RoctxApi a; return a; }
    bool on = false;
    TraceRange(const char *what, int n) {
        if (!api().push) return;
        char buf[96];
        snprintf(buf, sizeof buf, "mi355_decode %s n=%d", what, n);
        api().push(buf);
        on = true;
    }
    ~TraceRange() { if (on) api().pop(); }
};
}  // namespace

int Context::decode(int n_tokens, const int32_t *tokens, const int32_t *pos, const int32_t *n_seq_id, int32_t *const *seq_id,
                    const int8_t *logits_flags, const float *embd) {
    if (n_tokens <= 0) { last_error = "empty batch"; return -1; }
    const TraceRange trace_range(embd ? "embeddings" : n_tokens == 1 ? "step" : n_tokens <= 64 ? "steps" : "prompt", n_tokens);
    // llama_batch.embd: the rows ARE the layer stack's input (the image embeddings of a LLaVA request, llama_server_context.cc:1093-1107); no token ids then
    std::vector<int32_t> no_tokens;
    if (embd) {
        if (model->hp.encoder) { last_error = "an encoder model takes token ids, not embeddings"; return -1; }
        if (tokens) { last_error = "a batch carries token ids or embeddings, not both"; return -1; }
        if (!pos) { last_error = "an embeddings batch needs positions"; return -1; }
        no_tokens.assign((size_t)n_tokens, 0);
        tokens = no_tokens.data();
    } else if (!tokens) { last_error = "batch without tokens"; return -1; }
    if (hipSetDevice(model->device) != hipSuccess) return -1;
    const HParams &hp = model->hp;
    // bidirectional attention: every token of a sequence must see all the others, so a batch is never cut into micro-batches (llama.cpp asks the same: n_ubatch >= n_tokens)
    if (hp.encoder && (size_t)n_tokens > cp.n_ubatch) { last_error = "encoder model: a batch of " + std::to_string(n_tokens) + " tokens does not fit one micro-batch (n_ubatch " + std::to_string(cp.n_ubatch) + ")"; return -1; }
    if (has_shift_) apply_k_shift();
    if (debug_taps_ && !dbg_) dbg_ = (float *)dalloc((size_t)hp.n_layer * cp.n_ubatch * hp.n_embd * 4);

    std::vector<int32_t> seq((size_t)n_tokens);
    std::vector<uint64_t> mask((size_t)n_tokens);
    std::vector<int8_t> flags((size_t)n_tokens);
    int n_out = 0;
    out_row_of_batch_.assign((size_t)n_tokens, -1);
    for (int i = 0; i < n_tokens; i++) {
        if (tokens[i] < 0 || tokens[i] >= hp.n_vocab) { last_error = "token id " + std::to_string(tokens[i]) + " out of range [0, " + std::to_string(hp.n_vocab) + ")"; return -1; }
        // a negative position would mark the cell it is written to as free although its KV row was stored
        if (pos && pos[i] < 0) { last_error = "negative position " + std::to_string(pos[i]) + " for token " + std::to_string(i); return -1; }
        uint64_t mk = 0;
        const int ns = n_seq_id ? n_seq_id[i] : 1;
        for (int j = 0; j < ns; j++) {
            const int s = seq_id ? seq_id[i][j] : 0;
            if (s < 0 || s >= 64) { last_error = "seq id " + std::to_string(s) + " out of range [0, 64)"; return -1; }   // (the cell masks are 64 bits wide)
            mk |= 1ull << s;
        }
        seq[(size_t)i] = seq_id ? seq_id[i][0] : 0;
        mask[(size_t)i] = mk;
        // llama_decode: logits == NULL -> only the last token
        flags[(size_t)i] = logits_flags ? (logits_flags[i] != 0) : (i == n_tokens - 1);
        if (flags[(size_t)i]) out_row_of_batch_[(size_t)i] = n_out++;
    }
    // logits buffers
    if ((size_t)n_out > logits_cap_rows_) {
        (void)hipStreamSynchronize(stream_);
        if (d_logits_) { (void)hipFree(d_logits_); allocs_.erase(std::find(allocs_.begin(), allocs_.end(), (void *)d_logits_)); }
        if (h_logits_) (void)hipHostFree(h_logits_);
        if (d_argmax_) { /* sized n_ubatch; regrow below if needed */ }
        const size_t rows = std::max<size_t>((size_t)n_out, 1);
        d_logits_ = (float *)dalloc(rows * hp.n_vocab * 4);
        if (!d_logits_ || hipHostMalloc((void **)&h_logits_, rows * hp.n_vocab * 4, hipHostMallocDefault) != hipSuccess) { last_error = "logits alloc failed"; return -1; }
        if (rows > cp.n_ubatch) {
            d_argmax_ = (int32_t *)dalloc(rows * 4);
            (void)hipHostFree(h_argmax_);
            if (hipHostMalloc((void **)&h_argmax_, rows * 4, hipHostMallocDefault) != hipSuccess) return -1;
        }
        logits_cap_rows_ = rows;
        (void)hipDeviceSynchronize();                  // null-stream zero-fills of the new buffers (see init)
        for (auto &ge : graphs_) (void)hipGraphExecDestroy(ge.second);
        graphs_.clear();
        graph_exec_ = nullptr;
    }
    n_out_last_ = n_out;
    logits_fetched_ = false;
    embd_fetched_ = false;
    last_was_embd_ = embeddings_enabled;
    if (embeddings_enabled && (size_t)n_out > cp.n_ubatch) { last_error = "embeddings: more flagged rows than n_ubatch"; return -1; }
    argmax_fetched_ = false;

    const std::vector<KVCell> saved = cells_;
    const int saved_head = head_;
    int out_base = 0;
    for (int i0 = 0; i0 < n_tokens; i0 += (int)cp.n_ubatch) {
        const int n = std::min((int)cp.n_ubatch, n_tokens - i0);
        ub_embd_ = embd ? embd + (size_t)i0 * (size_t)hp.n_embd : nullptr;
        const int rc = decode_ubatch(n, tokens + i0, pos + i0, seq.data() + i0, mask.data() + i0, flags.data() + i0, out_base);
        ub_embd_ = nullptr;
        if (rc != 0) {
            cells_ = saved; head_ = saved_head; meta_dirty_ = true; region_next_.clear();
            return rc;
        }
        for (int i = 0; i < n; i++) out_base += flags[(size_t)(i0 + i)] ? 1 : 0;
    }
    moe_forced_T_ = 0;      // (force_moe_ids arms one call)
    return 0;
}

void Context::synchronize() { (void)hipStreamSynchronize(stream_); fused_release(); }

int Context::force_moe_ids(const int32_t *ids, int n_layer, int T, int k) {
    const HParams &hp = model->hp;
    if (!ids || hp.n_expert <= 0 || n_layer != hp.n_layer || k != hp.n_expert_used || T < 1 || T > (int)cp.n_ubatch) { last_error = "force_moe_ids: shape does not match the model"; return -1; }
    const int n = n_layer * T * k;
    if (hipSetDevice(model->device) != hipSuccess) return -1;
    if (n > moe_forced_cap_) {
        (void)hipStreamSynchronize(stream_);
        d_moe_forced_ = (int32_t *)dalloc((size_t)n * 4);
        if (!d_moe_forced_) { moe_forced_cap_ = 0; return -1; }
        moe_forced_cap_ = n;
    }
    if (hipMemcpyAsync(d_moe_forced_, ids, (size_t)n * 4, hipMemcpyHostToDevice, stream_) != hipSuccess || hipStreamSynchronize(stream_) != hipSuccess) return -1;
    moe_forced_T_ = T;
    return 0;
}

float *Context::logits_ith(int i) {
    if (last_was_embd_) return nullptr;            // embeddings mode computes no logits
    if (i < 0) i += (int)out_row_of_batch_.size();
    if (i < 0 || i >= (int)out_row_of_batch_.size() || out_row_of_batch_[(size_t)i] < 0) return nullptr;
    if (!logits_fetched_) {
        if (!cp.logits_to_host &&
            hipMemcpyAsync(h_logits_, d_logits_, (size_t)n_out_last_ * model->hp.n_vocab * 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return nullptr;
        if (hipStreamSynchronize(stream_) != hipSuccess || !mega_check() || !stream_check()) return nullptr;
        logits_fetched_ = true;
    }
    return h_logits_ + (size_t)out_row_of_batch_[(size_t)i] * model->hp.n_vocab;
}

float *Context::embeddings_ith(int i) {
    if (!last_was_embd_) return nullptr;
    if (i < 0) i += (int)out_row_of_batch_.size();
    if (i < 0 || i >= (int)out_row_of_batch_.size() || out_row_of_batch_[(size_t)i] < 0) return nullptr;
    const int E = model->hp.n_embd;
    if (!embd_fetched_) {
        if (hipMemcpyAsync(h_embd_, d_embd_, (size_t)n_out_last_ * E * 4, hipMemcpyDeviceToHost, stream_) != hipSuccess) return nullptr;
        if (hipStreamSynchronize(stream_) != hipSuccess) return nullptr;
        embd_fetched_ = true;
    }
    return h_embd_ + (size_t)out_row_of_batch_[(size_t)i] * E;
}

int32_t Context::argmax_ith(int i) {
    if (i < 0) i += (int)out_row_of_batch_.size();
    if (i < 0 || i >= (int)out_row_of_batch_.size() || out_row_of_batch_[(size_t)i] < 0) return -1;
    if (!argmax_fetched_) {
        if (hipStreamSynchronize(stream_) != hipSuccess || !mega_check() || !stream_check()) return -1;
        argmax_fetched_ = true;
    }
    return h_argmax_[out_row_of_batch_[(size_t)i]];
}

// n rows of the last batch in ONE set of launches and ONE synchronisation (a scheduler tick asks for all its sampling slots together: row by row, each
// with its own launch + sync, 32 slots cost more than copying 32 rows to the host did).  toks / logits: [n][TOPK_MAX_K]; ks[r] of each row are valid.
int Context::topk_rows(int n, const int *is, const int *ks, const TopkAdj *adjs, int32_t *toks, float *logits) {
    if (last_was_embd_ || n < 1) return -1;
    const int V = model->hp.n_vocab;
    int kmax = 0;
    std::vector<int> rows((size_t)n);
    for (int r = 0; r < n; r++) {
        int i = is[r];
        if (i < 0) i += (int)out_row_of_batch_.size();
        if (i < 0 || i >= (int)out_row_of_batch_.size() || out_row_of_batch_[(size_t)i] < 0) return -1;
        if (ks[r] < 1 || ks[r] > TOPK_MAX_K || ks[r] > V || adjs[r].n < 0 || adjs[r].n > TOPK_MAX_ADJ) return -1;
        rows[(size_t)r] = out_row_of_batch_[(size_t)i];
        kmax = std::max(kmax, ks[r]);
    }
    if (hipSetDevice(model->device) != hipSuccess) return -1;
    if (n > topk_rows_cap_) {
        (void)hipStreamSynchronize(stream_);
        if (topk_scratch_) (void)hipFree(topk_scratch_);
        if (d_topk_adj_) (void)hipFree(d_topk_adj_);
        if (h_topk_) (void)hipHostFree(h_topk_);
        if (h_topk_adj_) (void)hipHostFree(h_topk_adj_);
        topk_scratch_ = nullptr; d_topk_adj_ = nullptr; h_topk_ = nullptr; h_topk_adj_ = nullptr; topk_rows_cap_ = 0;
        const int cap = std::max(n, std::min(64, (int)cp.n_seq_max));
        const size_t adj_bytes = (size_t)cap * sizeof(TopkAdj) + (size_t)cap * sizeof(int);
        if (hipMalloc(&topk_scratch_, topk_scratch_bytes(V) * (size_t)cap) != hipSuccess || hipMalloc((void **)&d_topk_adj_, adj_bytes) != hipSuccess ||
            hipHostMalloc((void **)&h_topk_, (size_t)cap * TOPK_MAX_K * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&h_topk_adj_, adj_bytes, hipHostMallocDefault) != hipSuccess) { last_error = "top-k workspace allocation failed"; return -1; }
        topk_rows_cap_ = cap;
    }
    // adjustments and row numbers: one staged copy
    TopkAdj *ha = reinterpret_cast<TopkAdj *>(h_topk_adj_);
    int *hr = reinterpret_cast<int *>(h_topk_adj_ + (size_t)topk_rows_cap_ * sizeof(TopkAdj));
    for (int r = 0; r < n; r++) {
        TopkAdj &d = ha[r];
        const TopkAdj &a = adjs[r];
        d.n = a.n; d.repeat = a.repeat; d.freq = a.freq; d.present = a.present;
        memcpy(d.tok, a.tok, (size_t)a.n * sizeof(int)); memcpy(d.bias, a.bias, (size_t)a.n * sizeof(float)); memcpy(d.cnt, a.cnt, (size_t)a.n * sizeof(int));
        hr[r] = rows[(size_t)r];
    }
    const size_t rows_off = (size_t)topk_rows_cap_ * sizeof(TopkAdj);
    if (hipMemcpyAsync(d_topk_adj_, h_topk_adj_, (size_t)n * sizeof(TopkAdj), hipMemcpyHostToDevice, stream_) != hipSuccess ||
        hipMemcpyAsync(d_topk_adj_ + rows_off, h_topk_adj_ + rows_off, (size_t)n * sizeof(int), hipMemcpyHostToDevice, stream_) != hipSuccess) { last_error = "top-k staging failed"; return -1; }
    if (launch_topk_rows(d_logits_, V, n, reinterpret_cast<const int *>(d_topk_adj_ + rows_off), kmax, reinterpret_cast<const TopkAdj *>(d_topk_adj_), topk_scratch_, h_topk_,
                         stream_) != hipSuccess) { last_error = "top-k launch failed"; return -1; }
    if (hipStreamSynchronize(stream_) != hipSuccess || !mega_check() || !stream_check()) return -1;
    for (int r = 0; r < n; r++)
        for (int j = 0; j < ks[r]; j++) {
            const unsigned long long key = h_topk_[(size_t)r * TOPK_MAX_K + (size_t)j];
            unsigned u = (unsigned)(key >> 32);
            u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;        // inverse of the order-preserving image
            float f;
            memcpy(&f, &u, 4);
            toks[(size_t)r * TOPK_MAX_K + (size_t)j] = (int32_t)(0xffffffffu - (unsigned)(key & 0xffffffffull));
            logits[(size_t)r * TOPK_MAX_K + (size_t)j] = f;
        }
    return n;
}

int Context::topk_ith(int i, int k, const TopkAdj &adj, int32_t *toks, float *logits) {
    int32_t t[TOPK_MAX_K];
    float l[TOPK_MAX_K];
    if (topk_rows(1, &i, &k, &adj, t, l) != 1) return -1;
    memcpy(toks, t, (size_t)k * sizeof(int32_t));
    memcpy(logits, l, (size_t)k * sizeof(float));
    return k;
}

int Context::debug_layer_out(int il, float *dst, size_t cap) {
    const HParams &hp = model->hp;
    if (!dbg_ || il < 0 || il >= hp.n_layer) return -1;
    const size_t n = (size_t)dbg_tokens_ * hp.n_embd;
    if (cap < n) return -1;
    (void)hipStreamSynchronize(stream_);
    if (hipMemcpy(dst, dbg_ + (size_t)il * cp.n_ubatch * hp.n_embd, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return dbg_tokens_;
}

// every weight tensor once, as one decoded token reads them (no attention, no norms): the dominant kernel
// launches_out: mat-vec launches per sweep.  Where the step runs attn_output inside its attention launch (attn_out.hip) the sweep holds no attn_output
// launch either: it times the launches of the dominant kernel class as the step issues them, and counts their bytes only.
double Context::bench_weight_sweep(int iters, uint64_t *bytes_out, int *launches_out) {
    const HParams &hp = model->hp;
    const int E = hp.n_embd, FF = hp.n_ff;
    (void)hipMemsetAsync(x_, 0, (size_t)E * 4, stream_);
    uint64_t bytes = 0;
    int launches = 0;
    auto wo_in_attention = [&](const LayerWeights &L) {
        AttnArgs af{};
        af.type_k = cp.type_k; af.type_v = cp.type_v; af.T = 1; af.H = hp.n_head; af.G = hp.n_head_kv; af.D = hp.head_dim; af.n_ctx = (int)cp.n_ctx;
        af.n_kv_max = 64; af.splits = 1; af.out_q = &aq_o_; af.out_q8k = !act_is_q80(L.wo.type); af.out_q80 = act_is_q80(L.wo.type);
        const MMVQSeg so = make_seg(L.wo, x_, E, x_, nullptr);
        return !hp.tp_exchange && is_quant(L.wo.type) && attn_out_fused_applicable(af, rope_args(*model, true), so, (int)L.wo.K, EPI_ADD);
    };
    // round 6: ... and Q | K | V too where attn_out.hip takes them (run_layers): the sweep then holds neither
    auto qkv_in_attention = [&](const LayerWeights &L) {
        if (!wo_in_attention(L) || !d_qkv_gran_ || L.bq.valid() || L.bk.valid() || L.bv.valid()) return false;
        if (!is_quant(L.wq.type) || !is_quant(L.wk.type) || !is_quant(L.wv.type) || !can_fuse(E, 1)) return false;
        AttnArgs af{};
        af.type_k = cp.type_k; af.type_v = cp.type_v; af.T = 1; af.H = hp.n_head; af.G = hp.n_head_kv; af.D = hp.head_dim; af.n_ctx = (int)cp.n_ctx;
        af.n_kv_max = 64; af.splits = 1; af.out_q = &aq_o_; af.out_q8k = !act_is_q80(L.wo.type); af.out_q80 = act_is_q80(L.wo.type);
        const MMVQSeg so = make_seg(L.wo, x_, E, x_, nullptr);
        QKVFuse qf{};
        qf.seg[0] = make_seg(L.wq, q_, (int)L.wq.N, nullptr, nullptr); qf.seg[1] = make_seg(L.wk, k_, (int)L.wk.N, nullptr, nullptr); qf.seg[2] = make_seg(L.wv, v_, (int)L.wv.N, nullptr, nullptr);
        qf.nx = x_; qf.nw = (const float *)L.attn_norm.data; qf.neps = hp.eps; qf.K = E; qf.gran = d_qkv_gran_;
        return qkv_attn_out_applicable(af, rope_args(*model, true), so, (int)L.wo.K, EPI_ADD, qf);
    };
    auto sweep = [&](bool count) -> hipError_t {
        for (int il = 0; il < hp.n_layer; il++) {
            const LayerWeights &L = model->layers[(size_t)il];
            if (hp.n_expert > 0) {
                // mixture of experts: the attention projections and the token's n_expert_used experts (indices 0..k-1 stand in
                // for a selection; every expert has the same shape and bytes)
                const int KU = hp.n_expert_used;
                if (!moe_ids_ || KU < 1 || KU > 8 || (int)cp.n_ubatch < KU || L.gate_exps.type != L.up_exps.type) continue;
                if (count) { std::vector<int32_t> ids((size_t)KU); for (int j = 0; j < KU; j++) ids[(size_t)j] = j; HIP_TRY(hipMemcpy(moe_ids_, ids.data(), ids.size() * 4, hipMemcpyHostToDevice)); }
                const DevTensor *wsm[3] = {&L.wq, &L.wk, &L.wv};
                float *outm[3] = {q_, k_, v_};
                const bool qkv_fused = qkv_in_attention(L);
                pending_fuse_.mode = 1; pending_fuse_.x = x_; pending_fuse_.w = (const float *)L.attn_norm.data; pending_fuse_.eps = hp.eps;
                if (!qkv_fused) HIP_TRY(linear_multi(wsm, outm, 3, aq_e_, xn_, 1));
                pending_fuse_ = Fuse();
                const bool wo_fused = wo_in_attention(L);
                if (!wo_fused) HIP_TRY(linear(L.wo, aq_o_, att_, (int)L.wo.K, 1, xo_, E, nullptr, EPI_STORE));
                MMVQArgs a{};
                a.n_seg = 2; a.K = E; a.T = 1; a.epi = EPI_SWIGLU; a.n_sel = KU; a.sel_out_stride = FF;
                a.seg[0] = make_seg(L.gate_exps, ffn_, FF, nullptr, moe_ids_);
                a.seg[1] = make_seg(L.up_exps, ffn_u_, FF, nullptr, moe_ids_);
                chunk_act(a, aq_e_, E, 0);
                MMVQArgs d{};
                d.n_seg = 1; d.K = FF; d.T = 1; d.epi = EPI_STORE; d.fuse_mode = 2; d.nx = ffn_; d.n_sel = KU; d.sel_nx_stride = FF; d.sel_out_stride = E;
                d.seg[0] = make_seg(L.down_exps, moe_out_, E, nullptr, moe_ids_);
                chunk_act(d, aq_ff_, FF, 0);
                if (!mmvq_fast_applicable(a) || !mmvq_fast_applicable(d)) continue;
                HIP_TRY(launch_mmvq_fast(a, stream_));
                HIP_TRY(launch_mmvq_fast(d, stream_));
                if (count) {
                    bytes += (qkv_fused ? 0 : L.wq.ggml_bytes + L.wk.ggml_bytes + L.wv.ggml_bytes) + (wo_fused ? 0 : L.wo.ggml_bytes) +
                             (L.gate_exps.ggml_bytes + L.up_exps.ggml_bytes + L.down_exps.ggml_bytes) / (uint64_t)L.gate_exps.n_expert * (uint64_t)KU;
                    launches += (wo_fused ? 3 : 4) - (qkv_fused ? 1 : 0);
                }
                continue;
            }
            // the launches of a single-token step as run_layers issues them (same fused prologues, same epilogues)
            const DevTensor *ws[3] = {&L.wq, &L.wk, &L.wv};
            float *outs[3] = {q_, k_, v_};
            const bool qkv_q = is_quant(L.wq.type) && is_quant(L.wk.type) && is_quant(L.wv.type);
            const bool qkv_fused = qkv_in_attention(L);
            if (qkv_q && can_fuse(E, 1)) { pending_fuse_.mode = 1; pending_fuse_.x = x_; pending_fuse_.w = (const float *)L.attn_norm.data; pending_fuse_.eps = hp.eps; }
            if (!qkv_fused) HIP_TRY(linear_multi(ws, outs, 3, aq_e_, xn_, 1));
            pending_fuse_ = Fuse();
            const bool wo_fused = wo_in_attention(L);
            if (!wo_fused) HIP_TRY(linear(L.wo, aq_o_, att_, (int)L.wo.K, 1, xo_, E, xo_, EPI_ADD));
            if (is_quant(L.gate.type) && L.gate.type == L.up.type) {
                MMVQSeg segs[2] = {make_seg(L.gate, ffn_, FF, nullptr, nullptr), make_seg(L.up, ffn_u_, FF, nullptr, nullptr)};
                Fuse fz;
                if (can_fuse(E, 1)) { fz.mode = 1; fz.x = x_; fz.w = (const float *)L.ffn_norm.data; fz.eps = hp.eps; }
                HIP_TRY(mmvq_tokens(segs, 2, E, 1, EPI_SWIGLU, aq_e_, stream_, fz));
            }
            const bool halves = L.down_lo.valid() && L.down_hi.valid() && !hp.tp_exchange;      // (run_layers: the column halves, two launches)
            if (halves) {
                const int Kh = (int)L.down_lo.K;
                pending_fuse_.mode = 2; pending_fuse_.x = ffn_;
                HIP_TRY(linear(L.down_lo, aq_ff_, ffn_, Kh, 1, xo_, E, xo_, EPI_ADD));
                pending_fuse_.mode = 2; pending_fuse_.x = ffn_ + Kh;
                HIP_TRY(linear(L.down_hi, aq_ff_, ffn_ + Kh, Kh, 1, xo_, E, xo_, EPI_ADD));
            } else {
                if (is_quant(L.down.type) && (FF % 256) == 0) { pending_fuse_.mode = 2; pending_fuse_.x = ffn_; }
                HIP_TRY(linear(L.down, aq_ff_, ffn_, FF, 1, xo_, E, xo_, EPI_ADD));
            }
            pending_fuse_ = Fuse();
            if (count) {
                bytes += (qkv_fused ? 0 : L.wq.ggml_bytes + L.wk.ggml_bytes + L.wv.ggml_bytes) + (wo_fused ? 0 : L.wo.ggml_bytes) + L.gate.ggml_bytes + L.up.ggml_bytes + L.down.ggml_bytes;
                launches += (wo_fused ? 3 : 4) - (qkv_fused ? 1 : 0) + (halves ? 1 : 0);
            }
        }
        if (is_quant(model->output.type) && can_fuse(E, 1)) { pending_fuse_.mode = 1; pending_fuse_.x = x_; pending_fuse_.w = (const float *)model->out_norm.data; pending_fuse_.eps = hp.eps; }
        HIP_TRY(linear(model->output, aq_e_, xn_, E, 1, d_logits_, (int)model->output.N, nullptr, EPI_STORE));
        pending_fuse_ = Fuse();
        if (count) { bytes += model->output.ggml_bytes; launches += 1; }
        return hipSuccess;
    };
    if (!d_logits_) { d_logits_ = (float *)dalloc((size_t)hp.n_vocab * 4); logits_cap_rows_ = 0; }
    if (sweep(true) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipStreamSynchronize(stream_);
    // replayed from a hipGraph, as the decode step is (eager launches are host-bound at these kernel lengths)
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    bool graphed = cp.use_graphs && hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (graphed) {
        const hipError_t es = sweep(false);
        const hipError_t ec = hipStreamEndCapture(stream_, &g);
        graphed = es == hipSuccess && ec == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess;
        if (graphed) { (void)hipGraphLaunch(ge, stream_); (void)hipStreamSynchronize(stream_); }
    }
    (void)hipEventRecord(e0, stream_);
    for (int i = 0; i < iters; i++) {
        if (graphed) { if (hipGraphLaunch(ge, stream_) != hipSuccess) return -1.0; }
        else if (sweep(false) != hipSuccess) return -1.0;
    }
    (void)hipEventRecord(e1, stream_);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    if (bytes_out) *bytes_out = bytes;
    if (launches_out) *launches_out = launches;
    return (double)ms * 1000.0 / iters;
}

}  // namespace mi355
