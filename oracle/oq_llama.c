/*
 * oracle/oq_llama.c — GGUF v3 reader + `llama`-architecture forward pass + KV cell bookkeeping,
 * i.e. what llama_decode() does for the reference at src/llama_server_context.cc:1635, on CPU.
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle.h).
 *
 * Upstream homes (absent from /root/reference; restated from SURVEY.md §A.3/§A.4):
 *   ggml/src/gguf.cpp              container
 *   src/llama-model.cpp            llm_build_llama op order
 *   src/llama-graph.cpp            build_attn / build_ffn / build_moe_ffn
 *   src/llama-kv-cache.cpp         cells, seq_rm / seq_cp / seq_add (+ K re-rotation)
 * Reference call sites for the cache ops: src/llama_server_context.cc:287,661,1288-1291,1540-1547.
 */
#include "oracle.h"

#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#define MAX_LAYERS 128

typedef struct {
    char name[96];
    int type;
    int n_dims;
    int64_t ne[4];
    const uint8_t *data;
} oq_tensor;

typedef struct {
    const oq_tensor *attn_norm, *wq, *wk, *wv, *wo, *bq, *bk, *bv;
    const oq_tensor *ffn_norm, *gate, *up, *down;
    const oq_tensor *gate_inp, *gate_exps, *up_exps, *down_exps;
    /* nomic-bert (encoder): fused Q | K | V projection, LayerNorm (weight + bias) AFTER the attention and after the feed-forward block */
    const oq_tensor *wqkv, *bo, *attn_out_norm, *attn_out_norm_b, *layer_out_norm, *layer_out_norm_b;
} oq_layer;

struct oq_model {
    int fd;
    uint8_t *map;
    size_t map_size;
    int n_tensors;
    oq_tensor *tensors;
    char arch[32];
    int n_embd, n_layer, n_ff, n_head, n_head_kv, n_rot, n_vocab, n_expert, n_expert_used;
    int head_dim;
    float eps, rope_base, rope_scale;
    int rope_neox;
    /* {arch}.rope.scaling.*: type "linear" divides the angles by factor; "yarn" also blends per pair (oq_ops.c rope_table) */
    char rope_scaling[16];
    float rope_factor, yarn_attn_factor;
    int n_ctx_train, n_ctx_orig;
    oq_yarn yarn;
    const oq_tensor *tok_embd, *out_norm, *output, *rope_freqs;
    int is_bert;                  /* general.architecture nomic-bert: bidirectional encoder, embeddings only */
    const oq_tensor *tok_types, *tok_norm, *tok_norm_b;
    oq_layer layers[MAX_LAYERS];
};

/* ---------------------------------------------------------------- gguf parse */
typedef struct { const uint8_t *p, *end; int bad; } rd;
static uint64_t rd_u(rd *r, int n) {
    uint64_t v = 0;
    if (r->p + n > r->end) { r->bad = 1; return 0; }
    memcpy(&v, r->p, (size_t)n);
    r->p += n;
    return v;
}
static void rd_str(rd *r, char *dst, size_t cap) {
    uint64_t n = rd_u(r, 8);
    if (r->bad || r->p + n > r->end) { r->bad = 1; if (cap) dst[0] = 0; return; }
    size_t c = n < cap - 1 ? (size_t)n : cap - 1;
    memcpy(dst, r->p, c);
    dst[c] = 0;
    r->p += n;
}
static const int gguf_scalar_size[13] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};

typedef struct { int type; uint64_t u; double f; char s[64]; } kvval;

static void rd_value(rd *r, int type, kvval *out) {
    out->type = type; out->u = 0; out->f = 0; out->s[0] = 0;
    if (type == 8) { rd_str(r, out->s, sizeof out->s); return; }
    if (type == 9) {
        int et = (int)rd_u(r, 4);
        uint64_t n = rd_u(r, 8);
        out->u = n;
        for (uint64_t i = 0; i < n && !r->bad; i++) {
            if (et == 8) { uint64_t l = rd_u(r, 8); if (r->p + l > r->end) r->bad = 1; else r->p += l; }
            else if (et >= 0 && et < 13 && gguf_scalar_size[et]) r->p += gguf_scalar_size[et];
            else r->bad = 1;
        }
        return;
    }
    if (type < 0 || type > 12) { r->bad = 1; return; }
    uint64_t raw = rd_u(r, gguf_scalar_size[type]);
    out->u = raw;
    switch (type) {
        case 1: out->f = (int8_t)raw; out->u = (uint64_t)(int64_t)(int8_t)raw; break;
        case 3: out->f = (int16_t)raw; out->u = (uint64_t)(int64_t)(int16_t)raw; break;
        case 5: out->f = (int32_t)raw; out->u = (uint64_t)(int64_t)(int32_t)raw; break;
        case 6: { float f; uint32_t b = (uint32_t)raw; memcpy(&f, &b, 4); out->f = f; break; }
        case 12: { double d; memcpy(&d, &raw, 8); out->f = d; break; }
        default: out->f = (double)raw;
    }
}

static const oq_tensor *find_tensor(const oq_model *m, const char *name) {
    for (int i = 0; i < m->n_tensors; i++)
        if (!strcmp(m->tensors[i].name, name)) return &m->tensors[i];
    return NULL;
}
static const oq_tensor *layer_tensor(const oq_model *m, int il, const char *suffix) {
    char nm[96];
    snprintf(nm, sizeof nm, "blk.%d.%s", il, suffix);
    return find_tensor(m, nm);
}

oq_model *oq_model_load(const char *path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return NULL; }
    uint8_t *map = (uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return NULL; }
    oq_model *m = (oq_model *)calloc(1, sizeof *m);
    m->fd = fd; m->map = map; m->map_size = (size_t)st.st_size;
    m->eps = 1e-5f; m->rope_base = 10000.0f; m->rope_scale = 1.0f;
    m->rope_factor = 1.0f; m->yarn_attn_factor = 1.0f; m->yarn.attn_factor = 1.0f;
    rd r = {map, map + st.st_size, 0};
    uint32_t magic = (uint32_t)rd_u(&r, 4), version = (uint32_t)rd_u(&r, 4);
    uint64_t n_tensors = rd_u(&r, 8), n_kv = rd_u(&r, 8);
    if (magic != 0x46554747u || version < 2 || version > 3) { oq_model_free(m); return NULL; }
    uint64_t alignment = 32;
    char key[160];
    /* two passes over the kv section would need the arch first; keys carry the arch prefix, so match suffixes */
    for (uint64_t i = 0; i < n_kv && !r.bad; i++) {
        rd_str(&r, key, sizeof key);
        int type = (int)rd_u(&r, 4);
        kvval v;
        rd_value(&r, type, &v);
        const char *dot = strchr(key, '.');
        const char *suf = dot ? dot + 1 : key;
        if (!strcmp(key, "general.architecture")) snprintf(m->arch, sizeof m->arch, "%s", v.s);
        else if (!strcmp(key, "general.alignment")) alignment = v.u;
        else if (!strncmp(key, "general.", 8) || !strncmp(key, "tokenizer.", 10)) continue;
        else if (!strcmp(suf, "embedding_length")) m->n_embd = (int)v.u;
        else if (!strcmp(suf, "block_count")) m->n_layer = (int)v.u;
        else if (!strcmp(suf, "feed_forward_length")) m->n_ff = (int)v.u;
        else if (!strcmp(suf, "attention.head_count")) m->n_head = (int)v.u;
        else if (!strcmp(suf, "attention.head_count_kv")) m->n_head_kv = (int)v.u;
        else if (!strcmp(suf, "attention.layer_norm_rms_epsilon")) m->eps = (float)v.f;
        else if (!strcmp(suf, "attention.layer_norm_epsilon")) m->eps = (float)v.f;
        else if (!strcmp(suf, "rope.dimension_count")) m->n_rot = (int)v.u;
        else if (!strcmp(suf, "rope.freq_base")) m->rope_base = (float)v.f;
        else if (!strcmp(suf, "rope.scaling.type")) snprintf(m->rope_scaling, sizeof m->rope_scaling, "%s", v.s);
        else if (!strcmp(suf, "rope.scaling.factor")) m->rope_factor = (float)v.f;
        else if (!strcmp(suf, "rope.scaling.attn_factor")) m->yarn_attn_factor = (float)v.f;
        else if (!strcmp(suf, "rope.scaling.original_context_length")) m->n_ctx_orig = (int)v.u;
        else if (!strcmp(suf, "context_length")) m->n_ctx_train = (int)v.u;
        else if (!strcmp(suf, "expert_count")) m->n_expert = (int)v.u;
        else if (!strcmp(suf, "expert_used_count")) m->n_expert_used = (int)v.u;
    }
    m->n_tensors = (int)n_tensors;
    m->tensors = (oq_tensor *)calloc(n_tensors ? n_tensors : 1, sizeof(oq_tensor));
    uint64_t *offs = (uint64_t *)calloc(n_tensors ? n_tensors : 1, 8);
    for (uint64_t i = 0; i < n_tensors && !r.bad; i++) {
        oq_tensor *t = &m->tensors[i];
        rd_str(&r, t->name, sizeof t->name);
        t->n_dims = (int)rd_u(&r, 4);
        for (int d = 0; d < 4; d++) t->ne[d] = 1;
        for (int d = 0; d < t->n_dims && d < 4; d++) t->ne[d] = (int64_t)rd_u(&r, 8);
        t->type = (int)rd_u(&r, 4);
        offs[i] = rd_u(&r, 8);
    }
    if (r.bad || m->n_layer > MAX_LAYERS) { free(offs); oq_model_free(m); return NULL; }
    uint64_t data_off = (uint64_t)(r.p - map);
    data_off = (data_off + alignment - 1) / alignment * alignment;
    for (uint64_t i = 0; i < n_tensors; i++) m->tensors[i].data = map + data_off + offs[i];
    free(offs);

    m->tok_embd = find_tensor(m, "token_embd.weight");
    m->out_norm = find_tensor(m, "output_norm.weight");
    m->output = find_tensor(m, "output.weight");
    if (!m->output) m->output = m->tok_embd; /* tied embeddings */
    m->rope_freqs = find_tensor(m, "rope_freqs.weight");
    m->is_bert = !strcmp(m->arch, "nomic-bert");
    m->tok_types = find_tensor(m, "token_types.weight");
    m->tok_norm = find_tensor(m, "token_embd_norm.weight");
    m->tok_norm_b = find_tensor(m, "token_embd_norm.bias");
    if (!m->tok_embd || (!m->out_norm && !m->is_bert) || !m->n_head || !m->n_embd) { oq_model_free(m); return NULL; }
    if (m->is_bert && (!m->tok_norm || !m->tok_norm_b)) { oq_model_free(m); return NULL; }
    m->n_vocab = (int)m->tok_embd->ne[1];
    if (!m->n_head_kv) m->n_head_kv = m->n_head;
    m->head_dim = m->n_embd / m->n_head;
    if (!m->n_rot) m->n_rot = m->head_dim;
    m->rope_neox = !(strcmp(m->arch, "llama") == 0);
    /* upstream: src/llama-model.cpp (rope scaling hparams) + src/llama-context.cpp (freq_scale = 1 / factor; yarn: ext_factor 1, beta_fast 32, beta_slow 1) */
    if (!strcmp(m->rope_scaling, "linear") || !strcmp(m->rope_scaling, "yarn")) m->rope_scale = 1.0f / m->rope_factor;
    if (!strcmp(m->rope_scaling, "yarn")) {
        m->yarn.ext_factor = 1.0f;
        m->yarn.attn_factor = m->yarn_attn_factor;
        oq_yarn_corr_dims(m->n_rot, m->n_ctx_orig ? m->n_ctx_orig : (m->n_ctx_train ? m->n_ctx_train : 4096), m->rope_base, 32.0f, 1.0f,
                          &m->yarn.corr_lo, &m->yarn.corr_hi);
    }
    for (int il = 0; il < m->n_layer; il++) {
        oq_layer *L = &m->layers[il];
        L->attn_norm = layer_tensor(m, il, "attn_norm.weight");
        L->wq = layer_tensor(m, il, "attn_q.weight");
        L->wk = layer_tensor(m, il, "attn_k.weight");
        L->wv = layer_tensor(m, il, "attn_v.weight");
        L->wo = layer_tensor(m, il, "attn_output.weight");
        L->bq = layer_tensor(m, il, "attn_q.bias");
        L->bk = layer_tensor(m, il, "attn_k.bias");
        L->bv = layer_tensor(m, il, "attn_v.bias");
        L->ffn_norm = layer_tensor(m, il, "ffn_norm.weight");
        L->gate = layer_tensor(m, il, "ffn_gate.weight");
        L->up = layer_tensor(m, il, "ffn_up.weight");
        L->down = layer_tensor(m, il, "ffn_down.weight");
        L->wqkv = layer_tensor(m, il, "attn_qkv.weight");
        L->bo = layer_tensor(m, il, "attn_output.bias");
        L->attn_out_norm = layer_tensor(m, il, "attn_output_norm.weight");
        L->attn_out_norm_b = layer_tensor(m, il, "attn_output_norm.bias");
        L->layer_out_norm = layer_tensor(m, il, "layer_output_norm.weight");
        L->layer_out_norm_b = layer_tensor(m, il, "layer_output_norm.bias");
        L->gate_inp = layer_tensor(m, il, "ffn_gate_inp.weight");
        L->gate_exps = layer_tensor(m, il, "ffn_gate_exps.weight");
        L->up_exps = layer_tensor(m, il, "ffn_up_exps.weight");
        L->down_exps = layer_tensor(m, il, "ffn_down_exps.weight");
        if (m->is_bert) {
            if (!L->wqkv || !L->wo || !L->attn_out_norm || !L->attn_out_norm_b || !L->layer_out_norm || !L->layer_out_norm_b || !L->gate || !L->up || !L->down) { oq_model_free(m); return NULL; }
            continue;
        }
        if (!L->attn_norm || !L->wq || !L->wk || !L->wv || !L->wo || !L->ffn_norm) { oq_model_free(m); return NULL; }
        if (!L->gate_inp && (!L->gate || !L->up || !L->down)) { oq_model_free(m); return NULL; }
    }
    return m;
}

void oq_model_free(oq_model *m) {
    if (!m) return;
    if (m->map) munmap(m->map, m->map_size);
    if (m->fd >= 0) close(m->fd);
    free(m->tensors);
    free(m);
}
int oq_model_n_vocab(const oq_model *m) { return m->n_vocab; }
int oq_model_n_embd(const oq_model *m) { return m->n_embd; }
int oq_model_n_layer(const oq_model *m) { return m->n_layer; }

/* ---------------------------------------------------------------- context */
typedef struct { int32_t pos; int32_t delta; uint64_t seqs; } oq_cell;

struct oq_ctx {
    oq_model *m;
    int n_ctx, type_k, type_v, flash_attn, nth;
    int head;
    int has_shift;
    oq_cell *cells;
    size_t k_row, v_row;    /* bytes of one cell's K / V row (all kv heads) */
    uint8_t **k, **v;       /* per layer [n_ctx][k_row] ; non-FA V is stored transposed f16 [n_embd_v][n_ctx] */
    float **dbg;            /* per-layer residual snapshots of the last batch */
    int dbg_tokens;
};

oq_ctx *oq_ctx_new(oq_model *m, int n_ctx, int type_k, int type_v, int flash_attn, int nth) {
    if (!flash_attn && (type_k != OQ_TYPE_F16 || type_v != OQ_TYPE_F16)) return NULL;
    oq_ctx *c = (oq_ctx *)calloc(1, sizeof *c);
    c->m = m; c->n_ctx = n_ctx; c->type_k = type_k; c->type_v = type_v; c->flash_attn = flash_attn;
    c->nth = nth < 1 ? 1 : nth;
    c->cells = (oq_cell *)calloc((size_t)n_ctx, sizeof(oq_cell));
    for (int i = 0; i < n_ctx; i++) c->cells[i].pos = -1;
    const int64_t kv_dim = (int64_t)m->n_head_kv * m->head_dim;
    c->k_row = oq_row_bytes(type_k, kv_dim);
    c->v_row = oq_row_bytes(type_v, kv_dim);
    c->k = (uint8_t **)calloc((size_t)m->n_layer, sizeof(void *));
    c->v = (uint8_t **)calloc((size_t)m->n_layer, sizeof(void *));
    c->dbg = (float **)calloc((size_t)m->n_layer, sizeof(void *));
    for (int il = 0; il < m->n_layer; il++) {
        c->k[il] = (uint8_t *)calloc((size_t)n_ctx, c->k_row);
        c->v[il] = (uint8_t *)calloc((size_t)n_ctx, c->v_row);
    }
    return c;
}
void oq_ctx_free(oq_ctx *c) {
    if (!c) return;
    for (int il = 0; il < c->m->n_layer; il++) { free(c->k[il]); free(c->v[il]); free(c->dbg[il]); }
    free(c->k); free(c->v); free(c->dbg); free(c->cells); free(c);
}
const float *oq_debug_layer_out(oq_ctx *c, int il) { return c->dbg[il]; }

void oq_kv_clear(oq_ctx *c) {
    for (int i = 0; i < c->n_ctx; i++) { c->cells[i].pos = -1; c->cells[i].seqs = 0; c->cells[i].delta = 0; }
    c->head = 0;
}
int oq_kv_seq_rm(oq_ctx *c, int seq, int p0, int p1) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    int new_head = c->n_ctx;
    for (int i = 0; i < c->n_ctx; i++) {
        oq_cell *ce = &c->cells[i];
        if (ce->pos < p0 || ce->pos >= p1) continue;
        if (seq < 0) ce->seqs = 0;
        else if (ce->seqs & (1ull << seq)) ce->seqs &= ~(1ull << seq);
        else continue;
        if (!ce->seqs) { ce->pos = -1; ce->delta = 0; if (i < new_head) new_head = i; }
    }
    if (new_head < c->n_ctx && new_head < c->head) c->head = new_head;
    return 1;
}
void oq_kv_seq_cp(oq_ctx *c, int src, int dst, int p0, int p1) {
    if (src == dst) return;
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    for (int i = 0; i < c->n_ctx; i++) {
        oq_cell *ce = &c->cells[i];
        if ((ce->seqs & (1ull << src)) && ce->pos >= p0 && ce->pos < p1) ce->seqs |= 1ull << dst;
    }
}
void oq_kv_seq_add(oq_ctx *c, int seq, int p0, int p1, int delta) {
    if (p0 < 0) p0 = 0;
    if (p1 < 0) p1 = 0x7fffffff;
    if (p0 == p1 || delta == 0) return;
    for (int i = 0; i < c->n_ctx; i++) {
        oq_cell *ce = &c->cells[i];
        if (!(ce->seqs & (1ull << seq)) || ce->pos < p0 || ce->pos >= p1) continue;
        c->has_shift = 1;
        ce->pos += delta;
        ce->delta += delta;
        if (ce->pos < 0) { ce->pos = -1; ce->seqs = 0; ce->delta = 0; }
    }
}

static void rope_heads(const oq_model *m, float *x, int n_head, int32_t pos) {
    const float *ff = m->rope_freqs ? (const float *)m->rope_freqs->data : NULL;
    oq_rope_ext(x, n_head, m->head_dim, m->n_rot, pos, m->rope_base, m->rope_scale, ff, m->rope_neox, &m->yarn);
}

/* re-rotate cached K rows whose position was shifted by seq_add (upstream: K-shift graph) */
static void apply_k_shift(oq_ctx *c) {
    const oq_model *m = c->m;
    const int64_t kv_dim = (int64_t)m->n_head_kv * m->head_dim;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)kv_dim);
    for (int i = 0; i < c->n_ctx; i++) {
        oq_cell *ce = &c->cells[i];
        if (ce->delta == 0) continue;
        for (int il = 0; il < m->n_layer; il++) {
            uint8_t *row = c->k[il] + (size_t)i * c->k_row;
            oq_dequantize_row(c->type_k, row, tmp, kv_dim);
            rope_heads(m, tmp, m->n_head_kv, ce->delta);
            oq_quantize_row(c->type_k, tmp, row, kv_dim);
        }
        ce->delta = 0;
    }
    free(tmp);
    c->has_shift = 0;
}

static int find_slot(oq_ctx *c, int n) {
    if (n > c->n_ctx) return -1;
    int head = c->head, tested = 0;
    while (1) {
        if (head + n > c->n_ctx) { tested += c->n_ctx - head; head = 0; continue; }
        int ok = 1;
        for (int i = 0; i < n; i++)
            if (c->cells[head + i].pos >= 0) { ok = 0; head += i + 1; tested += i + 1; break; }
        if (ok) return head;
        if (tested >= c->n_ctx) return -1;
    }
}

static void linear(const oq_ctx *c, const oq_tensor *w, const float *x, int64_t T, float *y) {
    oq_mul_mat(w->type, w->data, w->ne[1], w->ne[0], x, T, y, c->nth);
}
static void add_bias(const oq_tensor *b, float *y, int64_t T) {
    if (!b) return;
    const int64_t n = b->ne[0];
    for (int64_t t = 0; t < T; t++) oq_add_f32(y + t * n, (const float *)b->data, y + t * n, n);
}

static void ffn_dense(const oq_ctx *c, const oq_layer *L, const float *h, int64_t T, float *out) {
    const oq_model *m = c->m;
    const int64_t F = L->gate->ne[1];
    float *g = (float *)malloc(sizeof(float) * (size_t)(F * T)), *u = (float *)malloc(sizeof(float) * (size_t)(F * T));
    linear(c, L->gate, h, T, g);
    linear(c, L->up, h, T, u);
    oq_silu_f32(g, g, F * T);
    oq_mul_f32(g, u, g, F * T);
    linear(c, L->down, g, T, out);
    (void)m;
    free(g); free(u);
}

/* test hook: the expert ids the router selected, appended in call order (decode call -> layer -> token -> rank), so that a test can hand the same routing
   to the implementation under test (a rounding flip on a near tie of the router otherwise sends a token to another expert on one side only) */
static int32_t *g_moe_rec = NULL;
static size_t g_moe_rec_n = 0, g_moe_rec_cap = 0;
void oq_moe_record_start(size_t cap) {
    free(g_moe_rec);
    g_moe_rec = cap ? (int32_t *)malloc(cap * sizeof(int32_t)) : NULL;
    g_moe_rec_cap = g_moe_rec ? cap : 0;
    g_moe_rec_n = 0;
}
size_t oq_moe_record_get(int32_t *out, size_t cap) {
    const size_t n = g_moe_rec_n < cap ? g_moe_rec_n : cap;
    if (out && n) memcpy(out, g_moe_rec, n * sizeof(int32_t));
    return g_moe_rec_n;
}

/* MoE: softmax router, top-k by probability, renormalised weights, experts summed in rank order */
static void ffn_moe(const oq_ctx *c, const oq_layer *L, const float *h, int64_t T, float *out) {
    const oq_model *m = c->m;
    const int E = m->n_expert, KU = m->n_expert_used, D = m->n_embd;
    const int64_t F = L->gate_exps->ne[1];
    const size_t gu_bytes = oq_row_bytes(L->gate_exps->type, L->gate_exps->ne[0]) * (size_t)F;
    const size_t up_bytes = oq_row_bytes(L->up_exps->type, L->up_exps->ne[0]) * (size_t)F;
    const size_t dn_bytes = oq_row_bytes(L->down_exps->type, L->down_exps->ne[0]) * (size_t)D;
    float *logits = (float *)malloc(sizeof(float) * (size_t)E), *probs = (float *)malloc(sizeof(float) * (size_t)E);
    float *g = (float *)malloc(sizeof(float) * (size_t)F), *u = (float *)malloc(sizeof(float) * (size_t)F);
    float *e_out = (float *)malloc(sizeof(float) * (size_t)D);
    for (int64_t t = 0; t < T; t++) {
        const float *x = h + t * D;
        linear(c, L->gate_inp, x, 1, logits);
        int ids[16]; float wts[16];
        oq_moe_route(logits, E, KU, probs, ids, wts);
        for (int k = 0; k < KU && g_moe_rec_n < g_moe_rec_cap; k++) g_moe_rec[g_moe_rec_n++] = ids[k];
        float *o = out + t * D;
        for (int k = 0; k < KU; k++) {
            const int e = ids[k];
            oq_mul_mat(L->gate_exps->type, L->gate_exps->data + (size_t)e * gu_bytes, F, D, x, 1, g, c->nth);
            oq_mul_mat(L->up_exps->type, L->up_exps->data + (size_t)e * up_bytes, F, D, x, 1, u, c->nth);
            oq_silu_f32(g, g, F);
            oq_mul_f32(g, u, g, F);
            oq_mul_mat(L->down_exps->type, L->down_exps->data + (size_t)e * dn_bytes, D, F, g, 1, e_out, c->nth);
            for (int d = 0; d < D; d++) {
                const float v = e_out[d] * wts[k];
                o[d] = k == 0 ? v : o[d] + v;
            }
        }
    }
    free(logits); free(probs); free(g); free(u); free(e_out);
}

/* LayerNorm with weight and bias (upstream: ggml_compute_forward_norm_f32 then ggml_mul, ggml_add in build_norm(LLM_NORM)): mean and variance summed in double */
static void layer_norm_rows(const float *x, float *y, int64_t n, int64_t T, float eps, const oq_tensor *w, const oq_tensor *b) {
    for (int64_t t = 0; t < T; t++) {
        const float *xr = x + t * n;
        float *yr = y + t * n;
        double sum = 0.0;
        for (int64_t i = 0; i < n; i++) sum += (double)xr[i];
        const float mean = (float)(sum / (double)n);
        double sum2 = 0.0;
        for (int64_t i = 0; i < n; i++) { const float v = xr[i] - mean; yr[i] = v; sum2 += (double)(v * v); }
        const float variance = (float)(sum2 / (double)n);
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int64_t i = 0; i < n; i++) yr[i] *= scale;
        oq_mul_f32(yr, (const float *)w->data, yr, n);
        oq_add_f32(yr, (const float *)b->data, yr, n);
    }
}

/* nomic-bert (upstream: src/llama-model.cpp llm_build_bert, the NOMIC_BERT branches): token + type-0 embeddings -> LayerNorm; per layer fused Q | K | V, NEOX
 * rope on Q and K, BIDIRECTIONAL attention over the sequence's tokens (build_attn_inp_no_cache: every token sees every token of its sequence), attn_output,
 * residual, LayerNorm, SwiGLU feed-forward (LLM_FFN_SILU, LLM_FFN_PAR: down(silu(gate x) * up x)), residual, LayerNorm.  The result is the hidden state of
 * the last layer (t_embd): no output projection.  K / V rows pass through the context's cache rows (f16: what the flash path casts them to anyway). */
static int bert_decode(oq_ctx *c, const int32_t *tokens, const int32_t *pos, const int32_t *seq, int n) {
    const oq_model *m = c->m;
    const int D = m->n_embd, H = m->n_head, G = m->n_head_kv, hd = m->head_dim;
    const int64_t kv_dim = (int64_t)G * hd, qkv_dim = D + 2 * kv_dim;
    const int slot = find_slot(c, n);
    if (slot < 0) return 1;
    for (int i = 0; i < n; i++) {
        c->cells[slot + i].pos = pos[i];
        c->cells[slot + i].seqs = 1ull << (seq ? seq[i] : 0);
        c->cells[slot + i].delta = 0;
    }
    c->head = slot + n;
    if (c->head >= c->n_ctx) c->head = 0;
    float *x = (float *)malloc(sizeof(float) * (size_t)D * n), *cur = (float *)malloc(sizeof(float) * (size_t)D * n);
    float *qkv = (float *)malloc(sizeof(float) * (size_t)qkv_dim * n);
    float *q = (float *)malloc(sizeof(float) * (size_t)D * n), *kk = (float *)malloc(sizeof(float) * (size_t)kv_dim * n), *vv = (float *)malloc(sizeof(float) * (size_t)kv_dim * n);
    float *att = (float *)malloc(sizeof(float) * (size_t)D * n), *tmp = (float *)malloc(sizeof(float) * (size_t)D * n);
    int32_t *vis = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->n_ctx);
    oq_get_rows(m->tok_embd->type, m->tok_embd->data, D, tokens, n, x);
    if (m->tok_types)                                      /* token types are all zero: row 0 of the type table */
        for (int t = 0; t < n; t++) oq_add_f32(x + (size_t)t * D, (const float *)m->tok_types->data, x + (size_t)t * D, D);
    layer_norm_rows(x, x, D, n, m->eps, m->tok_norm, m->tok_norm_b);
    const float kq_scale = 1.0f / sqrtf((float)hd);
    for (int il = 0; il < m->n_layer; il++) {
        const oq_layer *L = &m->layers[il];
        linear(c, L->wqkv, x, n, qkv);
        for (int t = 0; t < n; t++) {
            memcpy(q + (size_t)t * D, qkv + (size_t)t * qkv_dim, sizeof(float) * (size_t)D);
            memcpy(kk + (size_t)t * kv_dim, qkv + (size_t)t * qkv_dim + D, sizeof(float) * (size_t)kv_dim);
            memcpy(vv + (size_t)t * kv_dim, qkv + (size_t)t * qkv_dim + D + kv_dim, sizeof(float) * (size_t)kv_dim);
            rope_heads(m, q + (size_t)t * D, H, pos[t]);
            rope_heads(m, kk + (size_t)t * kv_dim, G, pos[t]);
            oq_quantize_row(c->type_k, kk + (size_t)t * kv_dim, c->k[il] + (size_t)(slot + t) * c->k_row, kv_dim);
            oq_quantize_row(c->type_v, vv + (size_t)t * kv_dim, c->v[il] + (size_t)(slot + t) * c->v_row, kv_dim);
        }
        const size_t k_head = oq_row_bytes(c->type_k, hd), v_head = oq_row_bytes(c->type_v, hd);
        for (int t = 0; t < n; t++) {
            const int s = seq ? seq[t] : 0;
            int nv = 0;
            for (int i = 0; i < c->n_ctx; i++)
                if (c->cells[i].pos >= 0 && (c->cells[i].seqs & (1ull << s))) vis[nv++] = i;       /* no causal test */
            oq_flash_attn_ext(q + (size_t)t * D, H, G, hd, hd, c->type_k, c->k[il], c->k_row, k_head,
                              c->type_v, c->v[il], c->v_row, v_head, vis, nv, kq_scale, att + (size_t)t * D);
        }
        linear(c, L->wo, att, n, tmp);
        add_bias(L->bo, tmp, n);
        oq_add_f32(tmp, x, cur, (int64_t)D * n);                                    /* re-add the layer input */
        layer_norm_rows(cur, cur, D, n, m->eps, L->attn_out_norm, L->attn_out_norm_b);
        ffn_dense(c, L, cur, n, tmp);
        oq_add_f32(tmp, cur, cur, (int64_t)D * n);                                  /* the attention output bypasses the feed-forward block */
        layer_norm_rows(cur, x, D, n, m->eps, L->layer_out_norm, L->layer_out_norm_b);
        c->dbg[il] = (float *)realloc(c->dbg[il], sizeof(float) * (size_t)D * n);
        memcpy(c->dbg[il], x, sizeof(float) * (size_t)D * n);
    }
    c->dbg_tokens = n;
    free(x); free(cur); free(qkv); free(q); free(kk); free(vv); free(att); free(tmp); free(vis);
    return 0;
}

/* llama_decode on a batch whose rows are embeddings (llama_batch.embd: the image embeddings of a LLaVA request,
 * /root/reference/src/llama_server_context.cc:1093-1107): the rows replace the token-embedding lookup, everything else is oq_decode */
static const float *g_embd_rows = NULL;
int oq_decode_embd(oq_ctx *c, const float *embd, const int32_t *pos, const int32_t *seq, const int8_t *want, int n, float *logits_out) {
    if (c->m->is_bert || !embd) return -1;
    int32_t *zeros = (int32_t *)calloc((size_t)n, sizeof(int32_t));
    g_embd_rows = embd;
    const int rc = oq_decode(c, zeros, pos, seq, want, n, logits_out);
    g_embd_rows = NULL;
    free(zeros);
    return rc;
}

int oq_decode(oq_ctx *c, const int32_t *tokens, const int32_t *pos, const int32_t *seq,
              const int8_t *want, int n, float *logits_out) {
    const oq_model *m = c->m;
    if (m->is_bert) {                                          /* encoder: no logits (rows of zeros); the embeddings are the last layer's tap */
        int rows = 0;
        for (int t = 0; t < n; t++) rows += want ? (want[t] != 0) : (t == n - 1);
        if (logits_out) memset(logits_out, 0, sizeof(float) * (size_t)rows * (size_t)m->n_vocab);
        return bert_decode(c, tokens, pos, seq, n);
    }
    const int D = m->n_embd, H = m->n_head, G = m->n_head_kv, hd = m->head_dim;
    const int64_t kv_dim = (int64_t)G * hd;
    if (c->has_shift) apply_k_shift(c);
    const int slot = find_slot(c, n);
    if (slot < 0) return 1;
    for (int i = 0; i < n; i++) {
        c->cells[slot + i].pos = pos[i];
        c->cells[slot + i].seqs = 1ull << (seq ? seq[i] : 0);
        c->cells[slot + i].delta = 0;
    }
    c->head = slot + n;
    if (c->head >= c->n_ctx) c->head = 0;

    float *x = (float *)malloc(sizeof(float) * (size_t)D * n);
    float *hbuf = (float *)malloc(sizeof(float) * (size_t)D * n);
    float *q = (float *)malloc(sizeof(float) * (size_t)D * n);
    float *kk = (float *)malloc(sizeof(float) * (size_t)kv_dim * n);
    float *vv = (float *)malloc(sizeof(float) * (size_t)kv_dim * n);
    float *att = (float *)malloc(sizeof(float) * (size_t)D * n);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)D * n);
    int32_t *vis = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->n_ctx);
    if (g_embd_rows) memcpy(x, g_embd_rows, sizeof(float) * (size_t)D * n);
    else oq_get_rows(m->tok_embd->type, m->tok_embd->data, D, tokens, n, x);
    const float kq_scale = 1.0f / sqrtf((float)hd);

    for (int il = 0; il < m->n_layer; il++) {
        const oq_layer *L = &m->layers[il];
        for (int t = 0; t < n; t++) {
            oq_rms_norm(x + (size_t)t * D, hbuf + (size_t)t * D, D, m->eps);
            oq_mul_f32(hbuf + (size_t)t * D, (const float *)L->attn_norm->data, hbuf + (size_t)t * D, D);
        }
        linear(c, L->wq, hbuf, n, q);   add_bias(L->bq, q, n);
        linear(c, L->wk, hbuf, n, kk);  add_bias(L->bk, kk, n);
        linear(c, L->wv, hbuf, n, vv);  add_bias(L->bv, vv, n);
        for (int t = 0; t < n; t++) {
            rope_heads(m, q + (size_t)t * D, H, pos[t]);
            rope_heads(m, kk + (size_t)t * kv_dim, G, pos[t]);
            oq_quantize_row(c->type_k, kk + (size_t)t * kv_dim, c->k[il] + (size_t)(slot + t) * c->k_row, kv_dim);
            oq_quantize_row(c->type_v, vv + (size_t)t * kv_dim, c->v[il] + (size_t)(slot + t) * c->v_row, kv_dim);
        }
        const size_t k_head = oq_row_bytes(c->type_k, hd), v_head = oq_row_bytes(c->type_v, hd);
        /* every K / V row of the batch is in the cache: the tokens' attention calls are independent of each other (each reads the cache and writes its own
         * output row), so a prompt batch spreads them over the threads - the same calls, the same bits; a 3968-token prompt otherwise spends half an hour here */
#pragma omp parallel num_threads(c->nth) if (n > 1)
        {
        int32_t *vis = (int32_t *)malloc(sizeof(int32_t) * (size_t)c->n_ctx);
#pragma omp for schedule(dynamic, 1)
        for (int t = 0; t < n; t++) {
            const int s = seq ? seq[t] : 0;
            int nv = 0;
            for (int i = 0; i < c->n_ctx; i++)
                if (c->cells[i].pos >= 0 && (c->cells[i].seqs & (1ull << s)) && c->cells[i].pos <= pos[t]) vis[nv++] = i;
            if (c->flash_attn) {
                oq_flash_attn_ext(q + (size_t)t * D, H, G, hd, hd, c->type_k, c->k[il], c->k_row, k_head,
                                  c->type_v, c->v[il], c->v_row, v_head, vis, nv, kq_scale, att + (size_t)t * D);
            } else {
                /* softmax(K^T q * scale) then V.p, both through the f16 vec_dot (activations cast to f16) */
                float *sc = (float *)malloc(sizeof(float) * (size_t)nv);
                uint16_t *q16 = (uint16_t *)malloc(2 * (size_t)hd), *p16 = (uint16_t *)malloc(2 * (size_t)nv);
                uint16_t *vcol = (uint16_t *)malloc(2 * (size_t)nv);
                for (int h = 0; h < H; h++) {
                    const int g = h / (H / G);
                    oq_quantize_row(OQ_TYPE_F16, q + (size_t)t * D + (size_t)h * hd, q16, hd);
                    for (int j = 0; j < nv; j++)
                        sc[j] = oq_vec_dot(OQ_TYPE_F16, hd, c->k[il] + (size_t)vis[j] * c->k_row + (size_t)g * k_head, q16);
                    oq_soft_max(sc, NULL, sc, nv, kq_scale);
                    oq_quantize_row(OQ_TYPE_F16, sc, p16, nv);
                    for (int d = 0; d < hd; d++) {
                        for (int j = 0; j < nv; j++)
                            vcol[j] = ((const uint16_t *)(c->v[il] + (size_t)vis[j] * c->v_row + (size_t)g * v_head))[d];
                        att[(size_t)t * D + (size_t)h * hd + d] = oq_vec_dot(OQ_TYPE_F16, nv, vcol, p16);
                    }
                }
                free(sc); free(q16); free(p16); free(vcol);
            }
        }
        free(vis);
        }
        linear(c, L->wo, att, n, tmp);
        oq_add_f32(x, tmp, x, (int64_t)D * n);
        for (int t = 0; t < n; t++) {
            oq_rms_norm(x + (size_t)t * D, hbuf + (size_t)t * D, D, m->eps);
            oq_mul_f32(hbuf + (size_t)t * D, (const float *)L->ffn_norm->data, hbuf + (size_t)t * D, D);
        }
        if (L->gate_inp) ffn_moe(c, L, hbuf, n, tmp);
        else ffn_dense(c, L, hbuf, n, tmp);
        oq_add_f32(x, tmp, x, (int64_t)D * n);
        c->dbg[il] = (float *)realloc(c->dbg[il], sizeof(float) * (size_t)D * n);
        memcpy(c->dbg[il], x, sizeof(float) * (size_t)D * n);
    }
    c->dbg_tokens = n;

    int row = 0;
    for (int t = 0; t < n; t++) {
        const int w = want ? want[t] : (t == n - 1);
        if (!w) continue;
        oq_rms_norm(x + (size_t)t * D, hbuf, D, m->eps);
        oq_mul_f32(hbuf, (const float *)m->out_norm->data, hbuf, D);
        linear(c, m->output, hbuf, 1, logits_out + (size_t)row * m->n_vocab);
        row++;
    }
    free(x); free(hbuf); free(q); free(kk); free(vv); free(att); free(tmp); free(vis);
    return 0;
}
