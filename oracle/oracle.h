/*
 * oracle/oracle.h — CPU restatement of the ggml-CPU arithmetic that sits under
 * cortex.llamacpp's llama_decode() call (reference call site:
 * src/llama_server_context.cc:1628-1635).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ may be imported, linked or
 * executed by the product path; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic lives in the third-party module
 * github.com/ggml-org/llama.cpp (submodule `llama.cpp`, branch master,
 * .gitmodules:1-4; contemporaneous tag ~b50xx, 2025-04-04), which is an empty
 * un-vendored directory in /root/reference.  The reference's own tests pin no
 * numeric result for this path.  This file restates the
 * published ggml algorithm (block formats of ggml-common.h, scalar
 * ggml_vec_dot_* of ggml-cpu-quants.c, ops of ggml-cpu) from its specification
 *  and is pinned only by the closed-form known-answer
 * tests in tests/test_oracle_kat.py and an independent numpy twin; its answers
 * of round 3 are frozen as fixtures under tests/golden/ (ops_v1.npz, e2e_v1.npz,
 * written by tests/golden/make_golden_ops.py) so that a later drift of this
 * file is visible (tests/test_golden_ops.py).
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids (upstream: ggml/include/ggml.h enum ggml_type) */
enum {
    OQ_TYPE_F32  = 0,
    OQ_TYPE_F16  = 1,
    OQ_TYPE_Q4_0 = 2,
    OQ_TYPE_Q5_0 = 6,
    OQ_TYPE_Q8_0 = 8,
    OQ_TYPE_Q2_K = 10,
    OQ_TYPE_Q3_K = 11,
    OQ_TYPE_Q4_K = 12,
    OQ_TYPE_Q5_K = 13,
    OQ_TYPE_Q6_K = 14,
    OQ_TYPE_Q8_K = 15,
    OQ_TYPE_IQ4_NL = 20,
};

#define OQ_QK_K 256
#define OQ_QK8_0 32
#define OQ_QK4_0 32

/* ---- type geometry ---------------------------------------------------- */
int    oq_block_elems(int type);            /* elements per block (1 for f32/f16) */
size_t oq_block_bytes(int type);            /* bytes per block */
size_t oq_row_bytes(int type, int64_t n);   /* bytes of n elements (n % block == 0) */

/* ---- fp16 -------------------------------------------------------------- */
float    oq_fp16_to_fp32(uint16_t h);
uint16_t oq_fp32_to_fp16(float f);

/* ---- weight-side dequantisation (upstream: dequantize_row_* in ggml-quants.c) */
void oq_dequantize_row(int type, const void *src, float *dst, int64_t n);

/* ---- activation / KV quantisation (upstream: quantize_row_*_ref) ------- */
void oq_quantize_row_q8_0(const float *x, void *dst, int64_t n);
void oq_quantize_row_q8_K(const float *x, void *dst, int64_t n);
void oq_quantize_row_q4_0(const float *x, void *dst, int64_t n);
void oq_quantize_row(int type, const float *x, void *dst, int64_t n); /* f16,q8_0,q4_0,q8_K,f32 */

/* ---- dot products (upstream: scalar ggml_vec_dot_* in ggml-cpu-quants.c) */
/* vec_dot_type(type): the activation format the CPU backend pairs with `type` */
int   oq_vec_dot_type(int type);
float oq_vec_dot(int type, int64_t n, const void *w_row, const void *act_q);
/* integer partials per super-block, for bit-exact checks of the HIP kernels:
 * isum[b] = sum_j sc_j * sum(q*q8)   and   msum[b] = sum_j m_j * bsums  (0 for q6_K/q8_0) */
void  oq_vec_dot_int_partials(int type, int64_t n, const void *w_row, const void *act_q,
                              int32_t *isum, int32_t *msum);

/* calibration knobs, tests only (default 0 = reference behaviour) */
void oq_set_assoc_variant(int v);   /* 1: re-associate the f32 sums of the K-quant dot products */
void oq_set_fa_v_acc_f32(int v);    /* 1: flash-attn accumulates an f16 V cache in f32 instead of fp16 */

/* ---- ops (upstream: ggml-cpu ops) -------------------------------------- */
/* y[N,T] = W[N,K] . x[K,T]; x is f32 [T][K], y is f32 [T][N]; nth = threads */
void oq_mul_mat(int type, const void *W, int64_t N, int64_t K,
                const float *x, int64_t T, float *y, int nth);
void oq_rms_norm(const float *x, float *y, int64_t n, float eps);
void oq_mul_f32(const float *a, const float *b, float *y, int64_t n);
void oq_add_f32(const float *a, const float *b, float *y, int64_t n);
void oq_silu_f32(const float *x, float *y, int64_t n);
void oq_soft_max(const float *x, const float *mask /*nullable*/, float *y, int64_t n, float scale);
void oq_moe_route(const float *logits, int n_expert, int k, float *probs /* scratch [n_expert] */, int32_t *ids, float *w);
/* YaRN parameters of a rotation (NULL where one is taken: none); oq_yarn_corr_dims gives corr_lo / corr_hi the way the context does (beta_fast 32, beta_slow 1) */
typedef struct { float ext_factor, attn_factor, corr_lo, corr_hi; } oq_yarn;
void oq_yarn_corr_dims(int n_rot, int n_ctx_orig, float freq_base, float beta_fast, float beta_slow, float *lo, float *hi);
void oq_rope_ext(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                 float freq_base, float freq_scale, const float *ff, int neox, const oq_yarn *y);
/* rope, NORM pairing (x[2i],x[2i+1]); x is [n_head][head_dim] for one token */
void oq_rope_norm(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                  float freq_base, float freq_scale, const float *freq_factors /*nullable*/);
/* rope NEOX pairing (x[i], x[i+n_rot/2]) */
void oq_rope_neox(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                  float freq_base, float freq_scale, const float *freq_factors /*nullable*/);
void oq_get_rows(int type, const void *table, int64_t row_elems, const int32_t *ids, int64_t n_ids, float *dst);

/* flash_attn_ext for ONE query token over ONE sequence's cells.
 * q:   f32 [n_head][dk]
 * k,v: cache rows of type type_k/type_v; cell c, kv-head g lives at
 *      base + c*row_stride + g*head_stride (bytes)
 * cells[n_cells]: cache cell indices visible to this query (mask != -inf), ascending
 * out: f32 [n_head][dv] */
void oq_flash_attn_ext(const float *q, int n_head, int n_head_kv, int dk, int dv,
                       int type_k, const void *k, size_t k_row_stride, size_t k_head_stride,
                       int type_v, const void *v, size_t v_row_stride, size_t v_head_stride,
                       const int32_t *cells, int n_cells, float scale, float *out);

/* ---- GGUF + llama-arch forward ----------------------------------------- */
typedef struct oq_model oq_model;
typedef struct oq_ctx   oq_ctx;

oq_model *oq_model_load(const char *path);         /* NULL on error */
void      oq_model_free(oq_model *m);
int       oq_model_n_vocab(const oq_model *m);
int       oq_model_n_embd(const oq_model *m);
int       oq_model_n_layer(const oq_model *m);

/* type_k/type_v: OQ_TYPE_F16 / Q8_0 / Q4_0; flash_attn 0 => softmax(KQ)V path (f16 only) */
/* test hook (mixture-of-experts files): record the expert ids the router selects from now on, in call order (decode call -> layer -> token -> rank) */
void   oq_moe_record_start(size_t cap);
size_t oq_moe_record_get(int32_t *out, size_t cap);     /* returns the number recorded so far */
oq_ctx *oq_ctx_new(oq_model *m, int n_ctx, int type_k, int type_v, int flash_attn, int n_threads);
void    oq_ctx_free(oq_ctx *c);
/* llama_decode equivalent: appends n tokens of sequence seq at pos[]; writes logits rows
 * for tokens with want_logits[i] != 0 (NULL: last only), packed in order.  0 ok, 1 no KV space. */
int     oq_decode(oq_ctx *c, const int32_t *tokens, const int32_t *pos, const int32_t *seq,
                  const int8_t *want_logits, int n, float *logits_out);
/* the same with embedding rows [n][n_embd] in place of token ids (llama_batch.embd) */
int     oq_decode_embd(oq_ctx *c, const float *embd, const int32_t *pos, const int32_t *seq,
                       const int8_t *want_logits, int n, float *logits_out);
void    oq_kv_clear(oq_ctx *c);
int     oq_kv_seq_rm(oq_ctx *c, int seq, int p0, int p1);
void    oq_kv_seq_cp(oq_ctx *c, int seq_src, int seq_dst, int p0, int p1);
void    oq_kv_seq_add(oq_ctx *c, int seq, int p0, int p1, int delta);
/* debugging taps: copy of the residual stream after layer il for the last decoded batch */
const float *oq_debug_layer_out(oq_ctx *c, int il);

/* ---- LLaVA image path (oq_clip.c): clip_image_preprocess + clip_image_encode of a LLaVA-1.5 style projector file ---- */
typedef struct oq_clip oq_clip;
oq_clip *oq_clip_load(const char *path);
void     oq_clip_free(oq_clip *c);
int      oq_clip_image_size(const oq_clip *c);
int      oq_clip_n_patches(const oq_clip *c);
int      oq_clip_n_mmproj_embd(const oq_clip *c);
/* LLaVA-1.6 (image grid): the most rows a picture can yield; every image the encoder sees for a picture (out [n][3][S][S], returns n, grid_w x grid_h tiles);
 * and the rows of a picture (overview first, then the tiles' rows in the canvas' row-major order; returns the row count or < 0) */
int      oq_clip_max_image_rows(const oq_clip *c);
int      oq_clip_preprocess_all(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out, int cap_images, int *grid_w, int *grid_h);
int      oq_clip_embed(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out, int cap_rows, int n_threads);
/* rgb: [ny][nx][3] bytes -> out: [3][S][S] normalised floats */
void     oq_clip_preprocess(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out);
/* img: [3][S][S] -> out: [n_patches][n_mmproj_embd] */
int      oq_clip_encode(const oq_clip *c, const float *img, float *out, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* ORACLE_H */
