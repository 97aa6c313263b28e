/*
 * include/mi355_llama.h — C-ABI of the MI355X-native GGUF inference backend.
 *
 * This is the drop-in boundary for the ONE hot path of cortex.llamacpp: everything the
 * reference's continuous-batching loop (src/llama_server_context.cc, LlamaServerContext) calls
 * in llama.cpp's `llama.h` to run a transformer forward on a GGUF model.  Each entry point
 * below names the upstream symbol it stands in for and the reference call site that consumes it
 * (file:line under /root/reference).  Plain C types, opaque handles, int status codes, no
 * exceptions, caller-owned buffers.  All arithmetic behind it is hand-written HIP for gfx950;
 * there is no CPU fallback: every call fails with MI355_ERR_NO_DEVICE when no GPU is present.
 *
 * Thread-safety: like llama.h — a context must be driven from one thread at a time (the
 * reference drives it from the single DoBackgroundTasks thread, llama_server_context.cc:280).
 */
#ifndef MI355_LLAMA_H
#define MI355_LLAMA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_API __attribute__((visibility("default")))

typedef struct mi355_model   mi355_model;
typedef struct mi355_context mi355_context;

typedef int32_t mi355_token;
typedef int32_t mi355_pos;
typedef int32_t mi355_seq_id;

enum mi355_status {
    MI355_OK            = 0,
    MI355_ERR_NO_DEVICE = -100,  /* no MI355X / HIP runtime unavailable */
    MI355_ERR_IO        = -101,
    MI355_ERR_FORMAT    = -102,  /* not a GGUF v2/v3 file, or unsupported arch / tensor type */
    MI355_ERR_ARG       = -103,
    MI355_ERR_OOM       = -104,
    MI355_ERR_HIP       = -105,
};

/* ggml type ids as stored in GGUF (upstream ggml.h enum ggml_type; used at llama_engine.cc:272-281) */
enum mi355_type {
    MI355_TYPE_F32 = 0, MI355_TYPE_F16 = 1, MI355_TYPE_Q4_0 = 2, MI355_TYPE_Q8_0 = 8,
    MI355_TYPE_Q4_K = 12, MI355_TYPE_Q5_K = 13, MI355_TYPE_Q6_K = 14, MI355_TYPE_Q8_K = 15,
};

/* ------------------------------------------------------------------ backend */
/* llama_backend_init (llama_engine.cc:688).  Returns MI355_OK or MI355_ERR_NO_DEVICE. */
MI355_API int         mi355_backend_init(void);
MI355_API void        mi355_backend_free(void);
MI355_API int         mi355_device_count(void);
/* last error text of the calling thread ("" if none) */
MI355_API const char *mi355_last_error(void);
/* ggml_time_us (llama_client_slot.cc:57; llama_server_context.cc:417,1314,1378,1685) */
MI355_API int64_t     mi355_time_us(void);
/* llama_print_system_info (llama_engine.cc:702) */
MI355_API const char *mi355_print_system_info(void);

/* ------------------------------------------------------------------ model */
typedef struct mi355_model_params {
    int32_t n_gpu_layers;   /* `ngl` (llama_engine.cc:609-611); this backend is device-only: must be > 0 */
    int32_t main_gpu;       /* HIP device ordinal */
    int32_t use_mmap;       /* `use_mmap` (llama_engine.cc:649) */
    int32_t use_mlock;      /* `mlock` (llama_engine.cc:569-571); accepted, ignored */
    /* row split over RCCL (see mi355_tp_init below; new keys proposed in SURVEY.md §2b) */
    int32_t tp_rank;        /* 0 when tp_size <= 1 */
    int32_t tp_size;        /* 1 = whole model on this GPU */
    /* prompt-processing copy of the projection weights as pre-expanded int8 MFMA operand planes (2 B / weight):
     * -1 = keep it when device memory allows (default), 0 = never (expand on the fly), 1 = required */
    int32_t prefill_planes;
} mi355_model_params;

MI355_API mi355_model_params mi355_model_default_params(void);
/* common_init_from_params -> llama_model_load_from_file (llama_server_context.cc:207) */
MI355_API mi355_model *mi355_model_load_from_file(const char *path_gguf, mi355_model_params params);
MI355_API void         mi355_model_free(mi355_model *model);

MI355_API int32_t  mi355_model_n_vocab(const mi355_model *m);       /* llama_vocab_n_tokens (ctx.cc:512) */
MI355_API int32_t  mi355_model_n_embd(const mi355_model *m);        /* llama_n_embd (ctx.cc:218,1033,1098) */
MI355_API int32_t  mi355_model_n_layer(const mi355_model *m);
MI355_API int32_t  mi355_model_n_head(const mi355_model *m);
MI355_API int32_t  mi355_model_n_head_kv(const mi355_model *m);
MI355_API int32_t  mi355_model_n_ctx_train(const mi355_model *m);
MI355_API uint64_t mi355_model_size(const mi355_model *m);          /* llama_model_size (llama_engine.cc:482) */
/* patches/0001-Add-API-query-buffer-size.patch:14-15,46-64 (used at llama_engine.cc:475-476) */
MI355_API uint64_t mi355_model_cpu_buffer(const mi355_model *m);    /* llama_get_cpu_buffer  -> `ram`  */
MI355_API uint64_t mi355_model_other_buffer(const mi355_model *m);  /* llama_get_other_buffer -> `vram` */
/* algorithmic weight bytes one decoded token reads; for roofline reporting */
MI355_API uint64_t mi355_model_bytes_per_token(const mi355_model *m);
MI355_API uint64_t mi355_model_planes_bytes(const mi355_model *m);  /* device bytes of the prefill planes (0 = none) */
MI355_API const char *mi355_model_desc(const mi355_model *m);
/* GGUF metadata lookup: returns 1 and fills buf if `key` exists and is a string/scalar */
MI355_API int mi355_model_meta_str(const mi355_model *m, const char *key, char *buf, size_t buf_size);

/* ------------------------------------------------------------------ context */
typedef struct mi355_context_params {
    uint32_t n_ctx;        /* `ctx_len` total KV cells (llama_engine.cc:612) */
    uint32_t n_batch;      /* `n_batch` (llama_engine.cc:617) */
    uint32_t n_ubatch;     /* `n_ubatch` (llama_engine.cc:618) */
    uint32_t n_seq_max;    /* `n_parallel` (llama_engine.cc:620) */
    int32_t  type_k;       /* `cache_type` -> mi355_type {F16,Q8_0,Q4_0} (llama_engine.cc:628-637) */
    int32_t  type_v;
    int32_t  flash_attn;   /* `flash_attn`, forced on for quantised caches (llama_engine.cc:639-647) */
    int32_t  embeddings;   /* `embedding` (llama_engine.cc:613-616) */
    int32_t  use_graphs;   /* capture the single-token decode step in a hipGraph (default 1) */
    int32_t  logits_to_host; /* 1 (default): flagged logits rows are copied to host by mi355_decode itself, as llama_decode
                              * does; 0: rows stay on the device until mi355_get_logits_ith asks (device-side greedy) */
} mi355_context_params;

MI355_API mi355_context_params mi355_context_default_params(void);
/* common_init_from_params -> llama_init_from_model (llama_server_context.cc:207-209) */
MI355_API mi355_context *mi355_context_new(mi355_model *model, mi355_context_params params);
MI355_API void           mi355_context_free(mi355_context *ctx);

MI355_API uint32_t mi355_n_ctx(const mi355_context *ctx);      /* llama_n_ctx   (ctx.cc:236) */
MI355_API uint32_t mi355_n_batch(const mi355_context *ctx);    /* llama_n_batch (ctx.cc:1351) */
MI355_API uint32_t mi355_n_ubatch(const mi355_context *ctx);   /* llama_n_ubatch(ctx.cc:1352) */
MI355_API uint64_t mi355_context_device_bytes(const mi355_context *ctx); /* KV + activations on device */

/* llama_batch exactly as the reference fills it (llama_server_context.cc:265,1630-1635) */
typedef struct mi355_batch {
    int32_t        n_tokens;
    mi355_token   *token;     /* [n_tokens] */
    float         *embd;      /* NULL, or [n_tokens][n_embd] rows that take the place of the token embeddings (then token is NULL): how the
                               * reference feeds image embeddings to the model (llava_embd_batch, llama_server_context.cc:1093-1107) */
    mi355_pos     *pos;       /* [n_tokens] */
    int32_t       *n_seq_id;  /* [n_tokens] */
    mi355_seq_id **seq_id;    /* [n_tokens][n_seq_id] */
    int8_t        *logits;    /* [n_tokens] != 0 => produce logits for that row */
} mi355_batch;

MI355_API mi355_batch mi355_batch_init(int32_t n_tokens, int32_t embd, int32_t n_seq_max); /* llama_batch_init (ctx.cc:265) */
MI355_API void        mi355_batch_free(mi355_batch batch);

/* llama_decode (llama_server_context.cc:654,1085,1103,1635).
 * Returns 0 on success, 1 if no KV slot could be found for the batch (caller halves n_batch and
 * retries, ctx.cc:1636-1663), < 0 on a fatal error.  Logits rows are host-visible afterwards. */
MI355_API int32_t mi355_decode(mi355_context *ctx, mi355_batch batch);
/* n greedy steps of one sequence in one call - per step: llama_decode of the previous step's token at pos0 + i, its logits row made host-visible
 * (llama_get_logits_ith), the arg-max as the next token (the greedy end of the sampler chain): the inner loop of the reference's UpdateSlots for a temperature-0
 * request (llama_server_context.cc:1628-1707), on the C side.  out_tokens (nullable) [n].  Returns the steps done (n, or fewer with the reason in mi355_last_error). */
MI355_API int32_t mi355_greedy_steps(mi355_context *ctx, mi355_token first, mi355_pos pos0, mi355_seq_id seq, int32_t n, mi355_token *out_tokens);
/* llama_get_logits_ith as used through common_sampler_sample(ctx, idx) (ctx.cc:1679-1680).
 * i indexes the batch of the last mi355_decode; NULL if that row had logits[i] == 0, in embeddings mode, or when the step's results were discarded
 * (an in-kernel wait that gave up, a failed copy) - mi355_last_error then says which. */
MI355_API float  *mi355_get_logits_ith(mi355_context *ctx, int32_t i);
/* device-side greedy front end: argmax token of row i without copying the row */
MI355_API int32_t mi355_get_argmax_ith(mi355_context *ctx, int32_t i);
/* device-side sampling front end beyond greedy (stands in for the head of common_sampler_sample's chain, reference call site
   src/llama_server_context.cc:1679-1698): the k <= 128 best candidates of row i after logit_bias and the repetition / frequency / presence penalties,
   best first (higher logit; lower token id on ties) - k (token, logit) pairs cross to the host instead of the whole row.  adj_tok / adj_bias / adj_count:
   n_adj <= 192 adjusted tokens (bias 0 = none, count = occurrences in the penalty window, 0 = not penalised).  Returns k or < 0. */
MI355_API int32_t mi355_get_topk_ith(mi355_context *ctx, int32_t i, int32_t k, int32_t n_adj, const int32_t *adj_tok, const float *adj_bias,
                                     const int32_t *adj_count, float penalty_repeat, float penalty_freq, float penalty_present,
                                     int32_t *toks_out, float *logits_out);
/* diagnosis: single-token steps this context ran as ONE launch (decode_mega.hip; see mi355_debug_set_option "decode_mega") */
MI355_API int64_t mi355_debug_mega_steps(const mi355_context *ctx);
/* diagnosis: single-token steps this context ran through the layer engine (decode_engine.hip: one persistent launch per layer for the mat-vecs between two
   attention calls; opt-in: mi355_debug_set_option "decode_engine" 1 or MI355_ENGINE=1) */
MI355_API int64_t mi355_debug_engine_steps(const mi355_context *ctx);
/* single-token steps of this context that took the wait-free launches because another context of the device had the one-launch attention + attn_output kernel
   (workgroups that wait for each other) in flight: two models of one server decoding at the same time never run that kernel beside each other */
MI355_API int64_t mi355_debug_fused_skipped_steps(const mi355_context *ctx);
/* diagnosis: launches of this context that ran a layer's Q | K | V mat-vecs INSIDE its attention + attn_output launch (csrc/attn_out.hip QF, round 6: one launch
   per layer for the whole attention block of a single-token step; "qkv_attn_fused" 0 / MI355_QKV_ATTN_FUSED=0 keeps Q | K | V a launch of its own).  Counts
   launches issued eagerly or while a graph was captured, not graph replays. */
MI355_API int64_t mi355_debug_qkv_attn_launches(const mi355_context *ctx);
/* host logic only (no GPU needed): would a single-token step of a layer with these tensor types (MI355_TYPE_*) and this geometry run its Q | K | V inside the attention
   launch at a context of n_kv cells?  Returns the launch's LDS bytes per workgroup (<= 163840) and, in *slots_out, the 4 KiB DMA slots of a workgroup's rows; 0 = the two
   launches (types without a form, n_embd != n_head * head_dim or > 4096, head_dim != 128, rows that do not fit beside W_o: Llama-2-7B) */
MI355_API int64_t mi355_debug_qkv_attn_plan(int32_t type_q, int32_t type_k, int32_t type_v, int32_t type_o, int32_t n_embd, int32_t n_head, int32_t n_head_kv, int32_t head_dim,
                                            int32_t type_kv, int32_t n_kv, int32_t *slots_out);
/* llama_set_embeddings (ctx.cc:299) */
MI355_API void    mi355_set_embeddings(mi355_context *ctx, int32_t enabled);
/* llama_get_embeddings_ith (ctx.cc:1042-1044): final-norm hidden state (n_embd floats, host memory) of batch row i of the
 * last mi355_decode issued while embeddings were enabled; NULL otherwise.  Pooling is NONE on this architecture, so
 * llama_get_embeddings_seq has no counterpart (the reference falls back to _ith, ctx.cc:1043-1045). */
MI355_API float  *mi355_get_embeddings_ith(mi355_context *ctx, int32_t i);
MI355_API void    mi355_synchronize(mi355_context *ctx);

/* KV cache bookkeeping (ctx.cc:287; 661,1547; 1288,1540,1542; 1290) */
MI355_API void    mi355_kv_cache_clear(mi355_context *ctx);
MI355_API int32_t mi355_kv_cache_seq_rm(mi355_context *ctx, mi355_seq_id seq, mi355_pos p0, mi355_pos p1); /* 1 = ok */
MI355_API void    mi355_kv_cache_seq_cp(mi355_context *ctx, mi355_seq_id src, mi355_seq_id dst, mi355_pos p0, mi355_pos p1);
MI355_API void    mi355_kv_cache_seq_add(mi355_context *ctx, mi355_seq_id seq, mi355_pos p0, mi355_pos p1, mi355_pos delta);
MI355_API int32_t mi355_kv_cache_used_cells(const mi355_context *ctx);

/* debugging taps for parity tests (enable before decode): residual stream after layer `il` for the last
 * decoded micro-batch, copied to dst[n_tokens * n_embd]; returns n_tokens or < 0 */
MI355_API void    mi355_debug_enable_taps(mi355_context *ctx, int32_t enabled);
MI355_API int32_t mi355_debug_layer_out(mi355_context *ctx, int32_t il, float *dst, size_t dst_floats);

/* ------------------------------------------------------------------ per-op entry points
 * Host-buffer wrappers around the individual HIP kernels, for parity tests and rocprof.
 * Each copies inputs to the device, runs the same kernel the decode graph uses, copies back. */

/* quantize_row_q8_K / quantize_row_q8_0 (activation side).  out receives ggml-layout blocks. */
MI355_API int mi355_op_quantize_act(int32_t act_type, const float *x, int64_t n_per_row, int64_t n_rows, void *out_blocks);
/* y[T][N] = W[N][K] . x[T][K]; W is ggml-layout blocks of `type`.  isum/msum (nullable):
 * per (token, row, block) integer partial sums for bit-exact checks. */
MI355_API int mi355_op_mul_mat(int32_t type, const void *W, int64_t N, int64_t K, const float *x, int64_t T,
                               float *y, int32_t *isum, int32_t *msum);
/* ffn_gate and ffn_up (one K-quant type, N rows each, N % 32 == 0) against the same T activation rows with SwiGLU in the
 * epilogue, as the prompt path launches them (mmq_planes2_swiglu_kernel): y[t][n] = silu(Wg[n] . x[t]) * (Wu[n] . x[t]).
 * Shapes too small for that launch are refused unless the debug option "mmq_tiles" = 4 forces the kernel. */
MI355_API int mi355_op_ffn_gate_up(int32_t type, const void *Wg, const void *Wu, int64_t N, int64_t K, const float *x, int64_t T, float *y);
MI355_API int mi355_op_rms_norm_mul(const float *x, const float *w, int64_t n, int64_t T, float eps, float *y);
MI355_API int mi355_op_rope(float *x, int32_t n_head, int32_t head_dim, int32_t n_rot, const int32_t *pos, int64_t T,
                            float freq_base, float freq_scale, const float *freq_factors, int32_t neox);
/* the same rotation with YaRN scaling ({arch}.rope.scaling.type "yarn": llama.cpp's rope_yarn; ext_factor 0 and attn_factor 1 reduce it to mi355_op_rope) */
MI355_API int mi355_op_rope_yarn(float *x, int32_t n_head, int32_t head_dim, int32_t n_rot, const int32_t *pos, int64_t T,
                                 float freq_base, float freq_scale, const float *freq_factors, int32_t neox,
                                 float ext_factor, float attn_factor, float corr_lo, float corr_hi);
MI355_API int mi355_op_get_rows(int32_t type, const void *table, int64_t row_elems, int64_t n_rows_table,
                                const int32_t *ids, int64_t n_ids, float *dst);
MI355_API int mi355_op_swiglu(const float *gate, const float *up, int64_t n, float *y);
MI355_API int mi355_op_soft_max(const float *x, const float *mask, int64_t n, int64_t rows, float scale, float *y);
/* build_moe_ffn's expert selection: softmax over n_expert router logits per token, the k largest (first index wins ties), weights
   renormalised; ids [T][k], w [T][k] */
MI355_API int mi355_op_moe_route(const float *logits, int64_t T, int32_t n_expert, int32_t k, int32_t *ids, float *w);
/* flash_attn_ext for T query tokens: K/V given as ggml-layout rows [n_cells][n_head_kv*head_dim] of type_k/type_v;
 * visibility: cell c is visible to token t iff cell_pos[c] >= 0 && cell_pos[c] <= q_pos[t]. */
MI355_API int mi355_op_flash_attn(const float *q, int64_t T, int32_t n_head, int32_t n_head_kv, int32_t head_dim,
                                  int32_t type_k, const void *k, int32_t type_v, const void *v, int32_t n_cells,
                                  const int32_t *cell_pos, const int32_t *q_pos, float scale, float *out);

/* The attention block of ONE single-token step as the decode path launches it (rope of q and of the token's K row, the K / V row quantised into the cache,
 * flash_attn_ext over the visible cells, Q8_K quantisation, the attn_output mat-vec + residual: the ops between two llama_decode graph nodes of
 * llm_build_llama that attn_out.hip runs as one launch; reached through llama_decode, src/llama_server_context.cc:1635).  fused = 1: the one-launch form,
 * 0: the single-launch decode attention followed by the weight-stream mat-vec.  k / v: the cache before the step as ggml-layout rows [n_cells][n_head_kv *
 * head_dim]; cell_pos[tok_cell] must already hold tok_pos.  att_out [n_head * head_dim], out [n_embd], k_row_out / v_row_out: the cache row the step wrote,
 * in ggml block layout (any of the four may be NULL). */
MI355_API int mi355_op_attn_step(const float *q, const float *k_new, const float *v_new, int32_t n_head, int32_t n_head_kv, int32_t head_dim, int32_t type_k,
                                 const void *k, int32_t type_v, const void *v, int32_t n_cells, const int32_t *cell_pos, int32_t tok_pos, int32_t tok_cell,
                                 float rope_base, int32_t n_rot, float scale, int32_t type_o, const void *W_o, int64_t n_embd, const float *resid, int32_t fused,
                                 float *att_out, float *out, void *k_row_out, void *v_row_out);

/* ------------------------------------------------------------------ tokenizer
 * llama_tokenize / llama_token_to_piece as reached through common_tokenize / common_token_to_piece
 * (src/llama_server_context.cc:395-410, 536, 644, 720, 936, 992) and llama_vocab_bos/eos/is_eog (:512-517, 792).
 * mi355_tokenize returns the token count, or -(count needed) when `cap` is too small. */
MI355_API int32_t mi355_tokenize(mi355_model *m, const char *text, int32_t text_len, mi355_token *out, int32_t cap,
                                 int32_t add_special, int32_t parse_special);
MI355_API int32_t mi355_token_to_piece(mi355_model *m, mi355_token tok, char *buf, int32_t cap, int32_t special);
MI355_API mi355_token mi355_token_bos(mi355_model *m);
MI355_API mi355_token mi355_token_eos(mi355_model *m);
MI355_API int32_t mi355_token_is_eog(mi355_model *m, mi355_token tok);

/* ------------------------------------------------------------------ engine (the reference's plugin surface)
 * class EngineI (base/cortex-common/enginei.h:13-74) as implemented by LlamaEngine (src/llama_engine.cc): the slot
 * loop, sampler and request/response shaping run in the host library; bodies cross the boundary as UTF-8 JSON text
 * instead of std::shared_ptr<Json::Value>.  The callback receives (status, body) exactly as the reference's
 * std::function<void(Json::Value&&, Json::Value&&)> does: status = {is_done, has_error, is_stream, status_code};
 * streaming completions call it once per chunk (body = {"data": "data: {...}\n\n"}) from a worker thread. */
typedef struct mi355_engine mi355_engine;
typedef void (*mi355_engine_callback)(const char *status_json, const char *body_json, void *user);
MI355_API mi355_engine *mi355_engine_create(void);                 /* get_engine()  (src/llama_engine.cc:1300-1304) */
MI355_API void mi355_engine_destroy(mi355_engine *e);
/* Lifetime of `user`: the reference's callback is a std::function that dies with the request (it is captured by value in the queued task,
   src/llama_engine.cc:946-948); a C callback has no destructor, so a host that allocates per-request state registers one function here and the library calls
   it exactly once per request with that request's `user`, after the LAST callback the request will ever make - also when the request ends without a terminal
   (is_done / has_error) callback, as a stream stopped by StopInferencing does (:950-955).  Set it before the first request; NULL (default) = nothing is called. */
MI355_API void mi355_engine_set_release_callback(mi355_engine *e, void (*release)(void *user));
MI355_API void mi355_engine_load_model(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API void mi355_engine_unload_model(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API void mi355_engine_get_model_status(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API void mi355_engine_get_models(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API void mi355_engine_handle_chat_completion(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API void mi355_engine_handle_embedding(mi355_engine *e, const char *body_json, mi355_engine_callback cb, void *user);
MI355_API int32_t mi355_engine_is_supported(mi355_engine *e, const char *feature);
MI355_API void mi355_engine_stop_inferencing(mi355_engine *e, const char *model_id);
/* EngineI::Load(EngineLoadOption) / Unload(EngineUnloadOption) (base/cortex-common/enginei.h:14-35; LlamaEngine: src/llama_engine.cc:289-303): Load =
   SetFileLogger(max_log_lines, log_path) + SetLogLevel(log_level).  Paths as UTF-8 strings (std::filesystem::path::string()), log_level = the value of
   trantor::Logger::LogLevel (kTrace 0, kDebug 1, kInfo 2, kWarn 3, kError 4, kFatal 5). */
MI355_API void mi355_engine_load(mi355_engine *e, const char *engine_path, const char *deps_path, int32_t is_custom_engine_path, const char *log_path,
                                 int32_t max_log_lines, int32_t log_level);
MI355_API void mi355_engine_unload(mi355_engine *e);
/* EngineI::SetFileLogger / SetLogLevel (enginei.h:69-71; src/llama_engine.cc:502-548): log lines go to log_path, kept to its last max_log_lines lines
   ("" = back to stderr); messages below log_level are dropped */
MI355_API void mi355_engine_set_file_logger(mi355_engine *e, int32_t max_log_lines, const char *log_path);
MI355_API void mi355_engine_set_log_level(mi355_engine *e, int32_t log_level);
/* the bridge the other way: every log line also goes to cb (level = trantor's numbers), so an adapter can re-emit it through the host's own LOG_* macros
   (the reference routes llama.cpp's log through trantor the same way: llama_log_set, src/llama_engine.cc:313-330).  Process-wide, like trantor::Logger. */
typedef void (*mi355_log_callback)(int level, const char *line, void *user);
MI355_API void mi355_engine_set_log_callback(mi355_engine *e, mi355_log_callback cb, void *user);

/* Test / tool switches of the per-op entry points: "mmq_planes" (1: mi355_op_mul_mat with T >= 32 expands the weight
 * into MFMA planes first, as a loaded model does; 0: expands on the fly inside the kernel), "mmq_tiles" (0 | 1 | 2
 * token tiles per wave), "mmq_ksplit" (1: 8 <= T <= 64 uses the K-split small-batch kernel, as the runtime does; 0: the
 * kernels the other T ranges use); and of contexts created afterwards: "decode_mega" (1: single-token steps of a dense
 * K-quant model with Llama-3-8B's layer geometry run every layer in one launch; 0, the default: one launch per operation
 * — both produce the same bits; the single launch measured slower); "moe_group_min" (batches of at least
 * this many tokens run a mixture-of-experts feed-forward grouped by expert, default 8; smaller ones loop over (token,
 * expert) with the mat-vec); "tp_null_group" (measurement aid: the process becomes rank 0 of a row-split group of `value`
 * ranks whose other members do not exist — every exchange is a device copy of this rank's own part, so a model loaded with
 * tp_rank 0 / tp_size value times ONE rank's compute without communication; its outputs are not the model's).
 * Returns MI355_OK or MI355_ERR_ARG for an unknown name. */
MI355_API int mi355_debug_set_option(const char *name, int32_t value);

/* ------------------------------------------------------------------ row split across GPUs (one process per GPU)
 * Not in the reference: cortex.llamacpp never sets llama.cpp's split_mode / tensor_split (llama_engine.cc:553-658
 * writes no such field), so every visible GPU gets whole layers.  Here a model loaded with tp_size > 1 holds rank tp_rank's
 * rows of attn_q/k/v, ffn_gate/up and output and the matching columns of attn_output / ffn_down; the partial sums of those
 * two are summed across the ranks (RCCL all-reduce over xGMI on the context's stream, inside its graphs), the logits slices
 * are gathered.  Every rank must issue the same mi355_decode calls.  Order: mi355_tp_unique_id on rank 0 -> hand the 128
 * bytes to the other ranks (any side channel) -> mi355_tp_init on every rank (after selecting its device with
 * main_gpu / HIP_VISIBLE_DEVICES) -> mi355_model_load_from_file with the same tp_rank / tp_size. */
#define MI355_TP_ID_BYTES 128
MI355_API int mi355_tp_unique_id(void *id_out, size_t cap);                         /* returns bytes written or < 0 */
MI355_API int mi355_tp_init(int32_t device, int32_t rank, int32_t size, const void *id, size_t id_len);
MI355_API void mi355_tp_shutdown(void);
/* ONE /loadmodel for the row split (what a user of the reference sees: its engine drives every visible device from one call, src/llama_engine.cc:609-611).
 * A load body with "split_mode": "row" (+ "tensor_split": even shares only, "main_gpu", "split_ranks": see host/tp_split.cc) given to
 * mi355_engine_load_model makes the calling process rank 0 of a group it forms itself: it starts bin/mi355_tp_worker once per further rank, hands it the RCCL id
 * (or, where ranks share a device, a shared-memory exchange segment) over a socket pair, and steps the workers in lock-step with every batch and KV operation.
 * None of the calls above is needed then.  mi355_tp_worker_main is that worker program's body: `sock_fd` is the inherited socket; returns the exit code. */
MI355_API int mi355_tp_worker_main(int sock_fd);
MI355_API int32_t mi355_tp_rank(void);
MI355_API int32_t mi355_tp_size(void);
/* Validation transport for boxes where the ranks share one GPU (RCCL refuses that): the exchange goes through this
 * host callback instead (op 0: sum `n` floats in place over the ranks; op 1: all-gather, `n` floats per rank, the caller's
 * part already at buf + rank * n).  Graphs are off while it is set.  fn == NULL removes it. */
/* One-shot peer-to-peer all-reduce for the decode-sized exchanges (host/tp_comm.h): after mi355_tp_init (or _set_host_exchange) every
 * rank calls mi355_tp_p2p_local_handle (64 bytes out; max_floats = the largest message it should take, e.g. n_embd * 8), the handles are gathered
 * rank-major over the side channel that carried the RCCL id, and every rank calls mi355_tp_p2p_enable with all of them.  All-reduces of up to max_floats
 * floats then run as one peer-to-peer kernel (IPC-mapped buffers, xGMI peer stores, rank-order sum); larger ones keep the base transport. */
#define MI355_TP_P2P_HANDLE_BYTES 64
MI355_API int mi355_tp_p2p_local_handle(void *out, size_t cap, size_t max_floats);   /* returns bytes written or < 0 */
MI355_API int mi355_tp_p2p_enable(const void *handles, size_t len);
MI355_API int64_t mi355_tp_p2p_exchanges(void);                                      /* diagnosis: all-reduces that took the peer-to-peer kernel */
/* The same bootstrap with a second size: all-reduces of (max_floats, prompt_floats] floats - the n_embd x n_ubatch partial sums of a prompt batch - run
 * as ONE reduce-scatter + all-gather kernel over all xGMI links at once (host/tp_comm.cc p2p_rsag_kernel: the message is cut into one segment per rank,
 * every rank stores its part of segment q into rank q's buffer, the owner adds the parts in rank order and stores the sum into everybody's buffer).  A
 * ring all-reduce keeps one link direction per rank busy; the reference has no counterpart (it copies activations between peers,
 * SURVEY.md §8e).  Option "tp_p2p_prompt" = 0 (mi355_debug_set_option) routes these messages back to RCCL. */
MI355_API int mi355_tp_p2p_local_handle2(void *out, size_t cap, size_t max_floats, size_t prompt_floats);
MI355_API int64_t mi355_tp_p2p_prompt_exchanges(void);                               /* diagnosis: all-reduces that took the reduce-scatter + all-gather kernel */
typedef int (*mi355_tp_host_exchange)(void *user, float *buf, size_t n, int32_t op);
MI355_API int mi355_tp_set_host_exchange(mi355_tp_host_exchange fn, void *user, int32_t rank, int32_t size);

/* ------------------------------------------------------------------ measurement hooks (bench.py) */
/* Streams `bytes` through a read-only reduction kernel `iters` times; returns achieved GB/s (HIP events). */
MI355_API double mi355_bench_hbm_read(size_t bytes, int iters);
/* Per-kernel-class device time of the LAST decode call in microseconds, measured with HIP events on the
 * context's stream (eager mode only).  names/us arrays of capacity cap; returns count. */
MI355_API int32_t mi355_profile_last_decode(mi355_context *ctx, const char **names, float *us, int32_t cap);
MI355_API void    mi355_profile_enable(mi355_context *ctx, int32_t enabled);
/* Runs the dominant decode kernel (quantised mat-vec over every weight tensor of the model, one token)
 * `iters` times on the context's stream, timed with HIP events; returns mean microseconds per sweep and
 * writes the algorithmic bytes of one sweep. */
/* test hook for mixture-of-experts files (build_moe_ffn's top-k): the NEXT mi355_decode call (one micro-batch of n_tokens tokens) takes the
 * experts ids[layer][token][rank] instead of its own router's selection (the weights stay this side's router probabilities of those experts, renormalised).
 * Parity tests hand over the CPU restatement's selection so that a rounding flip on a near tie of the router cannot send a token to another expert on one
 * side only.  Arms one call. */
MI355_API int     mi355_debug_force_moe_ids(mi355_context *ctx, const int32_t *ids, int32_t n_layer, int32_t n_tokens, int32_t k);
MI355_API double  mi355_bench_weight_sweep(mi355_context *ctx, int iters, uint64_t *bytes_per_sweep);
/* ... and the number of mat-vec launches the sweep holds (the step's own launches of the weight-stream kernel: where attn_output runs inside the attention
 * launch - attn_out.hip - it is not among them, and its bytes are not counted) */
MI355_API double  mi355_bench_weight_sweep2(mi355_context *ctx, int iters, uint64_t *bytes_per_sweep, int32_t *launches_per_sweep);

/* ------------------------------------------------------------------ LLaVA image path (projector file "mmproj"; llama.cpp examples/llava behind the reference)
 * The reference: clip_model_load (llama_server_context.cc:187), clip_n_mmproj_embd (:217), clip_image_load_from_bytes (:568),
 * llava_image_embed_make_with_clip_img (:820, = clip_image_preprocess + clip_image_encode); the rows then enter the model as llama_batch.embd (:1093-1107).
 * LLaVA-1.5 and LLaVA-1.6 style files (CLIP ViT tower, MLP projector, f16 weights; LLaVA-1.6 = clip.vision.image_grid_pinpoints with
 * clip.vision.mm_patch_merge_type "spatial_unpad": a picture becomes an overview plus the tiles of the best-fitting canvas). */
typedef struct mi355_clip mi355_clip;
MI355_API mi355_clip *mi355_clip_model_load(const char *path, int32_t main_gpu);       /* clip_model_load; NULL + mi355_last_error on failure */
MI355_API void        mi355_clip_free(mi355_clip *clip);                                /* clip_free */
MI355_API int32_t     mi355_clip_n_mmproj_embd(const mi355_clip *clip);                 /* clip_n_mmproj_embd: must equal the model's n_embd */
MI355_API int32_t     mi355_clip_n_patches(const mi355_clip *clip);                     /* clip_n_patches: embedding rows per image */
MI355_API int32_t     mi355_clip_image_size(const mi355_clip *clip);
/* the most rows mi355_llava_image_embed_from_bytes can write for one picture: n_patches (LLaVA-1.5) or n_patches * (1 + tiles of the largest canvas) */
MI355_API int32_t     mi355_clip_max_image_rows(const mi355_clip *clip);
/* clip_image_load_from_bytes: PNG / JPEG (Huffman-coded sequential or progressive) / BMP / binary PNM bytes -> 8-bit RGB [ny][nx][3].  rgb_out may be NULL to query the size.
 * Returns 0, or < 0 with the reason in mi355_last_error (unknown format, truncated data, rgb_cap too small). */
MI355_API int32_t     mi355_clip_image_load_from_bytes(const uint8_t *bytes, size_t n_bytes, int32_t *nx, int32_t *ny, uint8_t *rgb_out, size_t rgb_cap);
/* clip_image_preprocess (LLaVA-1.5: pad to a square with the mean colour, bilinear resample, normalise): rgb [ny][nx][3] -> out [3][S][S], S = image_size */
MI355_API int32_t     mi355_clip_image_preprocess(const mi355_clip *clip, const uint8_t *rgb, int32_t nx, int32_t ny, float *out);
/* clip_image_preprocess, every image the encoder sees for one picture: one (as above) without an image grid; with one, the overview (bicubic resize of the whole
 * picture to S x S) then the S x S tiles, row-major, of the picture fitted (aspect kept, bicubic, centred on black) to the best canvas of the grid.
 * out [n][3][S][S]; returns n (<= 1 + max tiles) or < 0; grid_w x grid_h = tiles across / down (0 x 0 without a grid). */
MI355_API int32_t     mi355_clip_image_preprocess_grid(const mi355_clip *clip, const uint8_t *rgb, int32_t nx, int32_t ny, float *out, size_t out_floats,
                                                       int32_t *grid_w, int32_t *grid_h);
/* clip_image_encode: img [3][S][S] -> out [n_patches][n_mmproj_embd] (host memory) */
MI355_API int32_t     mi355_clip_image_encode(mi355_clip *clip, const float *img, float *out);
/* llava_image_embed_make_with_clip_img on encoded image bytes: decode + preprocess + encode (+ with an image grid, the tiles' rows re-ordered to the canvas'
 * row-major order behind the overview's: clip_llava_handle_patches).  Returns the number of rows written (<= mi355_clip_max_image_rows) or < 0. */
MI355_API int32_t     mi355_llava_image_embed_from_bytes(mi355_clip *clip, const uint8_t *bytes, size_t n_bytes, float *out, size_t out_floats);

#ifdef __cplusplus
}
#endif
#endif /* MI355_LLAMA_H */
