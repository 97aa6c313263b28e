// runtime.h — model residency and the decode executor (host C++ above the HIP kernels).
//
// Plays the role of the llama.cpp objects the reference holds through common_init_result
// (src/llama_server_context.cc:207-209): llama_model (weights on device), llama_context (KV cache,
// batch execution, logits), llama_kv_cache (cells with positions and sequence sets).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../csrc/kernels.h"
#include "gguf.h"

namespace mi355 {

struct DevTensor {
    std::string name;
    int type = 0;
    int64_t K = 0;          // ne[0]: contraction / row length
    int64_t N = 0;          // rows per expert (ne[1])
    int64_t n_expert = 1;   // ne[2] for *_exps tensors
    uint8_t *data = nullptr;
    size_t row_bytes = 0;   // device row stride
    size_t bytes = 0;       // device bytes
    size_t ggml_bytes = 0;  // on-disk bytes
    uint8_t *planes = nullptr;   // pre-expanded MFMA operand planes for prompt processing (mmq.hip), optional
    size_t planes_bytes = 0;
    bool valid() const { return data != nullptr; }
};

struct LayerWeights {
    DevTensor attn_norm, wq, wk, wv, wo, bq, bk, bv;
    DevTensor ffn_norm, gate, up, down;
    // single-token steps of a feed-forward width the weight stream has no form for (K = 28672: a 16 KB row does not fit the ring pairwise) while half of
    // it has one: the column halves of ffn_down as two tensors of their own, contracted by two launches (x += W_lo a_lo; x += W_hi a_hi).  A second copy
    // of the tensor in HBM; prompt batches keep the whole tensor (and its planes).  Empty otherwise.
    DevTensor down_lo, down_hi;
    DevTensor gate_inp, gate_exps, up_exps, down_exps;
    // encoder files (nomic-bert): the fused Q | K | V projection (wq / wk / wv are row ranges of it), LayerNorms with biases after the attention and the feed-forward block
    DevTensor wqkv, bo, attn_out_norm, attn_out_norm_b, layer_out_norm, layer_out_norm_b;
};

struct HParams {
    std::string arch;
    int n_embd = 0, n_layer = 0, n_ff = 0, n_head = 0, n_head_kv = 0, n_rot = 0, n_vocab = 0;
    int n_expert = 0, n_expert_used = 0, head_dim = 0, n_ctx_train = 0;
    int pooling_type = 0;          // {arch}.pooling_type: 0 none, 1 mean, 2 cls, 3 last (what llama_get_embeddings_seq pools over a sequence's tokens)
    float eps = 1e-5f, rope_base = 10000.0f, rope_scale = 1.0f;
    int rope_neox = 0;
    bool encoder = false;          // bidirectional attention, embeddings only (general.architecture nomic-bert: llm_build_bert)
    float yarn_ext = 0.0f, yarn_attn = 1.0f, yarn_lo = 0.0f, yarn_hi = 0.0f;   // rope.scaling.type "yarn" (RopeArgs, kernels.h)
    // row split: n_head, n_head_kv and n_ff above are THIS RANK's share; the file's values are kept here.
    // A shard is the same graph with fewer heads and a narrower feed-forward, attn_output and ffn_down contracting over
    // the local slice only: their partial sums are the one thing exchanged (tp_comm.h).
    int tp_rank = 0, tp_size = 1;
    bool tp_exchange = false;    // the process has a matching group: partial sums and logits slices go through it
    int n_head_full = 0, n_head_kv_full = 0, n_ff_full = 0;
    int n_vocab_local = 0;       // rows of the output projection held here (n_vocab when it is not split)
};

struct Model {
    HParams hp;
    std::unique_ptr<GGUFFile> file;
    std::string path, desc;
    int device = 0;
    std::vector<uint8_t *> arenas;      // hipMalloc'd blocks
    DevTensor tok_embd, out_norm, output, rope_freqs;
    DevTensor tok_types, tok_norm, tok_norm_b;       // encoder files: token-type table (row 0 is added to every token), LayerNorm of the embeddings
    std::vector<LayerWeights> layers;
    uint64_t device_bytes = 0, host_bytes = 0, file_tensor_bytes = 0, bytes_per_token = 0, planes_bytes = 0;
    ~Model();
};

// prefill_planes: 0 = never, 1 = always (fails when memory is short), -1 = when device memory allows (default)
// tp_size > 1: load rank tp_rank's slice of every projection (rows of attn_q/k/v, ffn_gate/up and output; the matching
// super-block columns of attn_output and ffn_down), cut on head and 256-element boundaries
Model *model_load(const std::string &path, int main_gpu, std::string &err, int &status, int prefill_planes = -1, int tp_rank = 0, int tp_size = 1);

struct KVCell {
    int32_t pos = -1;
    int32_t delta = 0;
    uint64_t seqs = 0;
};

void debug_raise_stream_error(unsigned code);        // tests: raise the sticky error word of the cross-workgroup kernels as a timed-out wait would
struct ContextParams {
    uint32_t n_ctx = 512, n_batch = 2048, n_ubatch = 512, n_seq_max = 1;
    int type_k = T_F16, type_v = T_F16;
    bool flash_attn = true, embeddings = false, use_graphs = true;
    bool logits_to_host = true;   // llama_decode contract: flagged logits rows are copied to host memory by the decode itself
};

struct ProfileEntry { std::string name; float us; };

struct Fuse {   // fused activation prologue of the single-token mat-vec (mmvq.hip stage_act)
    int mode = 0;
    const float *x = nullptr, *w = nullptr;
    float eps = 0.0f;
    float *out_host = nullptr;   // the launch also stores its results here (pinned host memory; the output head of a single-token step) - weight stream only
};

void set_moe_group_min(int tokens);   // tests: batch size from which a mixture-of-experts feed-forward is grouped by expert
void set_attn_store_fuse(bool on);   // tests: 0 = batched steps store K / V in their own launch before the attention
void set_rope_fast(bool on);         // tests: 0 = prompt batches rotate q / store K, V with the one-workgroup-per-token kernel
void set_decode_engine(int on);  // 1 / 0: the layer engine (decode_engine.hip) for single-token steps of contexts created afterwards; -1: environment MI355_ENGINE / default (off)
void set_decode_mega(bool on);   // tests: compare the whole-step kernel with the per-launch path (read when a context first decodes one token)

class Context {
  public:
    int64_t engine_steps = 0;          // single-token steps issued through the layer engine (graph replays included)
    int64_t qkv_attn_launches = 0;     // launches that ran a layer's Q | K | V inside its attention + attn_output launch (attn_out.hip QF)
    int64_t fused_skipped_steps = 0;   // single-token steps that took the wait-free launches because another context of the device held the cross-workgroup-wait kernel
    int64_t mega_steps = 0;            // single-token steps issued as one whole-step launch (graph replays included)
    Context(Model *m, const ContextParams &p);
    ~Context();
    bool init(std::string &err);

    // llama_decode semantics: 0 ok, 1 no KV slot, <0 error
    // embd != nullptr: rows of n_embd floats in place of token ids (llama_batch.embd; tokens must then be nullptr)
    int decode(int n_tokens, const int32_t *tokens, const int32_t *pos, const int32_t *n_seq_id, int32_t *const *seq_id,
               const int8_t *logits_flags, const float *embd = nullptr);
    float *logits_ith(int i);
    // llama_get_embeddings_ith: the final-norm hidden state of batch row i (n_embd floats, host memory); rows are
    // produced for the flagged tokens of the last decode while embeddings_enabled is set (then no logits are computed)
    float *embeddings_ith(int i);
    int32_t argmax_ith(int i);
    // test hook (mixture-of-experts files): the NEXT decode call (one micro-batch of T tokens) takes these experts, ids [n_layer][T][n_expert_used], instead of
    // its router's selection; the weights stay this side's probabilities of them.  One call, then routing is free again.  No graphs while it is armed.
    int force_moe_ids(const int32_t *ids, int n_layer, int T, int k);
    // device-side sampling front end: the k best (token, logit) candidates of batch row i after the adjustments (kernels.h launch_topk_rows), best
    // first; returns the count written (k) or < 0
    int topk_ith(int i, int k, const TopkAdj &adj, int32_t *toks, float *logits);
    int topk_rows(int n, const int *is, const int *ks, const TopkAdj *adjs, int32_t *toks, float *logits);   // n rows at once; outputs [n][TOPK_MAX_K]
    void synchronize();

    void kv_clear();
    bool kv_seq_rm(int seq, int p0, int p1);
    void kv_seq_cp(int src, int dst, int p0, int p1);
    void kv_seq_add(int seq, int p0, int p1, int delta);
    int kv_used_cells() const;

    int debug_layer_out(int il, float *dst, size_t cap);
    void set_debug_taps(bool on) { debug_taps_ = on; }
    void set_profile(bool on) { profile_ = on; }
    const std::vector<ProfileEntry> &last_profile() const { return last_profile_; }
    double bench_weight_sweep(int iters, uint64_t *bytes, int *launches = nullptr);

    Model *model;
    ContextParams cp;
    uint64_t device_bytes = 0;
    std::string last_error;
    bool embeddings_enabled = false;

  private:
    struct Bufs;
    const float *ub_embd_ = nullptr;   // host rows of the micro-batch being decoded when the batch carries embeddings
    int decode_ubatch(int n, const int32_t *tokens, const int32_t *pos, const int32_t *seq, const uint64_t *seqmask,
                      const int8_t *flags, int out_base);
    int find_slot(int n);
    // one cell per token; with n_seq_max > 1 every sequence allocates from its own region of the cache
    // (n_ctx / n_seq_max cells, the reference's per-slot context), so a sequence's cells stay together
    bool alloc_cells(int n, const uint64_t *seqmask, std::vector<int> &out);
    std::vector<int> region_next_;     // per sequence: where its next cell is looked for first
    void apply_k_shift();
    hipError_t run_layers(int T, int n_kv_cap);
    hipError_t run_layers_encoder(int T, int n_kv_cap);
    hipError_t run_output(int n_out, int out_base);
    hipError_t linear(const DevTensor &w, const ActQuant &aq, const float *x_f32, int K, int T, float *out, int ld_out,
                      const float *resid, int epi);
    hipError_t linear_multi(const DevTensor *const *ws, float *const *outs, int n, const ActQuant &aq, const float *x_f32, int T);
    void prof_mark(const char *name);
    void prof_begin();
    void prof_end();
    void *dalloc(size_t bytes);

    hipStream_t stream_ = nullptr;
    int cur_T_ = 0;                    // tokens of the micro-batch run_layers last ran (run_output: a single row needs no gather)
    std::vector<void *> allocs_;
    std::vector<KVCell> cells_;
    int head_ = 0;
    bool has_shift_ = false;
    bool meta_dirty_ = true;

    // device state
    std::vector<KVLayerView> kv_;
    int32_t *d_cell_pos_ = nullptr;
    uint64_t *d_cell_seq_ = nullptr;
    int32_t *d_delta_ = nullptr;
    // per-ubatch token arrays (device) + pinned host staging
    int32_t *d_pos_open_ = nullptr;     // encoder models: INT_MAX per token - the attention kernels' "cell position <= token position" test always passes
    int32_t *d_tok_ = nullptr, *d_pos_ = nullptr, *d_seq_ = nullptr, *d_cell_ = nullptr, *d_nkv_ = nullptr, *d_outrow_ = nullptr;
    uint64_t *d_seqmask_ = nullptr;
    uint8_t *h_stage_ = nullptr;    // pinned
    size_t stage_bytes_ = 0;
    uint8_t *d_stage_ = nullptr;
    hipEvent_t stage_event_ = nullptr;
    // activations
    float *x_ = nullptr, *xn_ = nullptr, *q_ = nullptr, *k_ = nullptr, *v_ = nullptr, *att_ = nullptr, *ffn_ = nullptr, *ffn_u_ = nullptr;
    float *xo_ = nullptr, *router_ = nullptr, *moe_out_ = nullptr;
    float *tp_part_ = nullptr, *tp_logits_ = nullptr;   // row split: this rank's partial sums / logits slices before the exchange
    size_t tp_logits_rows_ = 0;
    hipError_t tp_reduce_into_x(int T);                 // x_ = sum over ranks of tp_part_
    int32_t *moe_ids_ = nullptr;
    float *moe_w_ = nullptr;
    // prompt batches of a mixture-of-experts model: (token, rank) pairs grouped by expert (run_layers)
    ActQuant aq_eg_, aq_ffg_;
    float *ffn_g_ = nullptr, *ffn_ug_ = nullptr, *y_g_ = nullptr;
    int32_t *moe_meta_ = nullptr, *moe_slot_ = nullptr, *moe_tok_ = nullptr, *h_moe_meta_ = nullptr;
    ActQuant aq_e_, aq_ff_, aq_o_;
    int8_t *mmq_bh_ = nullptr, *mmq_bl_ = nullptr;   // (hi, lo) planes of the 32-code block sums for the MFMA path
    MMQWorkspace mmq_ws_;                            // partial sums of the K-split prompt contraction
    // whose block sums mmq_bh_ / mmq_bl_ currently hold (code plane pointer, K, rows): the quantisers of a prompt batch write
    // the planes themselves, launch_mmq_prep runs only when they are not there (ensure_prep)
    const void *prep_owner_ = nullptr;
    int prep_K_ = 0, prep_T_ = 0;
    hipError_t ensure_prep(const ActQuant &aq, int K, int T);
    void prep_written(const ActQuant &aq, int K, int T) { prep_owner_ = aq.qs; prep_K_ = K; prep_T_ = T; }
    bool batch_distinct_ = false;                    // the micro-batch is one token each of different sequences (decode_ubatch)
    const int8_t *bh_over_ = nullptr, *bl_over_ = nullptr;   // linear(): block-sum planes prepared by the caller for a slice of a larger batch
    float *att_part_ = nullptr;
    size_t att_part_floats_ = 0;
    int cur_max_pos_ = 0;                            // largest position of the micro-batch being decoded
    float *d_embd_ = nullptr, *h_embd_ = nullptr;   // [n_ubatch][n_embd], embeddings mode
    bool embd_fetched_ = false, last_was_embd_ = false;
    // batched single-token steps: per-token lists of the 64-cell chunks that hold cells of the token's sequence
    int32_t *h_chunks_ = nullptr, *d_chunks_ = nullptr;     // [64][chunk_stride_] lists, then [64] counts
    int chunk_stride_ = 0, chunk_lmax_ = 0, chunk_cap_ = 0;   // cap = grid size used for the lists (>= lmax; rounded up for graph reuse)                 // lmax = longest list of the current batch (0 = lists not in use)
    // whole-step kernel (decode_mega.hip): per-layer descriptors in device memory, barrier words, pinned time-out flag
    MegaLayer *d_mega_layers_ = nullptr;
    unsigned *d_mega_sync_ = nullptr;
    int *h_mega_flag_ = nullptr;
    unsigned long long *d_mega_probe_ = nullptr;   // MI355_MEGA_PROBE=1: phase time stamps of the last step (printed by the destructor)
    size_t mega_lds_ = 0;
    bool last_layers_mega_ = false;      // what the last run_layers call issued
    std::map<hipGraphExec_t, bool> graph_is_mega_, graph_is_engine_;
    bool last_layers_engine_ = false;
    int mega_state_ = 0;                 // 0 = not looked at yet, 1 = descriptors built, -1 = this model / context takes the per-launch path
    bool mega_prepare();
    bool mega_check();                   // after a stream sync: false (and last_error set) if a barrier of the last step timed out
    // layer engine (decode_engine.hip): attn_output -> gate | up -> down -> next Q | K | V of a single-token step in one persistent launch per layer
    EngineLayer *d_engine_layers_ = nullptr;
    unsigned long long *d_engine_gran_ = nullptr;      // hand-over granules
    unsigned *d_engine_epoch_ = nullptr;               // step serial (incremented by step_setup)
    unsigned long long *d_engine_probe_ = nullptr;     // MI355_ENGINE_PROBE=<layer>: wall-clock stamps of that layer's launch (printed by the destructor)
    int engine_probe_layer_ = -1;
    unsigned err_epoch_seen_ = 0;                      // stream_check: the process-wide error epoch this context has already answered for
    unsigned long long first_unchecked_launch_ = 0;    // process-wide serial of this context's first step launch since its last stream_check (0: none)
    bool logits_on_host_ = false;                      // run_output: the head's launch stored the flagged row into h_logits_ itself (no copy node behind the step)
    bool attn_out_off_ = false;                        // after an answered error epoch: the two-launch attention path (nothing in it waits for another workgroup)
    int engine_state_ = 0;                             // 0 = not looked at yet, 1 = ready, -1 = this model / context takes one launch per mat-vec
    bool engine_prepare();
    // One context at a time per device may have a step with the cross-workgroup-wait attention kernel in flight: two such kernels placed on the same CUs at the
    // same time can hold each other's item workgroups out (attn_out.hip; seen between PROCESSES in round 6, profiles/r6_tp_shared_device_trace.txt - two models
    // of one server decoding concurrently are the in-process form of it).  A context that finds the device taken runs that step on the wait-free launches.
    bool fused_acquire();
    void fused_release();                              // called wherever this context has just drained its stream
    bool holds_fused_ = false, attn_out_skip_step_ = false;
    bool stream_check();                               // after a stream sync: false (and last_error set) if a bounded wait of a stream / engine kernel gave up
    int32_t *d_moe_forced_ = nullptr;    // force_moe_ids: [n_layer][T][k] on the device (armed while moe_forced_T_ > 0)
    int moe_forced_T_ = 0, moe_forced_cap_ = 0;
    unsigned *att_counters_ = nullptr;   // per-kv-head arrival tickets of the fused decode attention (zero between launches)
    unsigned *d_step_serial_ = nullptr;  // device word: serial number of the step (incremented by the step's set-up launch; the hand-over tags / flags of attn_out.hip and decode_engine.hip)
    unsigned long long *d_qkv_gran_ = nullptr;  // attn_out.hip QF (round 6): the token's q | k | v as tagged granules inside the one launch per layer's attention block
    unsigned long long *d_ao_gran_ = nullptr;   // attn_out.hip: the quantised attention output as tagged granules (shared by the layers of a step)
    unsigned *d_ao_flags_ = nullptr;     // attn_out.hip: [n_layer][64 * ATT_SYNC_STRIDE] flag words, one per merge ticket group (they hold the serial of the step that raised them)
    float *argmax_scratch_ = nullptr, *rope_cs_ = nullptr;
    void *topk_scratch_ = nullptr;                     // launch_topk_rows workspace (own allocation, sized for topk_rows_cap_ rows on first use)
    uint8_t *d_topk_adj_ = nullptr, *h_topk_adj_ = nullptr;   // [cap] TopkAdj + [cap] row numbers: device copy and its pinned staging
    int topk_rows_cap_ = 0;
    unsigned long long *h_topk_ = nullptr;             // pinned: the winning keys
    Fuse pending_fuse_;
    int att_splits_ = 1;
    float *dbg_ = nullptr;
    bool debug_taps_ = false;
    int dbg_tokens_ = 0;
    // outputs
    float *d_logits_ = nullptr, *h_logits_ = nullptr;
    int32_t *d_argmax_ = nullptr, *h_argmax_ = nullptr;
    size_t logits_cap_rows_ = 0;
    std::vector<int> out_row_of_batch_;   // batch index -> output row or -1
    int n_out_last_ = 0;
    bool logits_fetched_ = false, argmax_fetched_ = false;
    // graph for the single-token decode step
    hipGraphExec_t graph_exec_ = nullptr;
    std::map<int, hipGraphExec_t> graphs_;   // n_kv bucket -> captured decode step
    // profiling
    bool profile_ = false;
    std::vector<std::pair<std::string, hipEvent_t>> prof_events_;
    std::vector<ProfileEntry> last_profile_;
    struct KTimer;                                     // profile mode: per-kernel begin / end events of the stream and attention launches (set_kernel_timer)
    KTimer *ktimer_ = nullptr;
    int n_kv_ = 0;
};

}  // namespace mi355
