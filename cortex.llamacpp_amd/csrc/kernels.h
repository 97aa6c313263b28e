// kernels.h — host-visible launch interface of the gfx950 kernels (internal; the public C-ABI is
// include/mi355_llama.h).
#pragma once

#include "dev_common.h"

namespace mi355 {

// ---------------------------------------------------------------- quantised activations (device)
// Q8_K: codes + one f32 scale per 256 + int16 sums per 16 (ggml block_q8_K split into planes);
// Q8_0: codes + one f16 scale per 32 (ggml block_q8_0 split into planes).
struct ActQuant {
    int8_t *qs = nullptr;      // [T][K]   q8_K
    float *d = nullptr;        // [T][K/256]
    int16_t *bsums = nullptr;  // [T][K/16]
    int8_t *qs0 = nullptr;     // [T][K]   q8_0
    uint16_t *d0 = nullptr;    // [T][K/32]
};

// ---------------------------------------------------------------- mmvq
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_SWIGLU = 2 };

struct MMVQSeg {
    const uint8_t *W;     // device-layout rows
    float *out;           // out[t * ld_out + row]
    const float *resid;   // EPI_ADD: out = resid + y (same indexing)
    int type;
    int n_rows;
    int ld_out;
    size_t row_bytes;
    const int32_t *expert_sel;   // MoE: device pointer to the expert index (nullptr = dense)
    size_t expert_stride;        // bytes between experts
};

struct MMVQArgs {
    MMVQSeg seg[3];
    int seg_block0[4];    // blocks [seg_block0[s], seg_block0[s+1]) serve segment s
    int n_seg;
    int K;
    int T;                // tokens in this launch (1, 2 or 4)
    int epi;
    int need_q8k, need_q80;
    int nck;              // waves of a workgroup that share one row pair (split over K)
    int red_off;          // LDS byte offset of the reduction scratch
    const int8_t *aq; const float *ad; const int16_t *abs;   // q8_K planes for the T tokens
    const int8_t *aq0; const uint16_t *ad0;                  // q8_0 planes
    // fused prologue (T == 1, K <= 8192, K % 1024 == 0): 0 = planes above; 1 = rms_norm(nx)*nw then quantise; 2 = quantise nx
    int fuse_mode;
    const float *nx; const float *nw; float neps;
    // single-token mixture-of-experts step, fast path only: n_sel > 1 runs the n_sel selected experts of ONE token in one
    // launch — the workgroups are divided evenly among them; expert j reads expert_sel[j], writes out + j * sel_out_stride
    // and (fuse_mode 2) quantises nx + j * sel_nx_stride (all strides in floats).  0 / 1 = one expert (expert_sel[0]).
    int n_sel, sel_out_stride, sel_nx_stride;
    // single-token weight-stream launches with ONE segment (the output head): every result is also stored at out_host[row] - pinned host memory the device can
    // write (nullptr: no copy).  The logits row then crosses PCIe under the launch instead of in a copy behind it.
    float *out_host;
};

hipError_t launch_mmvq(MMVQArgs a, hipStream_t st);
// single-token fast path (mmvq_fast.hip): persistent, software-pipelined; K % 2048 == 0, K-quant types
void mmvq_fast_set_threads(int nt);
// contraction lengths the register-ring mat-vec is instantiated for, in passes of 2048 (the last pass may be partial): 1-4 hidden sizes up to 8192, the rest
// feed-forward widths (5: Qwen2-1.5B 8960; 6: 11008; 7: 13824 / 14336; 9: 17920; 10: 18944 / 20480; 11: 22016; 14: 27648 / 28672; 15: 29568)
inline bool mmvq_fast_kb_ok(int kb) { return (kb >= 1 && kb <= 7) || kb == 9 || kb == 10 || kb == 11 || kb == 14 || kb == 15; }
bool mmvq_fast_applicable(const MMVQArgs &a);
hipError_t launch_mmvq_fast(MMVQArgs a, hipStream_t st);
hipError_t launch_mmvq_ints(MMVQArgs a, int32_t *isum, int32_t *msum, hipStream_t st);
// workgroup ranges per segment, LDS bytes and the reduction-scratch offset for a grid of `blocks` workgroups of `waves`
// waves each (what launch_mmvq_fast works out for its own launch); returns the LDS bytes, 0 if the shape has no fast form
size_t mmvq_fast_plan(MMVQArgs &a, int blocks, int waves);
// single-token weight-stream form (mmvq_stream.hip): HBM -> LDS DMA rings, bit-identical to the fast path
void mmvq_set_stream(bool on);                            // tests: route single-token mat-vecs to mmvq_fast instead
bool mmvq_stream_applicable(const MMVQArgs &a);
void mmvq_stream_set_anyorder_for_timing(bool on);   // tools/exp_stream.hip only: barrier-less dispatch, results undefined
hipError_t launch_mmvq_stream(MMVQArgs a, hipStream_t st);

// workgroup ranges per segment for a grid of `blocks` workgroups, in proportion to the segments' bytes (what launch_mmvq_stream works out for its own launch)
void mmvq_stream_plan(MMVQArgs &a, int blocks);
// the sticky error word of the weight-stream kernels (pinned host memory, or nullptr): a bounded wait that gives up ORs a code into it
void mmvq_stream_set_error_word(unsigned *w);
// Measurement aid (bench.py `roles_live`): while a timer is set, every launch of the weight-stream kernel and of the attention + attn_output kernel is made with
// hipExtLaunchKernelGGL's start / stop events - the dispatch's own begin / end timestamps, what rocprofv3's kernel trace reports - taken from the timer by role
// ("qkv", "gate_up", "ffn_down", "head", "stream" for any other shape, "attn_out").  Process-wide and not synchronised: set it around a profiled step only.
struct KernelTimer {
    virtual ~KernelTimer() = default;
    virtual bool next(const char *role, hipEvent_t *start, hipEvent_t *stop) = 0;     // false: launch untimed
};
void set_kernel_timer(KernelTimer *t);
KernelTimer *kernel_timer();

// the two experiment translation units below (decode_engine.hip, decode_mega.hip) are in this library (build.py, MI355_BUILD_EXPERIMENTS=1); false: experiments_absent.cc
bool experiments_built();
// ---------------------------------------------------------------- one decoder layer's mat-vecs in one persistent launch (decode_engine.hip; experiment, see above)
// attn_output -> gate | up -> down -> the next layer's Q | K | V, hand-overs through tagged granules; descriptors in device memory, each planned with
// mmvq_stream_plan(.., num_cu()).  has_qkv = 0: the launch ends behind ffn_down (last layer).
struct EngineLayer {
    MMVQArgs wo, gu, dn, qkv;
    int has_qkv;
};
bool decode_engine_applicable(const EngineLayer &l, int E, int FF);   // on UNPLANNED descriptors (as launch_mmvq_stream would get them)
void decode_engine_plan(EngineLayer &l);                                // mmvq_stream_plan of the four mat-vecs for num_cu() workgroups
size_t decode_engine_granule_words(int E, int FF);          // 8-byte words of hand-over space a context needs
void decode_engine_set_error_word(unsigned *w);
// epoch_dev: device word holding the step serial (changes every step: step_setup); probe: nullable, num_cu() * 10 * 48 stamps
hipError_t launch_decode_engine(const EngineLayer *layer_dev, int E, int FF, unsigned long long *granules, const unsigned *epoch_dev, int layer_index,
                                unsigned long long *probe, hipStream_t st);

void set_num_cu(int n);
int num_cu();

// ---------------------------------------------------------------- batched prefill contraction on MFMA (mmq.hip)
bool mmq_applicable(int type, int K, int T);
void mmq_set_tiles(int mt);                                // tools: force 1 / 2 / 4 token tiles per wave (0 = by T)
// pre-expanded MFMA operand planes of a weight tensor (2 B / weight, built once at load; mmq.hip)
size_t mmq_planes_bytes(int type, int64_t n_rows, int K);   // 0 if the type has no planes form
// Q4_0 / Q5_0 / IQ4_NL: an exact Q8_0-layout copy of the tensor for prompt batches (mmq_q80.hip); 0 for other types
size_t mmq_q80_copy_bytes(int type, int64_t n_rows, int K);
hipError_t launch_expand_q80_copy(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, uint8_t *dst, hipStream_t st);
hipError_t launch_mmq_expand(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, uint8_t *planes, hipStream_t st);
// workspace for the K-split form of the planes kernel (tensors with few rows): n_split * T * n_rows floats; none = no split
struct MMQWorkspace { float *p = nullptr; size_t bytes = 0; };
void mmq_set_split(int n);                                 // tools: force the K split (0 = by shape)
void mmq_set_lds_form(int on);                             // tools: 0 = never the 128 x 256 LDS kernel, 1 = by shape, -1 = environment / default
// ggml_mul_mat_id on a prompt batch as one launch per projection (mmq.hip, round 5): every expert's grouped batch through the 128 x 256 LDS kernel, the (expert,
// token tile) of a workgroup found on the device from moe_group_kernel's meta words (counts at [0, NE), first grouped rows at [NE, 2 NE)); rows_max = rows of
// the grouped arrays; plane_stride = bytes between two experts' plane sets
bool mmq_planes_moe_ok(int type, int n_rows, int K);
hipError_t launch_mmq_planes_swiglu_moe(int type, const uint8_t *planes_gate, const uint8_t *planes_up, size_t plane_stride, int n_expert, const int32_t *meta,
                                        int n_rows, int K, int rows_max, const ActQuant &q, float *out, int ld_out, hipStream_t st);
hipError_t launch_mmq_planes_moe(int type, const uint8_t *planes, size_t plane_stride, int n_expert, const int32_t *meta, int n_rows, int K, int rows_max,
                                 const ActQuant &q, float *out, int ld_out, hipStream_t st);
hipError_t launch_mmq_planes(int type, const uint8_t *planes, int n_rows, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                             float *out, int ld_out, const float *resid, hipStream_t st, MMQWorkspace wsp = MMQWorkspace());
// small batches (continuous-batching decode steps, 8 <= T <= 64): K split over the waves of a workgroup, GGUF-form weights
struct MMQSeg {
    const uint8_t *W; size_t row_bytes; int n_rows, type;
    float *out; int ld_out; const float *resid;
    int tile0;                 // first workgroup of this segment (set by the launcher)
};
// prompt batches against Q8_0 weights (mmq_q80.hip): int8 MFMA per 32-block, f32 fold in block order (bit-exact with the CPU)
bool mmq_q80_applicable(int type, int K, int T);
hipError_t launch_mmq_q80(const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q, float *out, int ld_out,
                          const float *resid, hipStream_t st);
// up to three tensors of one plane format (Q4_K / Q5_K together, or Q6_K) whose plane sets are contiguous in memory, as one
// launch over the concatenated rows; seg_rows[i] rows go to outs[i] (leading dimension lds_out[i])
hipError_t launch_mmq_planes_multi(int type, const uint8_t *planes, const int *seg_rows, float *const *outs, const int *lds_out, int n_seg, int K, int T,
                                   const ActQuant &q, const int8_t *bh, const int8_t *bl, const float *resid, hipStream_t st,
                                   MMQWorkspace wsp = MMQWorkspace(), const int *seg_types = nullptr);
// seg_types (per segment; nullptr = all `type`): Q4_K / Q5_K segments beside Q6_K ones in ONE launch where this says so (segments end on
// 128-row tiles and the launch takes the 128 x 256 kernel)
bool mmq_planes_mixed_ok(const int *seg_rows, int n_seg, int K, int T, MMQWorkspace wsp);
// ffn_gate and ffn_up (one plane type, n_rows rows each) in one launch with SwiGLU in the epilogue: out[t][row] = silu(gate . x) * (up . x)
bool mmq_planes_swiglu_ok(int type_gate, int type_up, int n_rows, int K, int T);
hipError_t launch_mmq_planes_swiglu(int type, const uint8_t *planes_gate, const uint8_t *planes_up, int n_rows, int K, int T, const ActQuant &q,
                                    float *out, int ld_out, hipStream_t st);
bool mmq_ksplit_applicable(int type, int K, int T);
// ... and preferred over the planes kernels for a tensor of n_rows rows (has_planes: its pre-expanded planes exist)
bool mmq_ksplit_preferred(int type, int n_rows, int K, int T, bool has_planes, bool pair = false);
hipError_t launch_mmq_ksplit_multi(const MMQSeg *segs, int n_seg, int K, int T, const ActQuant &q, const int8_t *bh, const int8_t *bl,
                                   bool swiglu, hipStream_t st);
hipError_t launch_mmq_ksplit(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q,
                             const int8_t *bh, const int8_t *bl, float *out, int ld_out, const float *resid, hipStream_t st);
size_t mmq_prep_bytes(int K, int T);                       // bytes of each of the two block-sum planes
hipError_t launch_mmq_prep(const ActQuant &q, int K, int T, int8_t *bh, int8_t *bl, hipStream_t st);
hipError_t launch_mmq(int type, const uint8_t *W, size_t row_bytes, int n_rows, int K, int T, const ActQuant &q,
                      const int8_t *bh, const int8_t *bl, float *out, int ld_out, const float *resid, hipStream_t st);

// ---------------------------------------------------------------- activation-side kernels (act.hip)
// y = rms_norm(x) * w  for T rows of n; optionally f32 out and/or q8_K / q8_0 planes
// bh / bl (optional, with q8_K): also write the block sums split into the int8 planes the MFMA kernels contract
// (bsum = 64 * bh + bl, what launch_mmq_prep computes as its own launch), [T][n / 16] each
hipError_t launch_rmsnorm_quant(const float *x, const float *w, int n, int T, float eps,
                                float *y_f32 /*nullable*/, const ActQuant *q /*nullable*/, bool want_q8k, bool want_q80,
                                hipStream_t st, int8_t *bh = nullptr, int8_t *bl = nullptr);
// quantise f32 rows
hipError_t launch_quantize(const float *x, int n, int T, const ActQuant &q, bool want_q8k, bool want_q80, hipStream_t st,
                           int8_t *bh = nullptr, int8_t *bl = nullptr);
// LayerNorm rows with weight and bias (encoder models): y[t] = (x[t] - mean) * rsqrt(var + eps) * w + b; y may be x
hipError_t launch_layer_norm(const float *x, const float *w, const float *b, int n, int T, float eps, float *y, hipStream_t st);
// the same, and the rows once more rounded to f16 ([T][n] halves)
hipError_t launch_layer_norm_h(const float *x, const float *w, const float *b, int n, int T, float eps, float *y, void *yh, hipStream_t st);
hipError_t launch_swiglu(const float *g, const float *u, float *y, int64_t n, hipStream_t st);
// silu(g) * u quantised for the next mat-mul without an f32 round trip (n % 256 == 0); same blocks as launch_swiglu + launch_quantize
hipError_t launch_swiglu_quant(const float *g, const float *u, int n, int T, const ActQuant &q, bool want_q8k, bool want_q80, hipStream_t st,
                               int8_t *bh = nullptr, int8_t *bl = nullptr);
hipError_t launch_add(const float *a, const float *b, float *y, int64_t n, hipStream_t st);
// x[t][i] += bias[i] for the T rows of Q (nq wide), K and V (nkv wide); null biases are skipped
hipError_t launch_add_qkv_bias(float *q, float *k, float *v, const float *bq, const float *bk, const float *bv, int nq, int nkv, int T, hipStream_t st);
hipError_t launch_soft_max(const float *x, const float *mask, float *y, int n, int rows, float scale, hipStream_t st);
// pack device planes back into ggml blocks (parity tests)
hipError_t launch_pack_q8k_blocks(const ActQuant &q, int n, int T, uint8_t *blocks, hipStream_t st);
hipError_t launch_pack_q80_blocks(const ActQuant &q, int n, int T, uint8_t *blocks, hipStream_t st);

// ---------------------------------------------------------------- layout / lookup kernels (misc.hip)
// regroup ggml rows into the device row layout (Q6_K, Q8_0); other types are byte copies
hipError_t launch_repack_rows(int type, const uint8_t *src_ggml, uint8_t *dst_dev, int64_t K, int64_t n_rows, hipStream_t st);
// dst[i][:] = dequant(table row ids[i]) from device-layout rows
hipError_t launch_get_rows(int type, const uint8_t *table_dev, int64_t K, const int32_t *ids, int n_ids, float *dst, hipStream_t st);
// scratch: cap_rows * 129 words, zero-filled once: [cap_rows ticket words | cap_rows * 64 part values | cap_rows * 64 part indices]; rows <= cap_rows
hipError_t launch_argmax_rows(const float *x, int n, int rows, int32_t *out, float *scratch, int cap_rows, hipStream_t st);
// ---- device-side sampling front end (reference call site common_sampler_sample, src/llama_server_context.cc:1679-1698): the k best
// candidates of a logits row after logit_bias and the repetition / frequency / presence penalties, so that k (token, logit) pairs cross to the host
// instead of the 513 KB row.  Order = the host sampler's (host/sampling.cc `better`): higher logit first, lower token id on ties.  Exact: the same f32
// operations on the adjusted tokens (l + bias; l <= 0 ? l * repeat : l / repeat; l -= count * freq + present), keys compared as integers.
constexpr int TOPK_MAX_K = 128, TOPK_MAX_ADJ = 192;
struct TopkAdj {                        // one per row, in device memory
    int n;                              // adjusted tokens
    float repeat, freq, present;
    int tok[TOPK_MAX_ADJ];
    float bias[TOPK_MAX_ADJ];           // 0 = none
    int cnt[TOPK_MAX_ADJ];              // occurrences in the penalty window (0 = not penalised)
};
size_t topk_scratch_bytes(int n);       // workspace PER ROW of n logits
// n_rows rows of `base` ([.][n] f32; row r of the launch = base row rows_dev[r], adjustments adjs_dev[r]) in one set of launches: the k best of each
// into keys_out[r * TOPK_MAX_K + 0..k) (device or pinned host memory), key = (order-preserving image of the f32 logit) << 32 | (0xffffffff - token)
hipError_t launch_topk_rows(const float *base, int n, int n_rows, const int *rows_dev, int k, const TopkAdj *adjs_dev, void *scratch, unsigned long long *keys_out,
                            hipStream_t st);

// f32 / f16 weight mat-vec (router, unquantised models): y[t][r] = dot(W[r], x[t])
// batches against F16 weights on the matrix cores (mmf.hip): y[t][n] (+ resid) = sum_k W[n][k] * f16(x[t][k]); the same products as launch_mmv_float, f32 accumulation
// ---------------------------------------------------------------- LLaVA image encoder (clip.hip): the tower's small kernels; the projections run on launch_mmf16
hipError_t launch_clip_im2col(const float *img, int S, int P, int ld, float *patches, hipStream_t st);            // [n_patches][ld], columns (c, ky, kx), zero padded
hipError_t launch_clip_embed(const float *patch, const float *cls, const float *pos, int E, int T, float *emb, hipStream_t st);   // [class ; patches] + positions
hipError_t launch_clip_bias(float *x, const float *b, int n, int T, float scale, bool do_scale, hipStream_t st);  // x = (x + b) [* scale]
// (out_h / xh, optional: the result once more, rounded to f16 - what the f16 GEMM that consumes it reads)
hipError_t launch_clip_attn(const float *q, const float *k, const float *v, int T, int H, int D, float *out, void *out_h, int n_img, hipStream_t st);     // unmasked, q pre-scaled; n_img images of T rows
hipError_t launch_clip_gelu(float *x, size_t n, bool quick, void *xh, hipStream_t st);                            // ggml's f16-table GELU / quick-GELU
bool mmf16_applicable(int type, int n_rows, int K, int T, const void *W, const void *x, const void *y);
hipError_t launch_mmf16(const uint8_t *W, int n_rows, int K, const float *x, int T, float *y, int ld_out, const float *resid, hipStream_t st);
// the same with the activation rows rounded to f16 beforehand (once per row instead of once per workgroup that reads it): xh [T][K] halves
hipError_t launch_f32_to_f16(const float *x, void *y, size_t n, hipStream_t st);
// y = resid + (W xh + bias) * scale - bias [n_rows], the scale and resid [T][ld_out] each optional (resid may be y itself)
hipError_t launch_mmf16_xh(const uint8_t *W, int n_rows, int K, const void *xh, int T, float *y, int ld_out, const float *resid, const float *bias, float scale, bool do_scale,
                           hipStream_t st);
hipError_t launch_mmv_float(int type, const uint8_t *W, int n_rows, int K, const float *x, int T, float *y, int ld_out,
                            const float *resid, hipStream_t st);

// MoE router: softmax over n_expert logits per token, top-k (first index wins ties), weights renormalised
// forced (nullable, tests): [T][k] expert ids to take instead of the k most probable (weights = this side's probabilities of them, renormalised)
hipError_t launch_moe_route(const float *logits, int T, int n_expert, int k, int32_t *ids, float *w, hipStream_t st, const int32_t *forced = nullptr);
// the two above in one launch (f32 / f16 gate_inp), same arithmetic; logits_out may be null
hipError_t launch_moe_router(int type, const uint8_t *W, int n_expert, int K, const float *x, int T, int k, float *logits_out, int32_t *ids, float *w,
                             hipStream_t st, const int32_t *forced = nullptr);
// x[t][d] += sum_j eo[j][t][d] * w[t][j]   (experts added in rank order, then the residual)
hipError_t launch_moe_combine(float *x, const float *eo, const float *w, int T, int E, int k, size_t eo_stride, hipStream_t st);
hipError_t launch_gather_rows_f32(const float *src, const int32_t *rows, int n_rows, int n, float *dst, hipStream_t st);

// ---- ggml_mul_mat_id for more than a few tokens (build_moe_ffn): the (token, rank) pairs are grouped
// by the expert they selected, every expert then sees ONE contiguous batch of activation rows (its weights are read once
// per batch instead of once per token), and the expert outputs are combined back per token in rank order.
// Counting sort of the T*k selections by expert, stable in (token, rank) order: counts[e], offs[e] (exclusive prefix,
// offs[n_expert] = T*k), slot_of[t*k + j] = grouped row of the pair, tok_of[row] = its token.  meta = counts | offs.
hipError_t launch_moe_group(const int32_t *ids, int T, int k, int n_expert, int32_t *meta, int32_t *slot_of, int32_t *tok_of, hipStream_t st);
// grouped[r] = src[tok_of[r]] for the quantised activation planes that exist in both (K codes per row)
hipError_t launch_moe_gather_act(const ActQuant &src, const int32_t *tok_of, int n_rows, int K, const ActQuant &dst, hipStream_t st);
// x[t][d] += sum_j y[slot_of[t*k + j]][d] * w[t][j]   (same order of operations as launch_moe_combine)
hipError_t launch_moe_scatter_combine(float *x, const float *y, const float *w, const int32_t *slot_of, int T, int E, int k, hipStream_t st);

// ---------------------------------------------------------------- attention side (attn.hip)
// words between two merge-ticket counters / hand-over flags of the decode attention kernels (attn_decode_dev.h, attn_out.hip): one 128-byte line each
constexpr int ATT_SYNC_STRIDE = 32;
struct KVLayerView {
    // head-major cache planes of one layer: cell c of kv-head g
    //   f16 : k + ((g*n_ctx + c) * D) * 2
    //   q8_0: codes k + (g*n_ctx + c) * D ; scales kd + (g*n_ctx + c) * (D/32)
    //   q4_0: codes k + (g*n_ctx + c) * D/2 ; scales kd likewise
    uint8_t *k; uint16_t *kd;
    uint8_t *v; uint16_t *vd;
};

struct RopeArgs {
    int n_rot; float freq_base; float freq_scale; const float *freq_factors; int neox;
    // YaRN: ext_factor 1 mixes the interpolated angle (freq_scale * theta) with the original one per pair, by a ramp over the pair index that falls from 1 at
    // corr_lo to 0 at corr_hi, and multiplies cos / sin by attn_factor * (1 + 0.1 ln(1 / freq_scale)); ext_factor 0 leaves plain (linear) scaling
    float ext_factor = 0.0f, attn_factor = 1.0f, corr_lo = 0.0f, corr_hi = 0.0f;
};

// rope(q) in place, rope(k) -> K cache, v -> V cache for T tokens
hipError_t launch_rope_kv_store(float *q, const float *k, const float *v, int T, int n_head, int n_head_kv, int D,
                                const int32_t *tok_pos, const int32_t *tok_cell, RopeArgs ra,
                                KVLayerView kv, int type_k, int type_v, int n_ctx, const float *cs_table, hipStream_t st);
// cos/sin of every token of the micro-batch: cs_out[T][n_rot] (pairs c,s), reused by all layers
hipError_t launch_rope_table(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, hipStream_t st);
// launch_rope_table + launch_kv_meta_set (below) in one launch
// epoch_word (nullable): device word incremented once per call - the step serial the hand-over tags of decode_engine.hip are built from
hipError_t launch_step_setup(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell,
                             const uint64_t *tok_seqmask, unsigned *zero_word, hipStream_t st, unsigned *epoch_word = nullptr);
// launch_step_setup + launch_get_rows (the batch's embedding rows, dst[T][K]) in one launch
hipError_t launch_step_setup_embed(const int32_t *tok_pos, int T, RopeArgs ra, float *cs_out, int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell,
                                   const uint64_t *tok_seqmask, unsigned *zero_word, unsigned *epoch_word, int type, const uint8_t *table_dev, int64_t K,
                                   const int32_t *ids, float *dst, hipStream_t st);
hipError_t launch_rope_inplace(float *x, int T, int n_head, int D, const int32_t *tok_pos, RopeArgs ra, hipStream_t st);

struct AttnArgs {
    const float *q;          // [T][H][D] (already rotated)
    float *out;              // [T][H][D]
    KVLayerView kv;
    int type_k, type_v;
    int T, H, G, D, n_ctx;
    const int32_t *cell_pos;   // [n_ctx]  -1 = empty
    const uint64_t *cell_seq;  // [n_ctx]  bitmask of sequence ids
    const int32_t *tok_pos;    // [T]
    const int32_t *tok_seq;    // [T]
    const int32_t *n_kv_dev;   // device scalar: cells to scan (high-water mark)
    int n_kv_max;              // host upper bound used to size the grid
    float scale;
    float *part;               // workspace [T][H][splits][D+2]
    int splits;
    int pf_splits = 0;         // prompt kernel (attn_prefill.hip): key splits per query tile, workspace [T][H][pf_splits][D+2]; 0 / 1 = none
    // batched single-token steps: per token, the 64-cell chunks that can hold a visible cell (host-built from the cell
    // table); nullptr = every chunk of the scanned range.  tok_chunks[t * chunk_stride + i], i < tok_nchunks[t] <= splits
    const int32_t *tok_chunks = nullptr;
    const int32_t *tok_nchunks = nullptr;
    int chunk_stride = 0;
    const ActQuant *out_q;     // nullable: also quantise the merged rows (for attn_output)
    bool out_q8k, out_q80;
    int8_t *out_bh = nullptr, *out_bl = nullptr;   // nullable (with out_q, Q8_K): the block sums also as the int8 planes the MFMA kernels take (launch_mmq_prep's)
};
hipError_t launch_flash_attn(const AttnArgs &a, hipStream_t st);
// parity mode for an f16 K / V cache: the reference CPU path's cell-by-cell arithmetic with V accumulated in fp16 (attn.hip); 1 / 0, -1 = MI355_FA_V_ACC=f16
void set_fa_v_acc_f16(int on);
bool fa_v_acc_f16_enabled();
// prompt processing on the matrix cores (attn_prefill.hip): D = 128, q8_0 K / V, T >= 32; q already rotated
bool flash_attn_prefill_applicable(const AttnArgs &a);
int flash_attn_prefill_splits(int T, int H, int G, int D, int n_kv_max);
hipError_t launch_flash_attn_prefill(const AttnArgs &a, hipStream_t st);
// merge of [T][H][splits][D+2] partial records into a.out (+ quantised rows when a.out_q): attn.hip
hipError_t launch_flash_attn_combine(const AttnArgs &a, int splits, hipStream_t st);
// decode-step variants (attn.hip): single round trip per workgroup; q passed UN-rotated (rope fused), NORM rope, D = 128
bool flash_attn_decode_applicable(const AttnArgs &a, const RopeArgs &ra);
int flash_attn_decode_splits(int n_kv_max);
hipError_t launch_flash_attn_decode(const AttnArgs &a, const float *cs_table, RopeArgs ra, hipStream_t st, const float *knew = nullptr,
                                    const float *vnew = nullptr, const int32_t *tok_cell = nullptr);   // knew..: KV store inside (tokens of different sequences)
// one launch per layer for a single-token step: KV store + attention + split merge + quantise (a.splits set by the caller)
bool flash_attn_decode_fused_applicable(const AttnArgs &a, const RopeArgs &ra);
void attn_probe_report();   // MI355_ATTN_PROBE=1: phase timing of the last fused decode attention launch, on stderr
hipError_t launch_flash_attn_decode_fused(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew,
                                          const int32_t *tok_cell, unsigned *counters, hipStream_t st);

// ---------------------------------------------------------------- single-token step: decode attention + attn_output mat-vec in one launch (attn_out.hip)
// K rope + KV store + attention + split merge + Q8_K quantise + W_o mat-vec (+ residual): one workgroup of 8 waves per CU owns a run of W_o rows and at most
// one (kv head, chunk) attention item; the merged codes cross between workgroups behind one flag per kv-head group.  D = 128, NORM rope, q8_0 or f16 cache,
// 1 / 2 / 4 / 8 query heads per kv head, W_o in Q4_K / Q5_K / Q6_K.  a.splits = attn_out_fused_splits(a); flags: one word per ticket group of THIS layer
// (never reset: they hold the tag of the step that raised them); serial: device word the step's set-up launch increments (never 0).
void set_attn_out_fused(int on);                          // tests / A-B runs: 0 = the two launches, 1 = fused, -1 = environment (MI355_ATTN_OUT_FUSED, default on)
bool attn_out_fused_enabled();
int attn_out_fused_chunk(const AttnArgs &a);              // cells per attention item: 64, or 128 from the context length on where 64 would give a CU two items
int attn_out_fused_splits(const AttnArgs &a);
bool attn_out_fused_applicable(const AttnArgs &a, const RopeArgs &ra, const MMVQSeg &wo, int K, int epi);
// gran: attn_out_granule_words(K) 8-byte words shared by the layers of a step (zero once); layer: 0 .. 254, part of the hand-over tag
size_t attn_out_granule_words(int K);
hipError_t launch_attn_out_fused(const AttnArgs &a, const float *cs_table, RopeArgs ra, const float *knew, const float *vnew, const int32_t *tok_cell,
                                 unsigned *counters, unsigned *flags, unsigned long long *gran, int layer, const unsigned *serial, const MMVQSeg &wo, int K, int epi,
                                 hipStream_t st);
void attn_out_set_error_word(unsigned *w);
// Round 6: the layer's Q | K | V mat-vecs (RMSNorm -> Q8_K prologue included) in front of the attention in the SAME launch - one launch per layer for
// [Q | K | V, rope, KV store, attention, merge, Q8_K, attn_output + residual].  seg: attn_q, attn_k, attn_v (Q4_K / Q5_K / Q6_K, device-layout rows; `out` unused:
// the results travel inside the launch); nx: the layer input, nw: attn_norm; K = n_embd = H * D <= 4096; gran: qkv_attn_granule_words(H * D, G * D) 8-byte words
// shared by the layers of a step (zero once).  Outputs are bit-identical to launch_mmvq_stream + launch_attn_out_fused.
struct QKVFuse { MMVQSeg seg[3]; const float *nx, *nw; float neps; int K; unsigned long long *gran; };
void set_qkv_attn_fused(int on);                          // 0 = Q | K | V as a launch of its own, 1 = fused, -1 = environment (MI355_QKV_ATTN_FUSED, default on)
bool qkv_attn_fused_enabled();
size_t qkv_attn_granule_words(int n_q, int n_kv);
// host logic only: LDS bytes (and DMA slots) of the fused launch for a layer of this geometry at a context of n_kv cells; 0 = the two launches
size_t qkv_attn_out_plan_lds(int type_q, int type_k, int type_v, int type_o, int n_embd, int n_head, int n_head_kv, int head_dim, int type_kv, int n_kv, int *slots_out);
bool qkv_attn_out_applicable(const AttnArgs &a, const RopeArgs &ra, const MMVQSeg &wo, int K, int epi, const QKVFuse &q);
hipError_t launch_qkv_attn_out(const AttnArgs &a, const float *cs_table, RopeArgs ra, const int32_t *tok_cell, unsigned *counters, unsigned *flags, unsigned long long *gran,
                               int layer, const unsigned *serial, const MMVQSeg &wo, int K, int epi, const QKVFuse &q, hipStream_t st);
void attn_out_probe_report();                             // MI355_AO_PROBE=1: phase stamps of the last launch, on stderr

// ---------------------------------------------------------------- whole decode step in one launch (decode_mega.hip)
// Every mat-vec of every layer and the decode attention run as PHASES of one persistent kernel, separated by device-wide
// barriers; the first weight blocks of a phase are requested before its barrier.  Dense Llama layers, K-quant weights,
// one token.  The descriptors live in device memory (built once per context); what changes per step is in the launch.
struct MegaLayer {
    MMVQArgs qkv, wo, gate_up, down;   // planned with mmvq_fast_plan(.., mega_blocks(), 4)
    KVLayerView kv;
};
constexpr int MEGA_SYNC_WORDS = 32 * 9;
int mega_blocks();                                           // grid size: 2 workgroups of 256 threads per CU
bool decode_mega_applicable(int kb_e, int kb_ff, int R, int type_k, int type_v);
// a: the single-token AttnArgs of launch_flash_attn_decode_fused (kv is taken from the layer descriptors);
// sync: MEGA_SYNC_WORDS words; [0] and [32 * (1 + g)], g < 8 = barrier counters (0 at launch: launch_kv_meta_set zeroes them),
// sync[1] = sticky time-out flag (stays 0 on a healthy run; copied to the
// pinned host word host_flag when the kernel ends)
hipError_t launch_decode_mega(const MegaLayer *layers_dev, int n_layer, int kb_e, int kb_ff, const AttnArgs &a, const float *cs_table,
                              int n_rot, const float *knew, const float *vnew, const int32_t *tok_cell, unsigned *counters,
                              unsigned *sync, int *host_flag, unsigned long long *probe, size_t lds_mmvq, hipStream_t st);
constexpr int MEGA_PROBES_PER_LAYER = 10;  // probe (nullable): 1 + 10 * n_layer wall-clock stamps of workgroup 0: per phase (qkv, attention, wo, gate_up,
                                           // down) the end of its wait for the previous phase and its own arrival

// prompt batches: rope(q) in place + rope(k) -> cache + v -> cache, vectorised (cs_table required); same results as launch_rope_kv_store
bool rope_q_kv_store_fast_applicable(int H, int G, int D, int type_k, int type_v, const RopeArgs &ra);
hipError_t launch_rope_q_kv_store_fast(float *q, const float *k, const float *v, int T, int H, int G, int D, const float *cs_table, RopeArgs ra,
                                       const int32_t *tok_cell, KVLayerView kv, int type_k, int type_v, int n_ctx, hipStream_t st);
bool kv_store_fast_applicable(int G, int D, int type_k, int type_v, const RopeArgs &ra);
hipError_t launch_kv_store_fast(const float *k, const float *v, int T, int G, int D, const float *cs_table, RopeArgs ra,
                                const int32_t *tok_cell, KVLayerView kv, int type_k, int type_v, int n_ctx, hipStream_t st);
size_t flash_attn_workspace_floats(int T, int H, int D, int splits);
int flash_attn_pick_splits(int T, int G, int n_kv_max);

// re-rotate cached K rows after seq_add (K-shift): for each cell with delta[c] != 0
hipError_t launch_k_shift(KVLayerView kv, int type_k, int G, int D, int n_ctx, const int32_t *delta, RopeArgs ra, hipStream_t st);

// cell metadata update inside the decode graph: cell_pos[cell] = pos, cell_seq[cell] = mask
hipError_t launch_kv_meta_set(int32_t *cell_pos, uint64_t *cell_seq, const int32_t *tok_cell, const int32_t *tok_pos,
                              const uint64_t *tok_seqmask, int T, hipStream_t st, unsigned *zero_word = nullptr);

}  // namespace mi355
