/*
 * oracle/oq_ops.c — the graph ops under llama_decode() as the ggml CPU backend computes them.
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle.h).
 *
 * Upstream homes (absent from /root/reference; restated from SURVEY.md §A.2/§A.3):
 *   ggml/src/ggml-cpu/ggml-cpu.c   ggml_compute_forward_mul_mat (activation -> vec_dot_type, per-row vec_dot)
 *   ggml/src/ggml-cpu/ops.cpp      rms_norm, rope, soft_max, flash_attn_ext_f16, get_rows, silu
 * Reference call site that reaches them: src/llama_server_context.cc:1635 (llama_decode).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void oq_mul_mat(int type, const void *W, int64_t N, int64_t K,
                const float *x, int64_t T, float *y, int nth) {
    const int vt = oq_vec_dot_type(type);
    const size_t wrow = oq_row_bytes(type, K);
    const size_t arow = oq_row_bytes(vt, K);
    uint8_t *act = (uint8_t *)malloc(arow * (size_t)T);
    for (int64_t t = 0; t < T; t++) oq_quantize_row(vt, x + t * K, act + (size_t)t * arow, K);
    if (nth < 1) nth = 1;
#pragma omp parallel for num_threads(nth) schedule(static)
    for (int64_t r = 0; r < N; r++) {
        const uint8_t *wr = (const uint8_t *)W + (size_t)r * wrow;
        for (int64_t t = 0; t < T; t++) y[t * N + r] = oq_vec_dot(type, K, wr, act + (size_t)t * arow);
    }
    free(act);
}

void oq_rms_norm(const float *x, float *y, int64_t n, float eps) {
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) sum += (double)(x[i] * x[i]);
    const float mean = (float)(sum / (double)n);
    const float scale = 1.0f / sqrtf(mean + eps);
    for (int64_t i = 0; i < n; i++) y[i] = x[i] * scale;
}

void oq_mul_f32(const float *a, const float *b, float *y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = a[i] * b[i]; }
void oq_add_f32(const float *a, const float *b, float *y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = a[i] + b[i]; }
void oq_silu_f32(const float *x, float *y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = x[i] / (1.0f + expf(-x[i])); }

void oq_soft_max(const float *x, const float *mask, float *y, int64_t n, float scale) {
    float mx = -INFINITY;
    for (int64_t i = 0; i < n; i++) {
        y[i] = x[i] * scale + (mask ? mask[i] : 0.0f);
        if (y[i] > mx) mx = y[i];
    }
    double sum = 0.0;
    for (int64_t i = 0; i < n; i++) {
        const float e = expf(y[i] - mx);
        y[i] = e;
        sum += (double)e;
    }
    const float inv = (float)(1.0 / sum);
    for (int64_t i = 0; i < n; i++) y[i] *= inv;
}

/* cos/sin table for one position: theta_0 = pos, theta_{i+1} = theta_i * theta_scale (iterated in f32).  y != NULL with ext_factor != 0 is YaRN
 * (upstream: ggml/src/ggml-cpu/ops.cpp rope_yarn + rope_yarn_ramp): the interpolated angle freq_scale * theta and the original one are mixed per pair by a
 * ramp over the pair index, and cos / sin are scaled by attn_factor * (1 + 0.1 ln(1 / freq_scale)). */
static void rope_table(float *cs, int n_rot, int32_t pos, float freq_base, float freq_scale, const float *ff, const oq_yarn *y) {
    const float theta_scale = powf(freq_base, -2.0f / (float)n_rot);
    float theta = (float)pos;
    for (int i = 0; i < n_rot; i += 2) {
        const float f = ff ? ff[i / 2] : 1.0f;
        const float extrap = theta / f;
        float th = freq_scale * extrap, m = y ? y->attn_factor : 1.0f;
        if (y && y->ext_factor != 0.0f) {
            const float d = y->corr_hi - y->corr_lo;
            const float r = ((float)(i / 2) - y->corr_lo) / (d > 0.001f ? d : 0.001f);
            const float mix = (1.0f - (r < 0.0f ? 0.0f : r > 1.0f ? 1.0f : r)) * y->ext_factor;
            th = th * (1.0f - mix) + extrap * mix;
            m *= 1.0f + 0.1f * logf(1.0f / freq_scale);
        }
        cs[i] = cosf(th) * m;
        cs[i + 1] = sinf(th) * m;
        theta *= theta_scale;
    }
}

/* ggml_rope_yarn_corr_dims (upstream: ggml/src/ggml.c): the pair indices between which the ramp runs - a pair that turns beta_fast times over the original
 * context keeps its angle, one that turns beta_slow times or fewer is fully interpolated */
void oq_yarn_corr_dims(int n_rot, int n_ctx_orig, float freq_base, float beta_fast, float beta_slow, float *lo, float *hi) {
    const float pi = 3.14159265358979323846f;
    const float a = floorf((float)n_rot * logf((float)n_ctx_orig / (beta_fast * 2.0f * pi)) / (2.0f * logf(freq_base)));
    const float b = ceilf((float)n_rot * logf((float)n_ctx_orig / (beta_slow * 2.0f * pi)) / (2.0f * logf(freq_base)));
    *lo = a > 0.0f ? a : 0.0f;
    *hi = b < (float)(n_rot - 1) ? b : (float)(n_rot - 1);
}

/* build_moe_ffn's selection (upstream: src/llama-graph.cpp): softmax over the router logits, the k largest probabilities in descending order with the
 * first index winning ties (ggml_top_k = argsort), weights renormalised to sum 1.  probs: scratch of n_expert floats. */
void oq_moe_route(const float *logits, int n_expert, int k, float *probs, int32_t *ids, float *w) {
    oq_soft_max(logits, NULL, probs, n_expert, 1.0f);
    uint64_t used = 0;
    for (int j = 0; j < k; j++) {
        int best = -1;
        for (int e = 0; e < n_expert; e++)
            if (!(used & (1ull << e)) && (best < 0 || probs[e] > probs[best])) best = e;
        used |= 1ull << best;
        ids[j] = best; w[j] = probs[best];
    }
    float wsum = 0.0f;
    for (int j = 0; j < k; j++) wsum += w[j];
    for (int j = 0; j < k; j++) w[j] /= wsum;
}

void oq_rope_ext(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                 float freq_base, float freq_scale, const float *ff, int neox, const oq_yarn *y) {
    float cs[1024];
    if (n_rot > 1024) abort();
    rope_table(cs, n_rot, pos, freq_base, freq_scale, ff, y);
    for (int h = 0; h < n_head; h++) {
        float *p = x + (size_t)h * head_dim;
        for (int i = 0; i < n_rot; i += 2) {
            const int a = neox ? i / 2 : i, b = neox ? i / 2 + n_rot / 2 : i + 1;
            const float x0 = p[a], x1 = p[b];
            p[a] = x0 * cs[i] - x1 * cs[i + 1];
            p[b] = x0 * cs[i + 1] + x1 * cs[i];
        }
    }
}

void oq_rope_norm(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                  float freq_base, float freq_scale, const float *ff) {
    oq_rope_ext(x, n_head, head_dim, n_rot, pos, freq_base, freq_scale, ff, 0, NULL);
}

void oq_rope_neox(float *x, int n_head, int head_dim, int n_rot, int32_t pos,
                  float freq_base, float freq_scale, const float *ff) {
    oq_rope_ext(x, n_head, head_dim, n_rot, pos, freq_base, freq_scale, ff, 1, NULL);
}

void oq_get_rows(int type, const void *table, int64_t row_elems, const int32_t *ids, int64_t n_ids, float *dst) {
    const size_t rb = oq_row_bytes(type, row_elems);
    for (int64_t i = 0; i < n_ids; i++)
        oq_dequantize_row(type, (const uint8_t *)table + (size_t)ids[i] * rb, dst + i * row_elems, row_elems);
}

/*
 * flash_attn_ext, one query token (upstream: ggml_compute_forward_flash_attn_ext_f16).
 * Q is converted to K's vec_dot_type (f16 for f16 K, q8_0 for q8_0/q4_0 K); scores are
 * accumulated with an online softmax one cell at a time; V is accumulated in fp16 when the
 * V cache is f16 and in f32 (after dequantising the row) otherwise.
 */
static int g_fa_v_acc_f32 = 0;
void oq_set_fa_v_acc_f32(int v) { g_fa_v_acc_f32 = v; }

void oq_flash_attn_ext(const float *q, int n_head, int n_head_kv, int dk, int dv,
                       int type_k, const void *k, size_t k_row_stride, size_t k_head_stride,
                       int type_v, const void *v, size_t v_row_stride, size_t v_head_stride,
                       const int32_t *cells, int n_cells, float scale, float *out) {
    const int qt = oq_vec_dot_type(type_k);
    const int gqa = n_head / n_head_kv;
    uint8_t *qq = (uint8_t *)malloc(oq_row_bytes(qt, dk) + 16);
    float *acc32 = (float *)malloc(sizeof(float) * (size_t)dv);
    float *v32 = (float *)malloc(sizeof(float) * (size_t)dv);
    uint16_t *acc16 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)dv);
    for (int h = 0; h < n_head; h++) {
        const int g = h / gqa;
        oq_quantize_row(qt, q + (size_t)h * dk, qq, dk);
        float S = 0.0f, M = -INFINITY;
        if (type_v == OQ_TYPE_F16 && !g_fa_v_acc_f32) memset(acc16, 0, sizeof(uint16_t) * (size_t)dv);
        else memset(acc32, 0, sizeof(float) * (size_t)dv);
        for (int ci = 0; ci < n_cells; ci++) {
            const int32_t c = cells[ci];
            const uint8_t *kr = (const uint8_t *)k + (size_t)c * k_row_stride + (size_t)g * k_head_stride;
            const uint8_t *vr = (const uint8_t *)v + (size_t)c * v_row_stride + (size_t)g * v_head_stride;
            float s = oq_vec_dot(type_k, dk, kr, qq);
            s = s * scale; /* mask value for a visible cell is 0 */
            const float Mold = M;
            float ms = 1.0f, vs = 1.0f;
            if (type_v == OQ_TYPE_F16 && !g_fa_v_acc_f32) {
                const uint16_t *vh = (const uint16_t *)vr;
                if (s > M) {
                    M = s;
                    ms = expf(Mold - M);
                    for (int d = 0; d < dv; d++) acc16[d] = oq_fp32_to_fp16(oq_fp16_to_fp32(acc16[d]) * ms);
                } else {
                    vs = expf(s - M);
                }
                for (int d = 0; d < dv; d++)
                    acc16[d] = oq_fp32_to_fp16(oq_fp16_to_fp32(acc16[d]) + oq_fp16_to_fp32(vh[d]) * vs);
            } else {
                if (s > M) {
                    M = s;
                    ms = expf(Mold - M);
                    for (int d = 0; d < dv; d++) acc32[d] *= ms;
                } else {
                    vs = expf(s - M);
                }
                oq_dequantize_row(type_v, vr, v32, dv);
                for (int d = 0; d < dv; d++) acc32[d] += v32[d] * vs;
            }
            S = S * ms + vs;
        }
        if (type_v == OQ_TYPE_F16 && !g_fa_v_acc_f32)
            for (int d = 0; d < dv; d++) acc32[d] = oq_fp16_to_fp32(acc16[d]);
        const float inv = 1.0f / S;
        for (int d = 0; d < dv; d++) out[(size_t)h * dv + d] = acc32[d] * inv;
    }
    free(qq); free(acc32); free(v32); free(acc16);
}
