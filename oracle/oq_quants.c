/*
 * oracle/oq_quants.c — block formats, dequantisation, activation quantisation and
 * scalar integer dot products of the ggml CPU backend.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED (see oracle.h).  Restated from the published ggml format
 * specification; upstream homes, absent from /root/reference:
 *   ggml/src/ggml-common.h            block_q4_0/q8_0/q4_K/q5_K/q6_K/q8_K
 *   ggml/src/ggml-quants.c            dequantize_row_*, quantize_row_*_ref
 *   ggml/src/ggml-cpu/ggml-cpu-quants.c  ggml_vec_dot_*  (generic scalar branch)
 * These are what the reference's llama_decode() (src/llama_server_context.cc:1635)
 * executes on an ngl=0 model.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 */
float oq_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1fu;
    const uint32_t man  = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: value = man * 2^-24 */
            float f = (float)man * 5.9604644775390625e-8f;
            memcpy(&bits, &f, 4);
            bits |= sign;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float out;
    memcpy(&out, &bits, 4);
    return out;
}

uint16_t oq_fp32_to_fp16(float f) { /* IEEE round-to-nearest-even, like F16C / ggml's table-free path */
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u | ((x >> 13) & 0x3ffu) : 0));
    }
    if (x >= 0x477ff000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (x < 0x38800000u) { /* subnormal half or zero */
        if (x < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (ties at exactly 2^-25 go to even = 0) */
        const int e = (int)(x >> 23);          /* biased exp, 102..112 */
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;             /* 14..24 */
        const uint32_t halfway = 1u << (shift - 1);
        const uint32_t rem = m & ((1u << shift) - 1);
        uint32_t r = m >> shift;
        if (rem > halfway || (rem == halfway && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    /* normal */
    uint32_t e = (x >> 23) - 112u;
    uint32_t m = x & 0x7fffffu;
    uint32_t r = (e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

/* ------------------------------------------------------------ block structs */
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; } blk_q4_0;                 /* 18 B / 32 */
typedef struct { uint16_t d; int8_t qs[32]; } blk_q8_0;                  /* 34 B / 32 */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; } blk_q4_K;           /* 144 B */
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } blk_q5_K; /* 176 B */
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; } blk_q6_K;  /* 210 B */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; } blk_q8_K;                      /* 292 B */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; } blk_q5_0;                        /* 22 B / 32 */
typedef struct { uint8_t scales[16]; uint8_t qs[64]; uint16_t d, dmin; } blk_q2_K;             /* 84 B */
typedef struct { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; } blk_q3_K; /* 110 B */
#pragma pack(pop)

int oq_block_elems(int type) {
    switch (type) {
        case OQ_TYPE_F32: case OQ_TYPE_F16: return 1;
        case OQ_TYPE_Q4_0: case OQ_TYPE_Q8_0: case OQ_TYPE_Q5_0: case OQ_TYPE_IQ4_NL: return 32;
        case OQ_TYPE_Q4_K: case OQ_TYPE_Q5_K: case OQ_TYPE_Q6_K: case OQ_TYPE_Q8_K: case OQ_TYPE_Q2_K: case OQ_TYPE_Q3_K: return 256;
    }
    return 0;
}
size_t oq_block_bytes(int type) {
    switch (type) {
        case OQ_TYPE_F32: return 4;
        case OQ_TYPE_F16: return 2;
        case OQ_TYPE_Q4_0: case OQ_TYPE_IQ4_NL: return sizeof(blk_q4_0);   /* IQ4_NL: the same 18 bytes, the nibbles index a code book */
        case OQ_TYPE_Q8_0: return sizeof(blk_q8_0);
        case OQ_TYPE_Q4_K: return sizeof(blk_q4_K);
        case OQ_TYPE_Q5_K: return sizeof(blk_q5_K);
        case OQ_TYPE_Q6_K: return sizeof(blk_q6_K);
        case OQ_TYPE_Q8_K: return sizeof(blk_q8_K);
        case OQ_TYPE_Q5_0: return sizeof(blk_q5_0);
        case OQ_TYPE_Q2_K: return sizeof(blk_q2_K);
        case OQ_TYPE_Q3_K: return sizeof(blk_q3_K);
    }
    return 0;
}
size_t oq_row_bytes(int type, int64_t n) {
    const int be = oq_block_elems(type);
    if (be <= 0) return 0;
    return (size_t)(n / be) * oq_block_bytes(type);
}

/* 6-bit scale/min pair j (0..7) of a q4_K / q5_K super-block (upstream: get_scale_min_k4) */
static void k4_scale_min(int j, const uint8_t *p, uint8_t *sc, uint8_t *mn) {
    if (j < 4) {
        *sc = p[j] & 63;
        *mn = p[j + 4] & 63;
    } else {
        *sc = (uint8_t)((p[j + 4] & 0x0f) | ((p[j - 4] >> 6) << 4));
        *mn = (uint8_t)((p[j + 4] >> 4) | ((p[j] >> 6) << 4));
    }
}

/* ------------------------------------------------------------ dequantise */
static void deq_q4_0(const blk_q4_0 *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        for (int j = 0; j < 16; j++) {
            y[i * 32 + j]      = (float)((b[i].qs[j] & 0x0f) - 8) * d;
            y[i * 32 + j + 16] = (float)((b[i].qs[j] >> 4) - 8) * d;
        }
    }
}
static void deq_q8_0(const blk_q8_0 *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        for (int j = 0; j < 32; j++) y[i * 32 + j] = (float)b[i].qs[j] * d;
    }
}
static void deq_q4_K(const blk_q4_K *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d), dm = oq_fp16_to_fp32(b[i].dmin);
        const uint8_t *q = b[i].qs;
        for (int c = 0; c < 4; c++) { /* 64 weights per chunk: low nibbles then high nibbles */
            uint8_t s0, m0, s1, m1;
            k4_scale_min(2 * c, b[i].scales, &s0, &m0);
            k4_scale_min(2 * c + 1, b[i].scales, &s1, &m1);
            const float d0 = d * s0, n0 = dm * m0, d1 = d * s1, n1 = dm * m1;
            for (int l = 0; l < 32; l++) y[l]      = d0 * (float)(q[l] & 0x0f) - n0;
            for (int l = 0; l < 32; l++) y[32 + l] = d1 * (float)(q[l] >> 4) - n1;
            q += 32;
            y += 64;
        }
    }
}
static void deq_q5_K(const blk_q5_K *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d), dm = oq_fp16_to_fp32(b[i].dmin);
        const uint8_t *q = b[i].qs, *h = b[i].qh;
        for (int c = 0; c < 4; c++) {
            uint8_t s0, m0, s1, m1;
            k4_scale_min(2 * c, b[i].scales, &s0, &m0);
            k4_scale_min(2 * c + 1, b[i].scales, &s1, &m1);
            const float d0 = d * s0, n0 = dm * m0, d1 = d * s1, n1 = dm * m1;
            const uint8_t u0 = (uint8_t)(1u << (2 * c)), u1 = (uint8_t)(2u << (2 * c));
            for (int l = 0; l < 32; l++) y[l]      = d0 * (float)((q[l] & 0x0f) + ((h[l] & u0) ? 16 : 0)) - n0;
            for (int l = 0; l < 32; l++) y[32 + l] = d1 * (float)((q[l] >> 4) + ((h[l] & u1) ? 16 : 0)) - n1;
            q += 32;
            y += 64;
        }
    }
}
static void deq_q6_K(const blk_q6_K *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        const uint8_t *ql = b[i].ql, *qh = b[i].qh;
        const int8_t *sc = b[i].scales;
        for (int n = 0; n < 2; n++) { /* two halves of 128 */
            for (int l = 0; l < 32; l++) {
                const int is = l / 16;
                const int q1 = (int)((ql[l] & 0x0f) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int q2 = (int)((ql[l + 32] & 0x0f) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l]      = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}
/* upstream: kvalues_iq4nl (ggml-common.h) and dequantize_row_iq4_nl: y = d * level[nibble]; low nibbles are elements 0..15, high ones 16..31 */
static const int8_t iq4nl_levels[16] = {-127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113};
static void deq_iq4_nl(const blk_q4_0 *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        for (int j = 0; j < 16; j++) {
            y[i * 32 + j]      = d * (float)iq4nl_levels[b[i].qs[j] & 0x0f];
            y[i * 32 + j + 16] = d * (float)iq4nl_levels[b[i].qs[j] >> 4];
        }
    }
}
/* upstream: dequantize_row_q5_0 - 16 low nibbles then 16 high nibbles, bit j / j + 16 of qh is the fifth bit, offset 16 */
static void deq_q5_0(const blk_q5_0 *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        uint32_t qh; memcpy(&qh, b[i].qh, 4);
        for (int j = 0; j < 16; j++) {
            const uint8_t xh0 = (uint8_t)(((qh >> j) << 4) & 0x10), xh1 = (uint8_t)((qh >> (j + 12)) & 0x10);
            y[i * 32 + j]      = (float)((int)((b[i].qs[j] & 0x0f) | xh0) - 16) * d;
            y[i * 32 + j + 16] = (float)((int)((b[i].qs[j] >> 4) | xh1) - 16) * d;
        }
    }
}
/* upstream: dequantize_row_q2_K - 16 sub-blocks of 16; scales[is]: low nibble scale, high nibble min; 2-bit codes, four per byte: byte l of a 32-byte
 * half holds elements l, l + 32, l + 64, l + 96 of that half */
static void deq_q2_K(const blk_q2_K *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d), mn = oq_fp16_to_fp32(b[i].dmin);
        const uint8_t *q = b[i].qs;
        int is = 0;
        for (int n = 0; n < 256; n += 128) {
            int shift = 0;
            for (int j = 0; j < 4; j++) {
                uint8_t sc = b[i].scales[is++];
                float dl = d * (float)(sc & 0xf), ml = mn * (float)(sc >> 4);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l] >> shift) & 3)) - ml;
                sc = b[i].scales[is++];
                dl = d * (float)(sc & 0xf); ml = mn * (float)(sc >> 4);
                for (int l = 0; l < 16; l++) *y++ = dl * (float)((int8_t)((q[l + 16] >> shift) & 3)) - ml;
                shift += 2;
            }
            q += 32;
        }
    }
}
/* the sixteen 6-bit scales of a q3_K super-block (upstream: the kmask1 / kmask2 shuffle of dequantize_row_q3_K / ggml_vec_dot_q3_K_q8_K) */
static void q3_scales(const uint8_t *s12, int8_t *sc16) {
    uint32_t aux[4];
    memcpy(aux, s12, 12);
    const uint32_t kmask1 = 0x03030303u, kmask2 = 0x0f0f0f0fu, tmp = aux[2];
    aux[2] = ((aux[0] >> 4) & kmask2) | (((tmp >> 4) & kmask1) << 4);
    aux[3] = ((aux[1] >> 4) & kmask2) | (((tmp >> 6) & kmask1) << 4);
    aux[0] = (aux[0] & kmask2) | (((tmp >> 0) & kmask1) << 4);
    aux[1] = (aux[1] & kmask2) | (((tmp >> 2) & kmask1) << 4);
    memcpy(sc16, aux, 16);
}
/* signed 3-bit weights of a q3_K super-block: low two bits from qs, minus 4 where the hmask bit is CLEAR */
static void unpack_q3_K(const blk_q3_K *b, int8_t *w) {
    const uint8_t *q = b->qs, *hm = b->hmask;
    uint8_t m = 1;
    for (int n = 0; n < 256; n += 128) {
        for (int j = 0; j < 4; j++) {
            for (int l = 0; l < 32; l++) w[l] = (int8_t)((int8_t)((q[l] >> (2 * j)) & 3) - ((hm[l] & m) ? 0 : 4));
            w += 32; m <<= 1;
        }
        q += 32;
    }
}
static void deq_q3_K(const blk_q3_K *b, float *y, int64_t nb) {
    int8_t w[256], sc[16];
    for (int64_t i = 0; i < nb; i++) {
        const float d = oq_fp16_to_fp32(b[i].d);
        unpack_q3_K(&b[i], w);
        q3_scales(b[i].scales, sc);
        for (int g = 0; g < 16; g++) {
            const float dl = d * (float)(sc[g] - 32);
            for (int l = 0; l < 16; l++) y[i * 256 + g * 16 + l] = dl * (float)w[g * 16 + l];
        }
    }
}
static void deq_q8_K(const blk_q8_K *b, float *y, int64_t nb) {
    for (int64_t i = 0; i < nb; i++)
        for (int j = 0; j < 256; j++) y[i * 256 + j] = b[i].d * b[i].qs[j];
}

void oq_dequantize_row(int type, const void *src, float *dst, int64_t n) {
    switch (type) {
        case OQ_TYPE_F32: memcpy(dst, src, (size_t)n * 4); break;
        case OQ_TYPE_F16: for (int64_t i = 0; i < n; i++) dst[i] = oq_fp16_to_fp32(((const uint16_t *)src)[i]); break;
        case OQ_TYPE_Q4_0: deq_q4_0((const blk_q4_0 *)src, dst, n / 32); break;
        case OQ_TYPE_Q8_0: deq_q8_0((const blk_q8_0 *)src, dst, n / 32); break;
        case OQ_TYPE_Q4_K: deq_q4_K((const blk_q4_K *)src, dst, n / 256); break;
        case OQ_TYPE_Q5_K: deq_q5_K((const blk_q5_K *)src, dst, n / 256); break;
        case OQ_TYPE_Q6_K: deq_q6_K((const blk_q6_K *)src, dst, n / 256); break;
        case OQ_TYPE_Q8_K: deq_q8_K((const blk_q8_K *)src, dst, n / 256); break;
        case OQ_TYPE_Q5_0: deq_q5_0((const blk_q5_0 *)src, dst, n / 32); break;
        case OQ_TYPE_IQ4_NL: deq_iq4_nl((const blk_q4_0 *)src, dst, n / 32); break;
        case OQ_TYPE_Q2_K: deq_q2_K((const blk_q2_K *)src, dst, n / 256); break;
        case OQ_TYPE_Q3_K: deq_q3_K((const blk_q3_K *)src, dst, n / 256); break;
        default: abort();
    }
}

/* ------------------------------------------------------------ quantise */
/* round-to-nearest-even via the 1.5*2^23 magic add (upstream: nearest_int) */
static int nearest_int(float v) {
    float t = v + 12582912.f;
    int i;
    memcpy(&i, &t, 4);
    return (i & 0x007fffff) - 0x00400000;
}

void oq_quantize_row_q8_0(const float *x, void *dst, int64_t n) {
    blk_q8_0 *y = (blk_q8_0 *)dst;
    for (int64_t i = 0; i < n / 32; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float a = fabsf(x[i * 32 + j]); if (a > amax) amax = a; }
        const float d = amax / 127.0f;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oq_fp32_to_fp16(d);
        for (int j = 0; j < 32; j++) y[i].qs[j] = (int8_t)roundf(x[i * 32 + j] * id);
    }
}

void oq_quantize_row_q4_0(const float *x, void *dst, int64_t n) {
    blk_q4_0 *y = (blk_q4_0 *)dst;
    for (int64_t i = 0; i < n / 32; i++) {
        float amax = 0.0f, vmax = 0.0f; /* signed value of the abs-max element */
        for (int j = 0; j < 32; j++) {
            const float v = x[i * 32 + j];
            if (amax < fabsf(v)) { amax = fabsf(v); vmax = v; }
        }
        const float d = vmax / -8.0f;
        const float id = d ? 1.0f / d : 0.0f;
        y[i].d = oq_fp32_to_fp16(d);
        for (int j = 0; j < 16; j++) {
            const float x0 = x[i * 32 + j] * id, x1 = x[i * 32 + 16 + j] * id;
            int a = (int)(int8_t)(x0 + 8.5f); if (a > 15) a = 15;
            int b = (int)(int8_t)(x1 + 8.5f); if (b > 15) b = 15;
            y[i].qs[j] = (uint8_t)(a | (b << 4));
        }
    }
}

void oq_quantize_row_q8_K(const float *x, void *dst, int64_t n) {
    blk_q8_K *y = (blk_q8_K *)dst;
    for (int64_t i = 0; i < n / 256; i++) {
        float vmax = 0.0f, amax = 0.0f;
        for (int j = 0; j < 256; j++) {
            const float a = fabsf(x[j]);
            if (a > amax) { amax = a; vmax = x[j]; }
        }
        if (amax == 0.0f) {
            memset(&y[i], 0, sizeof(blk_q8_K));
            x += 256;
            continue;
        }
        const float iscale = -127.0f / vmax;
        for (int j = 0; j < 256; j++) {
            int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t)(v > 127 ? 127 : v);
        }
        for (int j = 0; j < 16; j++) {
            int s = 0;
            for (int k = 0; k < 16; k++) s += y[i].qs[j * 16 + k];
            y[i].bsums[j] = (int16_t)s;
        }
        y[i].d = 1.0f / iscale;
        x += 256;
    }
}

void oq_quantize_row(int type, const float *x, void *dst, int64_t n) {
    switch (type) {
        case OQ_TYPE_F32: memcpy(dst, x, (size_t)n * 4); break;
        case OQ_TYPE_F16: for (int64_t i = 0; i < n; i++) ((uint16_t *)dst)[i] = oq_fp32_to_fp16(x[i]); break;
        case OQ_TYPE_Q8_0: oq_quantize_row_q8_0(x, dst, n); break;
        case OQ_TYPE_Q4_0: oq_quantize_row_q4_0(x, dst, n); break;
        case OQ_TYPE_Q8_K: oq_quantize_row_q8_K(x, dst, n); break;
        default: abort();
    }
}

/* ------------------------------------------------------------ dot products */
int oq_vec_dot_type(int type) {
    switch (type) {
        case OQ_TYPE_F32: return OQ_TYPE_F32;
        case OQ_TYPE_F16: return OQ_TYPE_F16;
        case OQ_TYPE_Q4_0: case OQ_TYPE_Q8_0: case OQ_TYPE_Q5_0: case OQ_TYPE_IQ4_NL: return OQ_TYPE_Q8_0;
        case OQ_TYPE_Q4_K: case OQ_TYPE_Q5_K: case OQ_TYPE_Q6_K: case OQ_TYPE_Q2_K: case OQ_TYPE_Q3_K: return OQ_TYPE_Q8_K;
    }
    return -1;
}

/* Eight float lanes per row, as the generic scalar branch keeps them (lane = element index mod 8);
 * x86 SIMD builds of ggml differ from this only in float association. */
typedef struct { float lane[8]; float tail; } acc8;

/* Calibration knobs (tests only; default 0 = the restated reference behaviour):
 *  assoc variant 1 re-associates the f32 sums of the K-quant dots (block integer total first, one f32
 *  accumulator) -- an equally valid order, used to measure how much logits move under re-association. */
static int g_assoc_variant = 0;
void oq_set_assoc_variant(int v) { g_assoc_variant = v; }

static float acc8_finish(const acc8 *a) {
    float s = a->tail;
    for (int l = 0; l < 8; l++) s += a->lane[l];
    return s;
}

/* unpack a q4_K/q5_K/q6_K super-block into signed 8-bit weights w[256] */
static void unpack_q4_K(const blk_q4_K *b, int8_t *w) {
    const uint8_t *q = b->qs;
    for (int c = 0; c < 4; c++) {
        for (int l = 0; l < 32; l++) w[l] = (int8_t)(q[l] & 0x0f);
        for (int l = 0; l < 32; l++) w[32 + l] = (int8_t)(q[l] >> 4);
        w += 64; q += 32;
    }
}
static void unpack_q5_K(const blk_q5_K *b, int8_t *w) {
    const uint8_t *q = b->qs, *h = b->qh;
    uint8_t m = 1;
    for (int c = 0; c < 4; c++) {
        for (int l = 0; l < 32; l++) w[l] = (int8_t)((q[l] & 0x0f) + ((h[l] & m) ? 16 : 0));
        m <<= 1;
        for (int l = 0; l < 32; l++) w[32 + l] = (int8_t)((q[l] >> 4) + ((h[l] & m) ? 16 : 0));
        m <<= 1;
        w += 64; q += 32;
    }
}
static void unpack_q6_K(const blk_q6_K *b, int8_t *w) {
    const uint8_t *ql = b->ql, *qh = b->qh;
    for (int n = 0; n < 2; n++) {
        for (int l = 0; l < 32; l++) {
            w[l]      = (int8_t)(((ql[l] & 0x0f) | (((qh[l] >> 0) & 3) << 4)) - 32);
            w[l + 32] = (int8_t)(((ql[l + 32] & 0x0f) | (((qh[l] >> 2) & 3) << 4)) - 32);
            w[l + 64] = (int8_t)(((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32);
            w[l + 96] = (int8_t)(((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
        }
        w += 128; ql += 64; qh += 32;
    }
}

/* One super-block of a 4/5-bit K-quant against q8_K:
 * lanes[l] = sum over sub-blocks j of sc_j * sum_{i = l mod 8} w_i*q8_i ; msum = sum_j m_j * (bsums pair) */
static void k45_block(const int8_t *w, const uint8_t *scales12, const blk_q8_K *y, int32_t lanes[8], int32_t *msum) {
    uint8_t sc[8], mn[8];
    for (int j = 0; j < 8; j++) k4_scale_min(j, scales12, &sc[j], &mn[j]);
    int32_t ms = 0;
    for (int j = 0; j < 16; j++) ms += (int32_t)y->bsums[j] * mn[j / 2];
    *msum = ms;
    memset(lanes, 0, 8 * sizeof(int32_t));
    for (int j = 0; j < 8; j++)
        for (int i = 0; i < 32; i++)
            lanes[i & 7] += (int32_t)sc[j] * ((int32_t)y->qs[j * 32 + i] * w[j * 32 + i]);
}
static void q6_block(const int8_t *w, const int8_t *sc16, const blk_q8_K *y, int32_t lanes[8]) {
    memset(lanes, 0, 8 * sizeof(int32_t));
    for (int j = 0; j < 16; j++)
        for (int i = 0; i < 16; i++)
            lanes[i & 7] += (int32_t)sc16[j] * ((int32_t)y->qs[j * 16 + i] * w[j * 16 + i]);
}

static float dot_q4_K(int64_t n, const blk_q4_K *x, const blk_q8_K *y) {
    acc8 a; memset(&a, 0, sizeof a);
    int8_t w[256]; int32_t lanes[8], ms;
    for (int64_t i = 0; i < n / 256; i++) {
        unpack_q4_K(&x[i], w);
        k45_block(w, x[i].scales, &y[i], lanes, &ms);
        const float d = oq_fp16_to_fp32(x[i].d) * y[i].d;
        const float dm = oq_fp16_to_fp32(x[i].dmin) * y[i].d;
        if (g_assoc_variant == 1) {
            int32_t tot = 0;
            for (int l = 0; l < 8; l++) tot += lanes[l];
            a.tail += d * (float)tot - dm * (float)ms;
            continue;
        }
        for (int l = 0; l < 8; l++) a.lane[l] += d * (float)lanes[l];
        a.tail -= dm * (float)ms;
    }
    return acc8_finish(&a);
}
static float dot_q5_K(int64_t n, const blk_q5_K *x, const blk_q8_K *y) {
    acc8 a; memset(&a, 0, sizeof a);
    int8_t w[256]; int32_t lanes[8], ms;
    for (int64_t i = 0; i < n / 256; i++) {
        unpack_q5_K(&x[i], w);
        k45_block(w, x[i].scales, &y[i], lanes, &ms);
        const float d = oq_fp16_to_fp32(x[i].d) * y[i].d;
        const float dm = oq_fp16_to_fp32(x[i].dmin) * y[i].d;
        if (g_assoc_variant == 1) {
            int32_t tot = 0;
            for (int l = 0; l < 8; l++) tot += lanes[l];
            a.tail += d * (float)tot - dm * (float)ms;
            continue;
        }
        for (int l = 0; l < 8; l++) a.lane[l] += d * (float)lanes[l];
        a.tail -= dm * (float)ms;
    }
    return acc8_finish(&a);
}
static float dot_q6_K(int64_t n, const blk_q6_K *x, const blk_q8_K *y) {
    acc8 a; memset(&a, 0, sizeof a);
    int8_t w[256]; int32_t lanes[8];
    for (int64_t i = 0; i < n / 256; i++) {
        unpack_q6_K(&x[i], w);
        q6_block(w, x[i].scales, &y[i], lanes);
        const float d = oq_fp16_to_fp32(x[i].d) * y[i].d;
        if (g_assoc_variant == 1) {
            int32_t tot = 0;
            for (int l = 0; l < 8; l++) tot += lanes[l];
            a.tail += d * (float)tot;
            continue;
        }
        for (int l = 0; l < 8; l++) a.lane[l] += d * (float)lanes[l];
    }
    return acc8_finish(&a);
}
static float dot_q8_0(int64_t n, const blk_q8_0 *x, const blk_q8_0 *y) {
    float s = 0.0f;
    for (int64_t i = 0; i < n / 32; i++) {
        int32_t si = 0;
        for (int j = 0; j < 32; j++) si += (int32_t)x[i].qs[j] * y[i].qs[j];
        s += (float)si * (oq_fp16_to_fp32(x[i].d) * oq_fp16_to_fp32(y[i].d));
    }
    return s;
}
static float dot_q4_0(int64_t n, const blk_q4_0 *x, const blk_q8_0 *y) {
    float s = 0.0f;
    for (int64_t i = 0; i < n / 32; i++) {
        int32_t si = 0;
        for (int j = 0; j < 16; j++) {
            si += ((int)(x[i].qs[j] & 0x0f) - 8) * y[i].qs[j];
            si += ((int)(x[i].qs[j] >> 4) - 8) * y[i].qs[j + 16];
        }
        s += (float)si * oq_fp16_to_fp32(x[i].d) * oq_fp16_to_fp32(y[i].d);
    }
    return s;
}
static int32_t q4_0_block(const blk_q4_0 *x, const blk_q8_0 *y) {
    int32_t si = 0;
    for (int j = 0; j < 16; j++) {
        si += ((int)(x->qs[j] & 0x0f) - 8) * y->qs[j];
        si += ((int)(x->qs[j] >> 4) - 8) * y->qs[j + 16];
    }
    return si;
}
/* upstream scalar ggml_vec_dot_iq4_nl_q8_0: d = d_y * d_x; sumi1 over the low nibbles, sumi2 over the high ones; sumf += d * (sumi1 + sumi2) */
static int32_t iq4_nl_block(const blk_q4_0 *x, const blk_q8_0 *y) {
    int32_t s1 = 0, s2 = 0;
    for (int j = 0; j < 16; j++) {
        s1 += y->qs[j] * iq4nl_levels[x->qs[j] & 0x0f];
        s2 += y->qs[j + 16] * iq4nl_levels[x->qs[j] >> 4];
    }
    return s1 + s2;
}
static float dot_iq4_nl(int64_t n, const blk_q4_0 *x, const blk_q8_0 *y) {
    float s = 0.0f;
    for (int64_t i = 0; i < n / 32; i++) {
        const float d = oq_fp16_to_fp32(y[i].d) * oq_fp16_to_fp32(x[i].d);
        s += d * (float)iq4_nl_block(&x[i], &y[i]);
    }
    return s;
}
/* upstream scalar ggml_vec_dot_q5_0_q8_0: one integer sum per block, sumf += (d_x * d_y) * sumi */
static int32_t q5_0_block(const blk_q5_0 *x, const blk_q8_0 *y) {
    uint32_t qh; memcpy(&qh, x->qh, 4);
    int32_t si = 0;
    for (int j = 0; j < 16; j++) {
        const uint8_t xh0 = (uint8_t)(((qh & (1u << (j + 0))) >> (j + 0)) << 4), xh1 = (uint8_t)((qh & (1u << (j + 16))) >> (j + 12));
        const int32_t x0 = (int8_t)(((x->qs[j] & 0x0f) | xh0) - 16), x1 = (int8_t)(((x->qs[j] >> 4) | xh1) - 16);
        si += x0 * y->qs[j] + x1 * y->qs[j + 16];
    }
    return si;
}
static float dot_q5_0(int64_t n, const blk_q5_0 *x, const blk_q8_0 *y) {
    float s = 0.0f;
    for (int64_t i = 0; i < n / 32; i++)
        s += (oq_fp16_to_fp32(x[i].d) * oq_fp16_to_fp32(y[i].d)) * (float)q5_0_block(&x[i], &y[i]);
    return s;
}
/* upstream scalar ggml_vec_dot_q2_K_q8_K: per super-block isum = sum over the 16 sub-blocks of (scale nibble) * sum16(q2 * q8), summs = sum of
 * (min nibble) * bsums; ONE float accumulator: sumf += dall * isum - dmin * summs */
static void q2_block(const blk_q2_K *x, const blk_q8_K *y, int32_t *isum, int32_t *summs) {
    int32_t ms = 0;
    for (int j = 0; j < 16; j++) ms += (int32_t)y->bsums[j] * (x->scales[j] >> 4);
    const uint8_t *q2 = x->qs; const int8_t *q8 = y->qs;
    int32_t is_ = 0; int is = 0;
    for (int k = 0; k < 2; k++) {
        int shift = 0;
        for (int j = 0; j < 4; j++) {
            int32_t d = x->scales[is++] & 0xf, l16 = 0;
            for (int l = 0; l < 16; l++) l16 += q8[l] * ((q2[l] >> shift) & 3);
            is_ += d * l16;
            d = x->scales[is++] & 0xf; l16 = 0;
            for (int l = 16; l < 32; l++) l16 += q8[l] * ((q2[l] >> shift) & 3);
            is_ += d * l16;
            shift += 2; q8 += 32;
        }
        q2 += 32;
    }
    *isum = is_; *summs = ms;
}
static float dot_q2_K(int64_t n, const blk_q2_K *x, const blk_q8_K *y) {
    float s = 0.0f;
    for (int64_t i = 0; i < n / 256; i++) {
        int32_t isum, ms;
        q2_block(&x[i], &y[i], &isum, &ms);
        const float dall = y[i].d * oq_fp16_to_fp32(x[i].d), dmin = y[i].d * oq_fp16_to_fp32(x[i].dmin);
        s += dall * (float)isum - dmin * (float)ms;
    }
    return s;
}
/* upstream scalar ggml_vec_dot_q3_K_q8_K: eight integer lanes per super-block (lane = element index mod 8), each product scaled by (scale - 32) of its
 * 16-element sub-block, then sums[l] += d * aux32[l]; the eight float lanes are added at the end */
static void q3_block(const int8_t *w, const int8_t *sc16, const blk_q8_K *y, int32_t lanes[8]) {
    memset(lanes, 0, 8 * sizeof(int32_t));
    for (int j = 0; j < 16; j++)
        for (int i = 0; i < 16; i++)
            lanes[i & 7] += (int32_t)(sc16[j] - 32) * (int32_t)(int16_t)((int32_t)y->qs[j * 16 + i] * w[j * 16 + i]);
}
static float dot_q3_K(int64_t n, const blk_q3_K *x, const blk_q8_K *y) {
    acc8 a; memset(&a, 0, sizeof a);
    int8_t w[256], sc[16]; int32_t lanes[8];
    for (int64_t i = 0; i < n / 256; i++) {
        unpack_q3_K(&x[i], w);
        q3_scales(x[i].scales, sc);
        q3_block(w, sc, &y[i], lanes);
        const float d = oq_fp16_to_fp32(x[i].d) * y[i].d;
        if (g_assoc_variant == 1) {
            int32_t tot = 0;
            for (int l = 0; l < 8; l++) tot += lanes[l];
            a.tail += d * (float)tot;
            continue;
        }
        for (int l = 0; l < 8; l++) a.lane[l] += d * (float)lanes[l];
    }
    return acc8_finish(&a);
}
static float dot_f16(int64_t n, const uint16_t *x, const uint16_t *y) {
    double s = 0.0; /* upstream scalar: ggml_float accumulator */
    for (int64_t i = 0; i < n; i++) s += (double)(oq_fp16_to_fp32(x[i]) * oq_fp16_to_fp32(y[i]));
    return (float)s;
}
static float dot_f32(int64_t n, const float *x, const float *y) {
    double s = 0.0;
    for (int64_t i = 0; i < n; i++) s += (double)(x[i] * y[i]);
    return (float)s;
}

float oq_vec_dot(int type, int64_t n, const void *w, const void *a) {
    switch (type) {
        case OQ_TYPE_F32: return dot_f32(n, (const float *)w, (const float *)a);
        case OQ_TYPE_F16: return dot_f16(n, (const uint16_t *)w, (const uint16_t *)a);
        case OQ_TYPE_Q4_0: return dot_q4_0(n, (const blk_q4_0 *)w, (const blk_q8_0 *)a);
        case OQ_TYPE_Q8_0: return dot_q8_0(n, (const blk_q8_0 *)w, (const blk_q8_0 *)a);
        case OQ_TYPE_Q4_K: return dot_q4_K(n, (const blk_q4_K *)w, (const blk_q8_K *)a);
        case OQ_TYPE_Q5_K: return dot_q5_K(n, (const blk_q5_K *)w, (const blk_q8_K *)a);
        case OQ_TYPE_Q6_K: return dot_q6_K(n, (const blk_q6_K *)w, (const blk_q8_K *)a);
        case OQ_TYPE_Q5_0: return dot_q5_0(n, (const blk_q5_0 *)w, (const blk_q8_0 *)a);
        case OQ_TYPE_IQ4_NL: return dot_iq4_nl(n, (const blk_q4_0 *)w, (const blk_q8_0 *)a);
        case OQ_TYPE_Q2_K: return dot_q2_K(n, (const blk_q2_K *)w, (const blk_q8_K *)a);
        case OQ_TYPE_Q3_K: return dot_q3_K(n, (const blk_q3_K *)w, (const blk_q8_K *)a);
    }
    abort();
}

void oq_vec_dot_int_partials(int type, int64_t n, const void *wrow, const void *act,
                             int32_t *isum, int32_t *msum) {
    int8_t w[256]; int32_t lanes[8], ms;
    if (type == OQ_TYPE_Q8_0) {
        const blk_q8_0 *x = (const blk_q8_0 *)wrow, *y = (const blk_q8_0 *)act;
        for (int64_t i = 0; i < n / 32; i++) {
            int32_t si = 0;
            for (int j = 0; j < 32; j++) si += (int32_t)x[i].qs[j] * y[i].qs[j];
            isum[i] = si; msum[i] = 0;
        }
        return;
    }
    if (type == OQ_TYPE_Q4_0 || type == OQ_TYPE_IQ4_NL) {
        const blk_q4_0 *x = (const blk_q4_0 *)wrow; const blk_q8_0 *y = (const blk_q8_0 *)act;
        for (int64_t i = 0; i < n / 32; i++) { isum[i] = type == OQ_TYPE_Q4_0 ? q4_0_block(&x[i], &y[i]) : iq4_nl_block(&x[i], &y[i]); msum[i] = 0; }
        return;
    }
    if (type == OQ_TYPE_Q5_0) {
        const blk_q5_0 *x = (const blk_q5_0 *)wrow; const blk_q8_0 *y = (const blk_q8_0 *)act;
        for (int64_t i = 0; i < n / 32; i++) { isum[i] = q5_0_block(&x[i], &y[i]); msum[i] = 0; }
        return;
    }
    if (type == OQ_TYPE_Q2_K) {
        const blk_q2_K *x = (const blk_q2_K *)wrow; const blk_q8_K *y2 = (const blk_q8_K *)act;
        for (int64_t i = 0; i < n / 256; i++) q2_block(&x[i], &y2[i], &isum[i], &msum[i]);
        return;
    }
    if (type == OQ_TYPE_Q3_K) {
        const blk_q3_K *x = (const blk_q3_K *)wrow; const blk_q8_K *y3 = (const blk_q8_K *)act;
        int8_t sc[16];
        for (int64_t i = 0; i < n / 256; i++) {
            unpack_q3_K(&x[i], w); q3_scales(x[i].scales, sc); q3_block(w, sc, &y3[i], lanes);
            int32_t t = 0;
            for (int l = 0; l < 8; l++) t += lanes[l];
            isum[i] = t; msum[i] = 0;
        }
        return;
    }
    const blk_q8_K *y = (const blk_q8_K *)act;
    for (int64_t i = 0; i < n / 256; i++) {
        ms = 0;
        if (type == OQ_TYPE_Q4_K) {
            const blk_q4_K *x = (const blk_q4_K *)wrow;
            unpack_q4_K(&x[i], w); k45_block(w, x[i].scales, &y[i], lanes, &ms);
        } else if (type == OQ_TYPE_Q5_K) {
            const blk_q5_K *x = (const blk_q5_K *)wrow;
            unpack_q5_K(&x[i], w); k45_block(w, x[i].scales, &y[i], lanes, &ms);
        } else if (type == OQ_TYPE_Q6_K) {
            const blk_q6_K *x = (const blk_q6_K *)wrow;
            unpack_q6_K(&x[i], w); q6_block(w, x[i].scales, &y[i], lanes);
        } else {
            abort();
        }
        int32_t s = 0;
        for (int l = 0; l < 8; l++) s += lanes[l];
        isum[i] = s; msum[i] = ms;
    }
}
