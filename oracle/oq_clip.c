/*
 * oracle/oq_clip.c — CPU restatement of the LLaVA image path: clip_image_preprocess + clip_image_encode of a LLaVA-1.5 style projector file
 * (general.architecture "clip": CLIP ViT tower cut after its second-to-last block, two-layer MLP projector), and the LLaVA-1.6 image grid on top of it.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: the code restated here is llama.cpp's examples/llava/clip.cpp (submodule `llama.cpp`, an empty
 * directory in /root/reference); the reference reaches it through clip_model_load (/root/reference/src/llama_server_context.cc:187),
 * clip_image_load_from_bytes (:568) and llava_image_embed_make_with_clip_img (:820), and holds no vector for it.  Restated from the published graph:
 *   preprocess (LLaVA-1.5: no image grid): pad to a square with the mean colour (122, 116, 104), top-left aligned; bilinear resample to image_size with
 *     the half-pixel mapping sx = (x + 0.5) * scale - 0.5; round to a byte; (v / 255 - mean[c]) / std[c]; planar [3][S][S].
 *   preprocess (LLaVA-1.6: clip.vision.image_grid_pinpoints present): image 0 = the whole picture resized to S x S with clip.cpp's bicubic_resize (source index
 *     truncated, Catmull-Rom style cubic, separable, edges clamped, rounded to a byte); with clip.vision.mm_patch_merge_type "spatial_unpad" followed by the
 *     (without that merge type a non-square picture takes the LLaVA-1.5 branch - pad_to_square - and only a square one reaches the overview)
 *     S x S tiles, row-major, of the picture fitted to the best canvas (select_best_resolution: most kept pixels, then least waste; resize_and_pad_image:
 *     aspect-preserving bicubic resize, centred on black); every image normalised (v / 255 - mean) / std without further resampling.
 *   embed (llava.cpp encode_image_with_clip + clip_llava_handle_patches): encode every image; the overview's rows first, then the tiles' rows re-ordered from
 *     tile-by-tile to the canvas' row-major order ("without newline tokens": model.image_newline is not used).
 *   encode: patch embedding = 2-d convolution, stride = patch (im2col in f16 x f16 kernel); [class ; patches] + position embeddings; pre-LayerNorm; per block:
 *     LN1 -> Q (scaled by 1 / sqrt(d_head) after its bias), K, V with biases -> softmax(K^T Q) -> V -> output projection + bias -> residual ->
 *     LN2 -> ffn_down tensor (n_embd -> n_ff; the converter's names are swapped) + bias -> quick-GELU (or GELU: clip.use_gelu) -> ffn_up tensor + bias ->
 *     residual; the class row is dropped; mm.0 + bias -> GELU -> mm.2 + bias.
 *   GELU / quick-GELU are ggml's f16 lookup tables: y = f16(f(f16(x))) (GELU returns 0 below -10 and x above 10 before the table).
 *   mat-muls against f16 tensors convert their activation rows to f16 (oq_mul_mat); the attention products are f32 with double accumulators.
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

#define CLIP_MAX_LAYERS 64

typedef struct { char name[96]; int type, n_dims; int64_t ne[4]; const uint8_t *data; } ct;
typedef struct {
    const ct *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *ln1w, *ln1b, *ln2w, *ln2b, *ff_i_w, *ff_i_b, *ff_o_w, *ff_o_b;
} clayer;
struct oq_clip {
    int fd; uint8_t *map; size_t map_size;
    int n_tensors; ct *tensors;
    int image_size, patch_size, n_embd, n_ff, n_head, n_layer, proj_dim, use_gelu;
    float eps, mean[3], std[3];
    int n_pin, pin[128], spatial_unpad;           /* (width, height) pairs */
    const ct *class_embd, *patch_w, *pos_embd, *pre_ln_w, *pre_ln_b, *mm0w, *mm0b, *mm2w, *mm2b;
    clayer layers[CLIP_MAX_LAYERS];
};

typedef struct { const uint8_t *p, *end; int bad; } rd;
static uint64_t rd_u(rd *r, int n) {
    if (r->p + n > r->end) { r->bad = 1; return 0; }
    uint64_t v = 0;
    memcpy(&v, r->p, (size_t)n);
    r->p += n;
    return v;
}
static void rd_str(rd *r, char *dst, size_t cap) {
    const uint64_t l = rd_u(r, 8);
    if (r->bad || r->p + l > r->end) { r->bad = 1; if (cap) dst[0] = 0; return; }
    const size_t c = l < cap - 1 ? (size_t)l : cap - 1;
    memcpy(dst, r->p, c); dst[c] = 0;
    r->p += l;
}
static const int scalar_size[13] = {1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8};

static const ct *find(const oq_clip *c, const char *name) {
    for (int i = 0; i < c->n_tensors; i++) if (!strcmp(c->tensors[i].name, name)) return &c->tensors[i];
    return NULL;
}
static const ct *lt(const oq_clip *c, int il, const char *suf) {
    char nm[96];
    snprintf(nm, sizeof nm, "v.blk.%d.%s", il, suf);
    return find(c, nm);
}

void oq_clip_free(oq_clip *c) {
    if (!c) return;
    if (c->map) munmap(c->map, c->map_size);
    if (c->fd >= 0) close(c->fd);
    free(c->tensors);
    free(c);
}

oq_clip *oq_clip_load(const char *path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) return NULL;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return NULL; }
    uint8_t *map = (uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) { close(fd); return NULL; }
    oq_clip *c = (oq_clip *)calloc(1, sizeof *c);
    c->fd = fd; c->map = map; c->map_size = (size_t)st.st_size;
    c->eps = 1e-5f;
    rd r = {map, map + st.st_size, 0};
    const uint32_t magic = (uint32_t)rd_u(&r, 4), version = (uint32_t)rd_u(&r, 4);
    const uint64_t n_tensors = rd_u(&r, 8), n_kv = rd_u(&r, 8);
    if (magic != 0x46554747u || version < 2 || version > 3) { oq_clip_free(c); return NULL; }
    uint64_t alignment = 32;
    char key[160], sval[64];
    int is_clip = 0;
    for (uint64_t i = 0; i < n_kv && !r.bad; i++) {
        rd_str(&r, key, sizeof key);
        const int type = (int)rd_u(&r, 4);
        if (type == 8) {
            rd_str(&r, sval, sizeof sval);
            if (!strcmp(key, "general.architecture")) is_clip = !strcmp(sval, "clip");
            if (!strcmp(key, "clip.vision.mm_patch_merge_type")) c->spatial_unpad = !strcmp(sval, "spatial_unpad");
            continue;
        }
        if (type == 9) {
            const int et = (int)rd_u(&r, 4);
            const uint64_t n = rd_u(&r, 8);
            float *dst = !strcmp(key, "clip.vision.image_mean") ? c->mean : !strcmp(key, "clip.vision.image_std") ? c->std : NULL;
            const int is_pin = !strcmp(key, "clip.vision.image_grid_pinpoints") && (et == 4 || et == 5) && n <= 128 && n % 2 == 0;
            if (is_pin) c->n_pin = (int)n / 2;
            for (uint64_t j = 0; j < n && !r.bad; j++) {
                if (et == 8) { const uint64_t l = rd_u(&r, 8); if (r.p + l > r.end) r.bad = 1; else r.p += l; }
                else if (et >= 0 && et < 13 && scalar_size[et]) {
                    const uint64_t raw = rd_u(&r, scalar_size[et]);
                    if (dst && et == 6 && j < 3) { const uint32_t b = (uint32_t)raw; memcpy(&dst[j], &b, 4); }
                    if (is_pin) c->pin[j] = (int)(int32_t)(uint32_t)raw;
                } else r.bad = 1;
            }
            continue;
        }
        if (type < 0 || type > 12) { r.bad = 1; break; }
        const uint64_t raw = rd_u(&r, scalar_size[type]);
        float f = 0.0f;
        if (type == 6) { const uint32_t b = (uint32_t)raw; memcpy(&f, &b, 4); }
        if (!strcmp(key, "general.alignment")) alignment = raw;
        else if (!strcmp(key, "clip.vision.image_size")) c->image_size = (int)raw;
        else if (!strcmp(key, "clip.vision.patch_size")) c->patch_size = (int)raw;
        else if (!strcmp(key, "clip.vision.embedding_length")) c->n_embd = (int)raw;
        else if (!strcmp(key, "clip.vision.feed_forward_length")) c->n_ff = (int)raw;
        else if (!strcmp(key, "clip.vision.attention.head_count")) c->n_head = (int)raw;
        else if (!strcmp(key, "clip.vision.block_count")) c->n_layer = (int)raw;
        else if (!strcmp(key, "clip.vision.attention.layer_norm_epsilon")) c->eps = f;
        else if (!strcmp(key, "clip.use_gelu")) c->use_gelu = raw != 0;
    }
    c->n_tensors = (int)n_tensors;
    c->tensors = (ct *)calloc(n_tensors ? n_tensors : 1, sizeof(ct));
    uint64_t *offs = (uint64_t *)calloc(n_tensors ? n_tensors : 1, 8);
    for (uint64_t i = 0; i < n_tensors && !r.bad; i++) {
        ct *t = &c->tensors[i];
        rd_str(&r, t->name, sizeof t->name);
        t->n_dims = (int)rd_u(&r, 4);
        for (int d = 0; d < 4; d++) t->ne[d] = 1;
        for (int d = 0; d < t->n_dims && d < 4; d++) t->ne[d] = (int64_t)rd_u(&r, 8);
        t->type = (int)rd_u(&r, 4);
        offs[i] = rd_u(&r, 8);
    }
    if (r.bad || !is_clip || c->n_layer < 1 || c->n_layer > CLIP_MAX_LAYERS) { free(offs); oq_clip_free(c); return NULL; }
    uint64_t data_off = (uint64_t)(r.p - map);
    data_off = (data_off + alignment - 1) / alignment * alignment;
    for (uint64_t i = 0; i < n_tensors; i++) c->tensors[i].data = map + data_off + offs[i];
    free(offs);
    c->class_embd = find(c, "v.class_embd"); c->patch_w = find(c, "v.patch_embd.weight"); c->pos_embd = find(c, "v.position_embd.weight");
    c->pre_ln_w = find(c, "v.pre_ln.weight"); c->pre_ln_b = find(c, "v.pre_ln.bias");
    c->mm0w = find(c, "mm.0.weight"); c->mm0b = find(c, "mm.0.bias"); c->mm2w = find(c, "mm.2.weight"); c->mm2b = find(c, "mm.2.bias");
    int ok = c->class_embd && c->patch_w && c->pos_embd && c->pre_ln_w && c->pre_ln_b && c->mm0w && c->mm0b && c->mm2w && c->mm2b && c->n_head > 0 &&
             c->image_size > 0 && c->patch_size > 0 && c->image_size % c->patch_size == 0;
    for (int il = 0; il < c->n_layer && ok; il++) {
        clayer *L = &c->layers[il];
        L->wq = lt(c, il, "attn_q.weight"); L->bq = lt(c, il, "attn_q.bias"); L->wk = lt(c, il, "attn_k.weight"); L->bk = lt(c, il, "attn_k.bias");
        L->wv = lt(c, il, "attn_v.weight"); L->bv = lt(c, il, "attn_v.bias"); L->wo = lt(c, il, "attn_out.weight"); L->bo = lt(c, il, "attn_out.bias");
        L->ln1w = lt(c, il, "ln1.weight"); L->ln1b = lt(c, il, "ln1.bias"); L->ln2w = lt(c, il, "ln2.weight"); L->ln2b = lt(c, il, "ln2.bias");
        L->ff_i_w = lt(c, il, "ffn_down.weight"); L->ff_i_b = lt(c, il, "ffn_down.bias");       /* (the converter's names: "down" is the first projection) */
        L->ff_o_w = lt(c, il, "ffn_up.weight"); L->ff_o_b = lt(c, il, "ffn_up.bias");
        ok = L->wq && L->bq && L->wk && L->bk && L->wv && L->bv && L->wo && L->bo && L->ln1w && L->ln1b && L->ln2w && L->ln2b && L->ff_i_w && L->ff_i_b && L->ff_o_w && L->ff_o_b;
    }
    if (!ok) { oq_clip_free(c); return NULL; }
    c->proj_dim = (int)c->mm2w->ne[1];
    return c;
}

int oq_clip_image_size(const oq_clip *c) { return c->image_size; }
int oq_clip_n_patches(const oq_clip *c) { const int g = c->image_size / c->patch_size; return g * g; }
int oq_clip_n_mmproj_embd(const oq_clip *c) { return c->proj_dim; }

/* clip_image_preprocess, LLaVA-1.5 branch */
void oq_clip_preprocess(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out) {
    const int S = c->image_size;
    int tn = nx, tny = ny;
    uint8_t *tmp = NULL;
    const uint8_t *src = rgb;
    if (nx != ny) {
        const int L = nx > ny ? nx : ny;
        tmp = (uint8_t *)malloc((size_t)3 * L * L);
        static const uint8_t bc[3] = {122, 116, 104};
        for (size_t i = 0; i < (size_t)L * L; i++) { tmp[3 * i] = bc[0]; tmp[3 * i + 1] = bc[1]; tmp[3 * i + 2] = bc[2]; }
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++)
                for (int k = 0; k < 3; k++) tmp[3 * ((size_t)y * L + x) + k] = rgb[3 * ((size_t)y * nx + x) + k];
        src = tmp; tn = L; tny = L;
    }
    const float scale = (float)(tn > tny ? tn : tny) / (float)S;
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++)
            for (int k = 0; k < 3; k++) {
                const float sx = ((float)x + 0.5f) * scale - 0.5f, sy = ((float)y + 0.5f) * scale - 0.5f;
                int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
                if (x0 < 0) x0 = 0;
                if (y0 < 0) y0 = 0;
                const int x1 = x0 + 1 < tn - 1 ? x0 + 1 : tn - 1, y1 = y0 + 1 < tny - 1 ? y0 + 1 : tny - 1;
                const float dx = sx - (float)x0, dy = sy - (float)y0;
                const float v00 = src[3 * ((size_t)y0 * tn + x0) + k], v01 = src[3 * ((size_t)y0 * tn + x1) + k];
                const float v10 = src[3 * ((size_t)y1 * tn + x0) + k], v11 = src[3 * ((size_t)y1 * tn + x1) + k];
                const float v0 = v00 * (1.0f - dx) + v01 * dx, v1 = v10 * (1.0f - dx) + v11 * dx;
                const float v = v0 * (1.0f - dy) + v1 * dy;
                float rv = roundf(v);
                if (rv < 0.0f) rv = 0.0f;
                if (rv > 255.0f) rv = 255.0f;
                const uint8_t v2 = (uint8_t)rv;
                out[(size_t)k * S * S + (size_t)y * S + x] = (((float)v2 / 255.0f) - c->mean[k]) / c->std[k];
            }
    free(tmp);
}

/* ---- LLaVA-1.6 */
int oq_clip_max_image_rows(const oq_clip *c) {
    int tiles = 0;
    if (c->spatial_unpad)
        for (int i = 0; i < c->n_pin; i++) {
            const int t = (c->pin[2 * i] / c->image_size) * (c->pin[2 * i + 1] / c->image_size);
            if (t > tiles) tiles = t;
        }
    return oq_clip_n_patches(c) * (1 + tiles);
}
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static float cubic(float p0, float p1, float p2, float p3, float t) {
    /* the cubic of clip.cpp's bicubic_resize around p1: differences to the three neighbours, coefficients formed in double and stored as floats */
    const float d0 = p0 - p1, d2 = p2 - p1, d3 = p3 - p1, a0 = p1;
    const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
    const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
    const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
    return a0 + a1 * t + a2 * t * t + a3 * t * t * t;
}
static uint8_t *bicubic(const uint8_t *src, int nx, int ny, int tw, int th) {
    uint8_t *dst = (uint8_t *)malloc((size_t)3 * tw * th);
    const float tx = (float)nx / (float)tw, ty = (float)ny / (float)th;
    for (int i = 0; i < th; i++)
        for (int j = 0; j < tw; j++) {
            const int x = (int)(tx * j), y = (int)(ty * i);
            const float dx = tx * j - x, dy = ty * i - y;
            for (int k = 0; k < 3; k++) {
                float col[4];
                for (int jj = 0; jj < 4; jj++) {
                    const size_t row = (size_t)clampi(y - 1 + jj, 0, ny - 1) * nx;
                    col[jj] = cubic(src[(row + clampi(x - 1, 0, nx - 1)) * 3 + k], src[(row + clampi(x, 0, nx - 1)) * 3 + k],
                                    src[(row + clampi(x + 1, 0, nx - 1)) * 3 + k], src[(row + clampi(x + 2, 0, nx - 1)) * 3 + k], dx);
                }
                float v = roundf(cubic(col[0], col[1], col[2], col[3], dy));
                if (v < 0.0f) v = 0.0f;
                if (v > 255.0f) v = 255.0f;
                dst[((size_t)i * tw + j) * 3 + k] = (uint8_t)v;
            }
        }
    return dst;
}
static void best_canvas(const oq_clip *c, int ow, int oh, int *bw, int *bh) {
    int max_eff = 0, min_waste = 0x7fffffff;
    *bw = c->pin[0]; *bh = c->pin[1];
    for (int i = 0; i < c->n_pin; i++) {
        const int w = c->pin[2 * i], h = c->pin[2 * i + 1];
        const float sw = (float)w / ow, sh = (float)h / oh, scale = sw < sh ? sw : sh;
        const int dw = (int)(ow * scale), dh = (int)(oh * scale);
        const long long full = (long long)ow * oh, down = (long long)dw * dh;
        const int eff = (int)(down < full ? down : full), waste = w * h - eff;
        if (eff > max_eff || (eff == max_eff && waste < min_waste)) { max_eff = eff; min_waste = waste; *bw = w; *bh = h; }
    }
}
static void window_to_planar(const oq_clip *c, const uint8_t *img, int nx, int x0, int y0, float *out) {
    const int S = c->image_size;
    for (int k = 0; k < 3; k++)
        for (int y = 0; y < S; y++)
            for (int x = 0; x < S; x++)
                out[(size_t)k * S * S + (size_t)y * S + x] = ((float)img[((size_t)(y + y0) * nx + (x + x0)) * 3 + k] / 255.0f - c->mean[k]) / c->std[k];
}
int oq_clip_preprocess_all(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out, int cap_images, int *grid_w, int *grid_h) {
    const int S = c->image_size;
    const size_t per = (size_t)3 * S * S;
    *grid_w = *grid_h = 0;
    if (cap_images < 1) return -1;
    /* pad_to_square = merge type is not "spatial_unpad"; the grid branch is the else of (pad_to_square && nx != ny) */
    if (c->n_pin == 0 || (!c->spatial_unpad && nx != ny)) { oq_clip_preprocess(c, rgb, nx, ny, out); return 1; }
    uint8_t *ov = bicubic(rgb, nx, ny, S, S);
    window_to_planar(c, ov, S, 0, 0, out);
    free(ov);
    if (!c->spatial_unpad) return 1;
    int tw, th;
    best_canvas(c, nx, ny, &tw, &th);
    const float sw = (float)tw / nx, sh = (float)th / ny;
    int nw, nh;
    if (sw < sh) { nw = tw; nh = (int)ceilf(ny * sw); if (nh > th) nh = th; }
    else { nh = th; nw = (int)ceilf(nx * sh); if (nw > tw) nw = tw; }
    uint8_t *rs = bicubic(rgb, nx, ny, nw, nh);
    uint8_t *canvas = (uint8_t *)calloc((size_t)3 * tw * th, 1);
    const int ox = (tw - nw) / 2, oy = (th - nh) / 2;
    for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++)
            for (int k = 0; k < 3; k++) canvas[((size_t)(y + oy) * tw + (x + ox)) * 3 + k] = rs[((size_t)y * nw + x) * 3 + k];
    free(rs);
    const int gw = tw / S, gh = th / S;
    int n = 1;
    if (1 + gw * gh > cap_images) { free(canvas); return -1; }
    for (int gy = 0; gy < gh; gy++)
        for (int gx = 0; gx < gw; gx++) window_to_planar(c, canvas, tw, gx * S, gy * S, out + per * (size_t)n++);
    free(canvas);
    *grid_w = gw; *grid_h = gh;
    return n;
}
int oq_clip_embed(const oq_clip *c, const uint8_t *rgb, int nx, int ny, float *out, int cap_rows, int nth) {
    const int S = c->image_size, G = S / c->patch_size, NP = G * G, E = c->proj_dim, cap_images = oq_clip_max_image_rows(c) / NP;
    float *imgs = (float *)malloc(sizeof(float) * (size_t)3 * S * S * cap_images);
    int gw, gh;
    const int n = oq_clip_preprocess_all(c, rgb, nx, ny, imgs, cap_images, &gw, &gh);
    if (n < 1 || n * NP > cap_rows) { free(imgs); return -1; }
    oq_clip_encode(c, imgs, out, nth);
    float *tile = (float *)malloc(sizeof(float) * (size_t)NP * E);
    for (int t = 1; t < n; t++) {
        oq_clip_encode(c, imgs + (size_t)t * 3 * S * S, tile, nth);
        const int gy = (t - 1) / gw, gx = (t - 1) % gw;
        /* token (py, px) of tile (gy, gx) sits at canvas position (gy * G + py, gx * G + px) of a (gh * G) x (gw * G) row-major sheet */
        for (int py = 0; py < G; py++)
            for (int px = 0; px < G; px++) {
                const size_t dst = (size_t)NP + (size_t)(gy * G + py) * ((size_t)gw * G) + (size_t)(gx * G + px);
                memcpy(out + dst * E, tile + ((size_t)py * G + px) * E, sizeof(float) * (size_t)E);
            }
    }
    free(tile); free(imgs);
    return n * NP;
}

static void ln_rows(const float *x, float *y, int64_t n, int64_t T, float eps, const ct *w, const ct *b) {
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
static void lin(const ct *w, const ct *b, const float *x, int64_t T, float *y, int nth) {
    oq_mul_mat(w->type, w->data, w->ne[1], w->ne[0], x, T, y, nth);
    const int64_t n = w->ne[1];
    for (int64_t t = 0; t < T; t++) oq_add_f32(y + t * n, (const float *)b->data, y + t * n, n);
}
/* ggml's GELU family through the f16 tables */
static float gelu_tab(float x) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f) return x;
    const float xh = oq_fp16_to_fp32(oq_fp32_to_fp16(x));
    const float g = 0.5f * xh * (1.0f + tanhf(0.79788456080286535587989211986876f * xh * (1.0f + 0.044715f * xh * xh)));
    return oq_fp16_to_fp32(oq_fp32_to_fp16(g));
}
static float gelu_quick_tab(float x) {
    const float xh = oq_fp16_to_fp32(oq_fp32_to_fp16(x));
    const float g = xh * (1.0f / (1.0f + expf(-1.702f * xh)));
    return oq_fp16_to_fp32(oq_fp32_to_fp16(g));
}

int oq_clip_encode(const oq_clip *c, const float *img, float *out, int nth) {
    const int S = c->image_size, P = c->patch_size, G = S / P, NP = G * G, T = NP + 1, E = c->n_embd, FF = c->n_ff, H = c->n_head, D = E / H, KP = 3 * P * P;
    float *patches = (float *)malloc(sizeof(float) * (size_t)NP * KP);
    for (int py = 0; py < G; py++)
        for (int px = 0; px < G; px++)
            for (int k = 0; k < 3; k++)
                for (int ky = 0; ky < P; ky++)
                    for (int kx = 0; kx < P; kx++)
                        patches[((size_t)(py * G + px)) * KP + ((size_t)k * P + ky) * P + kx] = img[(size_t)k * S * S + (size_t)(py * P + ky) * S + (px * P + kx)];
    float *emb = (float *)malloc(sizeof(float) * (size_t)T * E), *cur = (float *)malloc(sizeof(float) * (size_t)T * E);
    float *q = (float *)malloc(sizeof(float) * (size_t)T * E), *k = (float *)malloc(sizeof(float) * (size_t)T * E), *v = (float *)malloc(sizeof(float) * (size_t)T * E);
    float *att = (float *)malloc(sizeof(float) * (size_t)T * E), *ff = (float *)malloc(sizeof(float) * (size_t)T * FF), *pr = (float *)malloc(sizeof(float) * (size_t)T);
    float *pos = (float *)malloc(sizeof(float) * (size_t)E);
    /* patch embedding: the convolution kernel [out][c][ky][kx] as a matrix of KP columns */
    oq_mul_mat(c->patch_w->type, c->patch_w->data, E, KP, patches, NP, emb + E, nth);
    memcpy(emb, c->class_embd->data, sizeof(float) * (size_t)E);
    for (int t = 0; t < T; t++) {
        oq_dequantize_row(c->pos_embd->type, c->pos_embd->data + (size_t)t * oq_row_bytes(c->pos_embd->type, E), pos, E);
        oq_add_f32(emb + (size_t)t * E, pos, emb + (size_t)t * E, E);
    }
    ln_rows(emb, emb, E, T, c->eps, c->pre_ln_w, c->pre_ln_b);
    const float qs = 1.0f / sqrtf((float)D);
    /* clip.cpp: a LLaVA projector reads the output of block_count - 1 blocks (get_deepest_feature_layer); the file's last block stays unused */
    for (int il = 0; il < c->n_layer - 1; il++) {
        const clayer *L = &c->layers[il];
        ln_rows(emb, cur, E, T, c->eps, L->ln1w, L->ln1b);
        lin(L->wq, L->bq, cur, T, q, nth);
        for (size_t i = 0; i < (size_t)T * E; i++) q[i] *= qs;
        lin(L->wk, L->bk, cur, T, k, nth);
        lin(L->wv, L->bv, cur, T, v, nth);
        for (int h = 0; h < H; h++)
            for (int tq = 0; tq < T; tq++) {
                const float *qr = q + (size_t)tq * E + (size_t)h * D;
                for (int tk = 0; tk < T; tk++) {
                    const float *kr = k + (size_t)tk * E + (size_t)h * D;
                    double s = 0.0;
                    for (int d = 0; d < D; d++) s += (double)(kr[d] * qr[d]);
                    pr[tk] = (float)s;
                }
                oq_soft_max(pr, NULL, pr, T, 1.0f);
                for (int d = 0; d < D; d++) {
                    double s = 0.0;
                    for (int tk = 0; tk < T; tk++) s += (double)(v[(size_t)tk * E + (size_t)h * D + d] * pr[tk]);
                    att[(size_t)tq * E + (size_t)h * D + d] = (float)s;
                }
            }
        lin(L->wo, L->bo, att, T, cur, nth);
        for (size_t i = 0; i < (size_t)T * E; i++) emb[i] = cur[i] + emb[i];
        ln_rows(emb, cur, E, T, c->eps, L->ln2w, L->ln2b);
        lin(L->ff_i_w, L->ff_i_b, cur, T, ff, nth);
        for (size_t i = 0; i < (size_t)T * FF; i++) ff[i] = c->use_gelu ? gelu_tab(ff[i]) : gelu_quick_tab(ff[i]);
        lin(L->ff_o_w, L->ff_o_b, ff, T, cur, nth);
        for (size_t i = 0; i < (size_t)T * E; i++) emb[i] = emb[i] + cur[i];
    }
    /* the projector on the patch rows (the class row is dropped) */
    const int PD = c->proj_dim;
    float *h1 = (float *)malloc(sizeof(float) * (size_t)NP * PD);
    lin(c->mm0w, c->mm0b, emb + E, NP, h1, nth);
    for (size_t i = 0; i < (size_t)NP * PD; i++) h1[i] = gelu_tab(h1[i]);
    lin(c->mm2w, c->mm2b, h1, NP, out, nth);
    free(patches); free(emb); free(cur); free(q); free(k); free(v); free(att); free(ff); free(pr); free(pos); free(h1);
    return 0;
}
