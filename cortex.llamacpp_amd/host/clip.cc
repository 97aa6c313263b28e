// clip.cc — see clip.h.  The tower runs as a short sequence of launches per block (LayerNorm, three projections, bias / scale, attention, projection, residual,
// LayerNorm, two projections around the GELU, residual): f16 weights on the matrix cores (mmf.hip: activations rounded to f16 as the CPU's f16 dot does),
// everything else f32.  One image is one pass; nothing here is on the token path.
#include "clip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "../csrc/kernels.h"
#include "gguf.h"
#include "parallel_rows.h"

namespace mi355 {

#define CLIP_TRY(x)                                                                                      \
    do {                                                                                                 \
        const hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) return std::string(#x) + ": " + hipGetErrorString(e_);                     \
    } while (0)

ClipModel::~ClipModel() {
    for (void *p : allocs_) (void)hipFree(p);
    if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
}
void *ClipModel::dalloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    allocs_.push_back(p);
    device_bytes += bytes;
    return p;
}

static uint16_t f32_to_f16_bits(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float f16_bits_to_f32(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

std::string ClipModel::load(const std::string &path, int device) {
    GGUFFile f;
    const std::string err = f.open(path);
    if (!err.empty()) return err;
    if (f.get_s("general.architecture", "") != "clip") return "not a projector file: general.architecture is '" + f.get_s("general.architecture", "") + "', expected 'clip'";
    if (!f.get_b("clip.has_vision_encoder", false)) return "projector file without a vision encoder";
    if (!f.get_b("clip.has_llava_projector", false)) return "projector file without a LLaVA projector (clip.has_llava_projector)";
    const std::string ptype = f.get_s("clip.projector_type", "mlp");
    if (ptype != "mlp") return "projector type '" + ptype + "' is not supported (mlp only: LLaVA-1.5)";
    image_size = (int)f.get_u("clip.vision.image_size", 0); patch_size = (int)f.get_u("clip.vision.patch_size", 0);
    n_embd = (int)f.get_u("clip.vision.embedding_length", 0); n_ff = (int)f.get_u("clip.vision.feed_forward_length", 0);
    n_head = (int)f.get_u("clip.vision.attention.head_count", 0); n_layer = (int)f.get_u("clip.vision.block_count", 0);
    eps = (float)f.get_f("clip.vision.attention.layer_norm_epsilon", 1e-5);
    use_gelu = f.get_b("clip.use_gelu", false);
    if (image_size <= 0 || patch_size <= 0 || image_size % patch_size || n_embd <= 0 || n_ff <= 0 || n_head <= 0 || n_layer <= 0 || n_embd % n_head)
        return "projector file: bad vision geometry";
    const int D = n_embd / n_head;
    if (D != 32 && D != 64 && D != 80 && D != 128) return "projector file: head size " + std::to_string(D) + " is not supported";
    if ((n_embd % 16) || (n_ff % 16)) return "projector file: widths must be multiples of 16";
    for (int k = 0; k < 3; k++) { mean[k] = 0.0f; stdv[k] = 1.0f; }
    if (const GGUFValue *v = f.find("clip.vision.image_mean")) if (v->type == GV_ARR && v->elem_type == GV_F32 && v->u >= 3 && v->raw) memcpy(mean, v->raw, 12);
    if (const GGUFValue *v = f.find("clip.vision.image_std")) if (v->type == GV_ARR && v->elem_type == GV_F32 && v->u >= 3 && v->raw) memcpy(stdv, v->raw, 12);
    pinpoints.clear();
    if (const GGUFValue *v = f.find("clip.vision.image_grid_pinpoints")) {
        if (v->type != GV_ARR || (v->elem_type != GV_I32 && v->elem_type != GV_U32) || !v->raw || (v->u % 2) || v->u > 128) return "projector file: bad clip.vision.image_grid_pinpoints";
        const int32_t *pp = reinterpret_cast<const int32_t *>(v->raw);
        for (uint64_t i = 0; i + 1 < v->u; i += 2) {
            // every canvas is a whole number of tiles, and small enough that the rows of one picture stay a sane prompt (at most 64 tiles)
            if (pp[i] <= 0 || pp[i + 1] <= 0 || pp[i] % image_size || pp[i + 1] % image_size || (int64_t)(pp[i] / image_size) * (pp[i + 1] / image_size) > 64)
                return "projector file: clip.vision.image_grid_pinpoints entry " + std::to_string(pp[i]) + "x" + std::to_string(pp[i + 1]) + " is not a grid of " + std::to_string(image_size) + "-pixel tiles";
            pinpoints.emplace_back(pp[i], pp[i + 1]);
        }
    }
    merge_type = f.get_s("clip.vision.mm_patch_merge_type", "flat");

    device_ = device;
    if (hipSetDevice(device) != hipSuccess) return "hipSetDevice failed";
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return "stream creation failed";
    stream_ = st;

    std::string bad;
    // a weight matrix [rows][K]: f16 in the file (what the converter writes); uploaded with rows of ld >= K halves (zero padded)
    auto up_w = [&](const std::string &name, int rows, int K, int ld) -> void * {
        const GGUFTensorInfo *t = f.tensor(name);
        if (!t) { bad = "projector file: tensor " + name + " is missing"; return nullptr; }
        int64_t n = 1;
        for (int d = 0; d < t->n_dims; d++) n *= t->ne[d];
        if (n != (int64_t)rows * K) { bad = "projector file: tensor " + name + " has the wrong shape"; return nullptr; }
        if (t->type != 1) { bad = "projector file: tensor " + name + " is " + ggml_type_name(t->type) + "; f16 weights only"; return nullptr; }
        std::vector<uint16_t> h((size_t)rows * ld, 0);
        const uint16_t *src = reinterpret_cast<const uint16_t *>(t->data);
        for (int r = 0; r < rows; r++) memcpy(h.data() + (size_t)r * ld, src + (size_t)r * K, (size_t)K * 2);
        void *d = dalloc(h.size() * 2);
        if (!d || hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { bad = "device upload failed"; return nullptr; }
        return d;
    };
    // a vector / table of floats: f32 or f16 in the file, f32 on the device
    auto up_f = [&](const std::string &name, size_t n) -> float * {
        const GGUFTensorInfo *t = f.tensor(name);
        if (!t) { bad = "projector file: tensor " + name + " is missing"; return nullptr; }
        int64_t m = 1;
        for (int d = 0; d < t->n_dims; d++) m *= t->ne[d];
        if ((size_t)m != n || (t->type != 0 && t->type != 1)) { bad = "projector file: tensor " + name + " has the wrong shape or type"; return nullptr; }
        std::vector<float> h(n);
        if (t->type == 0) memcpy(h.data(), t->data, n * 4);
        else for (size_t i = 0; i < n; i++) h[i] = f16_bits_to_f32(reinterpret_cast<const uint16_t *>(t->data)[i]);
        float *d = (float *)dalloc(n * 4);
        if (!d || hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) { bad = "device upload failed"; return nullptr; }
        return d;
    };
    const int E = n_embd, FF = n_ff, KP = 3 * patch_size * patch_size, T = n_patches() + 1;
    kp_pad_ = (KP + 15) & ~15;
    const GGUFTensorInfo *mm2 = f.tensor("mm.2.weight");
    if (!mm2 || mm2->n_dims != 2) return "projector file: tensor mm.2.weight is missing";
    proj_dim = (int)mm2->ne[1];
    if (proj_dim < 32 || (proj_dim % 16)) return "projector file: bad projection width";
    class_ = up_f("v.class_embd", (size_t)E);
    patch_w_ = up_w("v.patch_embd.weight", E, KP, kp_pad_);
    pos_ = up_f("v.position_embd.weight", (size_t)T * E);
    pre_w_ = up_f("v.pre_ln.weight", (size_t)E); pre_b_ = up_f("v.pre_ln.bias", (size_t)E);
    mm0w_ = up_w("mm.0.weight", proj_dim, E, E); mm0b_ = up_f("mm.0.bias", (size_t)proj_dim);
    mm2w_ = up_w("mm.2.weight", proj_dim, proj_dim, proj_dim); mm2b_ = up_f("mm.2.bias", (size_t)proj_dim);
    if (!bad.empty()) return bad;
    layers_.resize((size_t)n_layer);
    for (int il = 0; il < n_layer; il++) {
        const std::string p = "v.blk." + std::to_string(il) + ".";
        ClipLayerDev &L = layers_[(size_t)il];
        L.wq = up_w(p + "attn_q.weight", E, E, E); L.bq = up_f(p + "attn_q.bias", (size_t)E);
        L.wk = up_w(p + "attn_k.weight", E, E, E); L.bk = up_f(p + "attn_k.bias", (size_t)E);
        L.wv = up_w(p + "attn_v.weight", E, E, E); L.bv = up_f(p + "attn_v.bias", (size_t)E);
        L.wo = up_w(p + "attn_out.weight", E, E, E); L.bo = up_f(p + "attn_out.bias", (size_t)E);
        L.ln1w = up_f(p + "ln1.weight", (size_t)E); L.ln1b = up_f(p + "ln1.bias", (size_t)E);
        L.ln2w = up_f(p + "ln2.weight", (size_t)E); L.ln2b = up_f(p + "ln2.bias", (size_t)E);
        // the converter's names: "ffn_down" is the FIRST projection (n_embd -> n_ff), "ffn_up" the second
        L.ff_i = up_w(p + "ffn_down.weight", FF, E, E); L.ff_i_b = up_f(p + "ffn_down.bias", (size_t)FF);
        L.ff_o = up_w(p + "ffn_up.weight", E, FF, FF); L.ff_o_b = up_f(p + "ffn_up.bias", (size_t)E);
        if (!bad.empty()) return bad;
    }
    const int NP = n_patches();
    // scratch for every image of one picture at once (one for LLaVA-1.5; the overview + the tiles of the largest canvas with an image grid)
    const size_t MI = (size_t)(max_image_rows() / NP);
    max_images_ = (int)MI;
    d_img_ = (float *)dalloc(MI * 3 * image_size * image_size * 4);
    d_patches_ = (float *)dalloc(MI * NP * kp_pad_ * 4);
    d_pe_ = (float *)dalloc(MI * NP * E * 4);
    d_emb_ = (float *)dalloc(MI * T * E * 4); d_cur_ = (float *)dalloc(MI * T * E * 4);
    d_q_ = (float *)dalloc(MI * T * E * 4); d_k_ = (float *)dalloc(MI * T * E * 4); d_v_ = (float *)dalloc(MI * T * E * 4);
    d_att_ = (float *)dalloc(MI * T * E * 4); d_ff_ = (float *)dalloc(MI * T * FF * 4);
    d_h1_ = (float *)dalloc(MI * T * proj_dim * 4); d_out_ = (float *)dalloc(MI * T * proj_dim * 4);
    d_xh_ = dalloc(MI * T * std::max(std::max(FF, E), std::max(proj_dim, kp_pad_)) * 2);
    if (!d_img_ || !d_patches_ || !d_pe_ || !d_emb_ || !d_cur_ || !d_q_ || !d_k_ || !d_v_ || !d_att_ || !d_ff_ || !d_h1_ || !d_out_ || !d_xh_) return "out of device memory";
    if (NP < 8) return "projector file: fewer than 8 patches";
    return "";
}

// clip_image_preprocess, the LLaVA-1.5 branch: pad to a square with the mean colour (top-left aligned), bilinear resample with the half-pixel mapping, round to
// a byte, normalise; planar output
void ClipModel::preprocess(const ClipImageU8 &img, std::vector<float> &out) const {
    const int S = image_size;
    out.assign((size_t)3 * S * S, 0.0f);
    // the square is never built (a 16384 x 1 picture would ask for 805 MB of it): a pixel outside the picture IS the pad colour
    static const uint8_t bc[3] = {122, 116, 104};
    const int tn = std::max(img.nx, img.ny), tny = tn;
    const int nx = img.nx, ny = img.ny;
    const uint8_t *rgb = img.rgb.data();
    auto px = [&](int y, int x, int k) -> float { return (float)(x < nx && y < ny ? rgb[3 * ((size_t)y * nx + x) + k] : bc[k]); };
    const float scale = (float)std::max(tn, tny) / (float)S;
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++)
            for (int k = 0; k < 3; k++) {
                const float sx = ((float)x + 0.5f) * scale - 0.5f, sy = ((float)y + 0.5f) * scale - 0.5f;
                const int x0 = std::max(0, (int)floorf(sx)), y0 = std::max(0, (int)floorf(sy));
                const int x1 = std::min(x0 + 1, tn - 1), y1 = std::min(y0 + 1, tny - 1);
                const float dx = sx - (float)x0, dy = sy - (float)y0;
                const float v00 = px(y0, x0, k), v01 = px(y0, x1, k);
                const float v10 = px(y1, x0, k), v11 = px(y1, x1, k);
                const float v0 = v00 * (1.0f - dx) + v01 * dx, v1 = v10 * (1.0f - dx) + v11 * dx;
                const float v = v0 * (1.0f - dy) + v1 * dy;
                const uint8_t v2 = (uint8_t)std::min(std::max(roundf(v), 0.0f), 255.0f);
                out[(size_t)k * S * S + (size_t)y * S + x] = (((float)v2 / 255.0f) - mean[k]) / stdv[k];
            }
}

int ClipModel::max_image_rows() const {
    int tiles = 0;
    if (merge_type == "spatial_unpad")
        for (const auto &p : pinpoints) tiles = std::max(tiles, (p.first / image_size) * (p.second / image_size));
    return n_patches() * (1 + tiles);
}

// ---- LLaVA-1.6 preprocessing (clip.cpp: bicubic_resize, select_best_resolution, resize_and_pad_image, divide_to_patches_u8)
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// the cubic through four samples with Catmull-Rom style coefficients, evaluated separably (rows first, then the column), the source index truncated (not
// half-pixel centred), edges clamped, the result rounded and clamped to a byte
static void bicubic_resize(const ClipImageU8 &img, ClipImageU8 &dst, int tw, int th) {
    const int nx = img.nx, ny = img.ny;
    dst.nx = tw; dst.ny = th;
    dst.rgb.assign((size_t)3 * tw * th, 0);
    const float tx = (float)nx / (float)tw, ty = (float)ny / (float)th;
    parallel_rows(th, (int64_t)tw * th, [&](int i0, int i1) {
    for (int i = i0; i < i1; i++)
        for (int j = 0; j < tw; j++) {
            const int x = (int)(tx * (float)j), y = (int)(ty * (float)i);
            const float dx = tx * (float)j - (float)x, dy = ty * (float)i - (float)y;
            for (int k = 0; k < 3; k++) {
                float C[4];
                for (int jj = 0; jj <= 3; jj++) {
                    const uint8_t *row = img.rgb.data() + (size_t)3 * clampi(y - 1 + jj, 0, ny - 1) * nx;
                    const float a0 = row[3 * clampi(x, 0, nx - 1) + k];
                    const float d0 = (float)row[3 * clampi(x - 1, 0, nx - 1) + k] - a0;
                    const float d2 = (float)row[3 * clampi(x + 1, 0, nx - 1) + k] - a0;
                    const float d3 = (float)row[3 * clampi(x + 2, 0, nx - 1) + k] - a0;
                    const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);        // (the coefficients in double, rounded to float; the polynomial in float)
                    const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
                    const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
                    C[jj] = a0 + a1 * dx + a2 * dx * dx + a3 * dx * dx * dx;
                }
                const float a0 = C[1], d0 = C[0] - C[1], d2 = C[2] - C[1], d3 = C[3] - C[1];
                const float a1 = (float)(-1.0 / 3 * d0 + d2 - 1.0 / 6 * d3);
                const float a2 = (float)(1.0 / 2 * d0 + 1.0 / 2 * d2);
                const float a3 = (float)(-1.0 / 6 * d0 - 1.0 / 2 * d2 + 1.0 / 6 * d3);
                const float Cc = a0 + a1 * dy + a2 * dy * dy + a3 * dy * dy * dy;
                dst.rgb[(size_t)3 * ((size_t)i * tw + j) + k] = (uint8_t)std::min(std::max(roundf(Cc), 0.0f), 255.0f);
            }
        }
    });
}

// the canvas that keeps the most of the picture's pixels after an aspect-preserving fit; among equals, the one that wastes the least area
static std::pair<int, int> select_best_resolution(int ow, int oh, const std::vector<std::pair<int, int>> &cands) {
    std::pair<int, int> best = cands.front();
    int max_eff = 0, min_waste = INT32_MAX;
    for (const auto &r : cands) {
        const float scale = std::min((float)r.first / (float)ow, (float)r.second / (float)oh);
        const int dw = (int)((float)ow * scale), dh = (int)((float)oh * scale);
        const int eff = (int)std::min((int64_t)dw * dh, (int64_t)ow * oh);
        const int waste = r.first * r.second - eff;
        if (eff > max_eff || (eff == max_eff && waste < min_waste)) { max_eff = eff; min_waste = waste; best = r; }
    }
    return best;
}

// aspect-preserving bicubic fit, centred on a black canvas
static void resize_and_pad(const ClipImageU8 &img, ClipImageU8 &out, int tw, int th) {
    const float sw = (float)tw / (float)img.nx, sh = (float)th / (float)img.ny;
    int nw, nh;
    if (sw < sh) { nw = tw; nh = std::min((int)ceilf((float)img.ny * sw), th); }
    else { nh = th; nw = std::min((int)ceilf((float)img.nx * sh), tw); }
    ClipImageU8 rs;
    bicubic_resize(img, rs, nw, nh);
    out.nx = tw; out.ny = th;
    out.rgb.assign((size_t)3 * tw * th, 0);
    const int ox = (tw - nw) / 2, oy = (th - nh) / 2;
    for (int y = 0; y < nh; y++) memcpy(out.rgb.data() + (size_t)3 * ((size_t)(y + oy) * tw + ox), rs.rgb.data() + (size_t)3 * y * nw, (size_t)3 * nw);
}

void ClipModel::preprocess_all(const ClipImageU8 &img, std::vector<std::vector<float>> &out, int &grid_w, int &grid_h) const {
    out.clear();
    grid_w = grid_h = 0;
    // (clip_image_preprocess pads to a square unless the merge type is "spatial_unpad", and reaches the grid branch only when it did NOT pad: a file with a grid
    // but another merge type treats a non-square picture the LLaVA-1.5 way and a square one as its bicubic overview)
    if (!has_grid() || (merge_type != "spatial_unpad" && img.nx != img.ny)) { out.emplace_back(); preprocess(img, out.back()); return; }
    const int S = image_size;
    auto normalise = [&](const ClipImageU8 &im, int x0, int y0, std::vector<float> &f) {       // the S x S window at (x0, y0) -> planar, normalised
        f.resize((size_t)3 * S * S);
        for (int k = 0; k < 3; k++)
            for (int y = 0; y < S; y++)
                for (int x = 0; x < S; x++)
                    f[(size_t)k * S * S + (size_t)y * S + x] = ((float)im.rgb[(size_t)3 * ((size_t)(y + y0) * im.nx + (x + x0)) + k] / 255.0f - mean[k]) / stdv[k];
    };
    ClipImageU8 overview;
    bicubic_resize(img, overview, S, S);          // (the whole picture, aspect not kept)
    out.emplace_back();
    normalise(overview, 0, 0, out.back());
    if (merge_type != "spatial_unpad") return;
    const std::pair<int, int> best = select_best_resolution(img.nx, img.ny, pinpoints);
    ClipImageU8 canvas;
    resize_and_pad(img, canvas, best.first, best.second);
    grid_w = best.first / S; grid_h = best.second / S;
    for (int gy = 0; gy < grid_h; gy++)
        for (int gx = 0; gx < grid_w; gx++) { out.emplace_back(); normalise(canvas, gx * S, gy * S, out.back()); }
}

std::string ClipModel::embed(const ClipImageU8 &img, std::vector<float> &rows, int &n_rows) {
    std::vector<std::vector<float>> imgs;
    int gw = 0, gh = 0;
    preprocess_all(img, imgs, gw, gh);
    n_rows = n_patches() * (int)imgs.size();
    rows.resize((size_t)n_rows * proj_dim);
    if (imgs.size() == 1) return encode(imgs[0].data(), rows.data());
    // all images of the picture through the tower together (the projections see n x 577 rows: more workgroups than CUs; attention stays within an image)
    const size_t per = (size_t)3 * image_size * image_size;
    std::vector<float> flat(per * imgs.size());
    for (size_t i = 0; i < imgs.size(); i++) memcpy(flat.data() + i * per, imgs[i].data(), per * sizeof(float));
    // clip_llava_handle_patches: the overview's rows, then the tiles' rows - each tile G x G row-major - re-ordered to the canvas' (grid_h * G) x (grid_w * G)
    // row-major order: the copies out of device memory land there directly
    return encode_batch(flat.data(), (int)imgs.size(), rows.data(), gw);
}

std::string ClipModel::encode(const float *img, float *out) { return encode_batch(img, 1, out, 0); }

// n images [n][3][S][S] -> out [n][n_patches][proj_dim]: every row-wise step runs over the n x T rows at once, attention within each image.  grid_w > 0: images
// 1 .. n - 1 are the tiles, row-major, of a canvas grid_w tiles wide, and their rows are written in the canvas' row-major order behind image 0's.
std::string ClipModel::encode_batch(const float *img, int n, float *out, int grid_w) {
    if (n < 1 || n > max_images_) return "clip encode: bad image count";
    if (hipSetDevice(device_) != hipSuccess) return "hipSetDevice failed";
    hipStream_t st = (hipStream_t)stream_;
    const int S = image_size, E = n_embd, FF = n_ff, H = n_head, D = E / H, NP = n_patches(), T1 = NP + 1, T = n * T1;
    CLIP_TRY(hipMemcpyAsync(d_img_, img, (size_t)n * 3 * S * S * 4, hipMemcpyHostToDevice, st));
    for (int i = 0; i < n; i++) CLIP_TRY(launch_clip_im2col(d_img_ + (size_t)i * 3 * S * S, S, patch_size, kp_pad_, d_patches_ + (size_t)i * NP * kp_pad_, st));
    // Every projection reads its activation rows as f16 (what the CPU's f16 dot product does to them): rounded once per row here, not once per tile in the GEMM;
    // bias, the scale of Q and the residual row are the GEMM's epilogue.  MI355_CLIP_XH=0 runs the unfused form (f32 rows into the GEMM, bias and residual as
    // launches of their own) - the same values in the same order, kept so that a test can hold the two against each other.
    const char *sw = getenv("MI355_CLIP_XH");
    const bool xh = !(sw && atoi(sw) == 0);
    // y = resid + (W x + bias) * scale; have_h: d_xh_ already holds these rows as f16 (the producer wrote them, or the projection before read the same rows);
    // tmp: [T_][rows] scratch of the unfused form when resid is given
    auto proj = [&](const void *w, const float *bias, int rows, int K, const float *x, bool have_h, int T_, float *y, float scale, bool do_scale, const float *resid,
                    float *tmp) -> hipError_t {
        hipError_t e = hipSuccess;
        if (xh) {
            if (!have_h && (e = launch_f32_to_f16(x, d_xh_, (size_t)T_ * K, st)) != hipSuccess) return e;
            return launch_mmf16_xh((const uint8_t *)w, rows, K, d_xh_, T_, y, rows, resid, bias, scale, do_scale, st);
        }
        float *dst = resid ? tmp : y;
        if ((e = launch_mmf16((const uint8_t *)w, rows, K, x, T_, dst, rows, nullptr, st)) != hipSuccess) return e;
        if (bias && (e = launch_clip_bias(dst, bias, rows, T_, scale, do_scale, st)) != hipSuccess) return e;
        if (resid) e = launch_add(resid, dst, y, (int64_t)T_ * rows, st);
        return e;
    };
    void *const h = xh ? d_xh_ : nullptr;                       // where the producers leave the f16 copy of their rows
    CLIP_TRY(proj(patch_w_, nullptr, E, kp_pad_, d_patches_, false, n * NP, d_pe_, 1.0f, false, nullptr, nullptr));
    for (int i = 0; i < n; i++) CLIP_TRY(launch_clip_embed(d_pe_ + (size_t)i * NP * E, class_, pos_, E, T1, d_emb_ + (size_t)i * T1 * E, st));
    CLIP_TRY(launch_layer_norm(d_emb_, pre_w_, pre_b_, E, T, eps, d_emb_, st));
    const float qs = 1.0f / sqrtf((float)D);
    // clip.cpp runs the tower up to the feature layer a LLaVA projector reads: block_count - 1 blocks (get_deepest_feature_layer: `hparams.n_layer - 1`, + 1
    // only for the minicpmv / glm / qwen2vl projectors).  The converter has already dropped the tower's last block and written block_count = layers - 1, so of a
    // ViT-L/14-336's 24 blocks the file holds 23 and 22 run; the file's last block is loaded and unused, as upstream.  (Rounds 4-5 ran all of the file's blocks.)
    const int n_run = n_layer - 1;
    for (int il = 0; il < n_run; il++) {
        const ClipLayerDev &L = layers_[(size_t)il];
        CLIP_TRY(launch_layer_norm_h(d_emb_, L.ln1w, L.ln1b, E, T, eps, d_cur_, h, st));
        CLIP_TRY(proj(L.wq, L.bq, E, E, d_cur_, true, T, d_q_, qs, true, nullptr, nullptr));
        CLIP_TRY(proj(L.wk, L.bk, E, E, d_cur_, true, T, d_k_, 1.0f, false, nullptr, nullptr));
        CLIP_TRY(proj(L.wv, L.bv, E, E, d_cur_, true, T, d_v_, 1.0f, false, nullptr, nullptr));
        CLIP_TRY(launch_clip_attn(d_q_, d_k_, d_v_, T1, H, D, d_att_, h, n, st));
        CLIP_TRY(proj(L.wo, L.bo, E, E, d_att_, true, T, d_emb_, 1.0f, false, d_emb_, d_cur_));
        CLIP_TRY(launch_layer_norm_h(d_emb_, L.ln2w, L.ln2b, E, T, eps, d_cur_, h, st));
        CLIP_TRY(proj(L.ff_i, L.ff_i_b, FF, E, d_cur_, true, T, d_ff_, 1.0f, false, nullptr, nullptr));
        CLIP_TRY(launch_clip_gelu(d_ff_, (size_t)T * FF, !use_gelu, h, st));
        CLIP_TRY(proj(L.ff_o, L.ff_o_b, E, FF, d_ff_, true, T, d_emb_, 1.0f, false, d_emb_, d_cur_));
    }
    // the projector on the patch rows (the class row, row 0 of an image, is dropped: one image projects its 576 rows; several project every row and leave
    // the class rows behind when the result is copied out)
    const float *px = n == 1 ? d_emb_ + E : d_emb_;
    const int PT = n == 1 ? NP : T;
    CLIP_TRY(proj(mm0w_, mm0b_, proj_dim, E, px, false, PT, d_h1_, 1.0f, false, nullptr, nullptr));
    CLIP_TRY(launch_clip_gelu(d_h1_, (size_t)PT * proj_dim, false, h, st));
    CLIP_TRY(proj(mm2w_, mm2b_, proj_dim, proj_dim, d_h1_, true, PT, d_out_, 1.0f, false, nullptr, nullptr));
    if (n == 1) CLIP_TRY(hipMemcpyAsync(out, d_out_, (size_t)NP * proj_dim * 4, hipMemcpyDeviceToHost, st));
    else {
        const int G = image_size / patch_size;
        for (int i = 0; i < n; i++) {
            const float *src = d_out_ + ((size_t)i * T1 + 1) * proj_dim;
            if (i == 0 || grid_w <= 0) { CLIP_TRY(hipMemcpyAsync(out + (size_t)i * NP * proj_dim, src, (size_t)NP * proj_dim * 4, hipMemcpyDeviceToHost, st)); continue; }
            const int gy = (i - 1) / grid_w, gx = (i - 1) % grid_w;
            for (int py = 0; py < G; py++) {
                const size_t dst_row = (size_t)NP + ((size_t)(gy * G + py) * grid_w + gx) * G;
                CLIP_TRY(hipMemcpyAsync(out + dst_row * proj_dim, src + (size_t)py * G * proj_dim, (size_t)G * proj_dim * 4, hipMemcpyDeviceToHost, st));
            }
        }
    }
    CLIP_TRY(hipStreamSynchronize(st));
    return "";
}

}  // namespace mi355
