// clip.h — the image side of a LLaVA request: projector file ("mmproj", general.architecture clip) on the device, image bytes -> RGB, the LLaVA-1.5 and
// LLaVA-1.6 ("anyres" image grid) preprocessing, and the encoder (CLIP ViT tower + MLP projector) -> n_patches rows of the language model's n_embd per
// encoded image (one for LLaVA-1.5; an overview + the tiles of the chosen grid for LLaVA-1.6).
//
// Replaces what the reference takes from llama.cpp's examples/llava for this path (clip.cpp / llava.cpp; the submodule is not in the mount):
//   clip_model_load            /root/reference/src/llama_server_context.cc:187
//   clip_n_mmproj_embd         :216
//   clip_image_load_from_bytes :568      (stb_image there; PNG / BMP / PNM / JPEG decoders here)
//   llava_image_embed_make_with_clip_img :820   (= clip_image_preprocess + clip_image_encode)
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace mi355 {

struct ClipImageU8 { int nx = 0, ny = 0; std::vector<uint8_t> rgb; };      // [ny][nx][3]

// PNG (8-bit grey / RGB / palette, with or without alpha, interlaced or not), BMP (24 / 32 bit), binary PNM (P5 / P6), JPEG (sequential or progressive, 8-bit, Huffman, any
// sampling factors, restart intervals).  Returns an empty string on success, else why not.
std::string clip_image_load_from_bytes(const uint8_t *data, size_t n, ClipImageU8 &out);

struct ClipLayerDev {
    void *wq, *wk, *wv, *wo, *ff_i, *ff_o;                 // f16 [rows][K]
    float *bq, *bk, *bv, *bo, *ln1w, *ln1b, *ln2w, *ln2b, *ff_i_b, *ff_o_b;
};

class ClipModel {
  public:
    ~ClipModel();
    // empty string on success
    std::string load(const std::string &path, int device);
    int image_size = 0, patch_size = 0, n_embd = 0, n_ff = 0, n_head = 0, n_layer = 0, proj_dim = 0;
    bool use_gelu = false;
    float eps = 1e-5f, mean[3] = {0, 0, 0}, stdv[3] = {1, 1, 1};
    // LLaVA-1.6: clip.vision.image_grid_pinpoints = the (width, height) canvases a picture may be fitted to, clip.vision.mm_patch_merge_type ("spatial_unpad"
    // turns the grid on; with anything else a non-square picture is padded the LLaVA-1.5 way and a square one becomes its bicubic overview)
    std::vector<std::pair<int, int>> pinpoints;
    std::string merge_type = "flat";
    bool has_grid() const { return !pinpoints.empty(); }
    int n_patches() const { const int g = image_size / patch_size; return g * g; }
    // the most rows one picture can produce: n_patches for LLaVA-1.5, n_patches * (1 + the tiles of the largest canvas) with a grid
    int max_image_rows() const;
    // clip_image_preprocess (LLaVA-1.5): [3][S][S] floats
    void preprocess(const ClipImageU8 &img, std::vector<float> &out) const;
    // clip_image_preprocess, every image the encoder is to see for one picture: LLaVA-1.5 the one above; with a grid the overview (the whole picture resized to
    // S x S) followed by the S x S tiles of the fitted canvas, row-major; grid_w x grid_h = tiles across / down (0 x 0 without a grid)
    void preprocess_all(const ClipImageU8 &img, std::vector<std::vector<float>> &out, int &grid_w, int &grid_h) const;
    // llava_image_embed_make_with_clip_img: rows [n_rows][proj_dim] of one picture - preprocess_all, encode each, and with a grid the tiles' rows re-ordered into
    // whole-canvas row-major order behind the overview's.  Empty string on success.
    std::string embed(const ClipImageU8 &img, std::vector<float> &rows, int &n_rows);
    // clip_image_encode: img [3][S][S] (host) -> out [n_patches][proj_dim] (host).  Empty string on success.
    std::string encode(const float *img, float *out);
    // the same for n images at once (n <= the images of one picture): img [n][3][S][S] -> out [n][n_patches][proj_dim]
    // grid_w > 0: images 1 .. are the tiles of a canvas grid_w wide; their rows come out in the canvas' row-major order (clip_llava_handle_patches)
    std::string encode_batch(const float *img, int n, float *out, int grid_w = 0);
    uint64_t device_bytes = 0;

  private:
    int device_ = 0, kp_pad_ = 0, max_images_ = 1;
    void *stream_ = nullptr;
    std::vector<void *> allocs_;
    void *patch_w_ = nullptr, *mm0w_ = nullptr, *mm2w_ = nullptr;
    float *class_ = nullptr, *pos_ = nullptr, *pre_w_ = nullptr, *pre_b_ = nullptr, *mm0b_ = nullptr, *mm2b_ = nullptr;
    std::vector<ClipLayerDev> layers_;
    // scratch
    float *d_img_ = nullptr, *d_patches_ = nullptr, *d_pe_ = nullptr, *d_emb_ = nullptr, *d_cur_ = nullptr, *d_q_ = nullptr, *d_k_ = nullptr, *d_v_ = nullptr,
          *d_att_ = nullptr, *d_ff_ = nullptr, *d_h1_ = nullptr, *d_out_ = nullptr;
    void *d_xh_ = nullptr;                 // the activation rows of the projection at hand, rounded to f16
    void *dalloc(size_t bytes);
};

}  // namespace mi355
