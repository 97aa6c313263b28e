"""The image side of a LLaVA request on the GPU: projector file -> device, clip_image_preprocess, clip_image_encode, against the CPU
restatement (oracle/oq_clip.c).  Reference call sites: clip_model_load (/root/reference/src/llama_server_context.cc:187), clip_image_load_from_bytes (:568),
llava_image_embed_make_with_clip_img (:820).  Files are synthetic (gguf_synth.write_synthetic_clip: the converter's keys and tensor names, random weights)."""
import io
import os

import numpy as np
import pytest

import oracle_py as oq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be(pkg):
    return pkg.Backend()


def make_clip(pkg, tmp_models, cfg):
    path = os.path.join(tmp_models, f"mmproj-{cfg}.gguf")
    if not os.path.exists(path):
        pkg.gguf_synth.write_synthetic_clip(path, cfg)
    return path


def photo(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x * y) % 256)], -1).astype(np.int32)
    img += rng.integers(-40, 41, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(56, 56), (70, 40), (31, 90), (300, 200), (7, 5)])
def test_preprocess_is_the_cpu_paths(be, pkg, tmp_models, w, h):
    """pad to a square with the mean colour, bilinear resample, normalise: the same floats as the CPU restatement, bit for bit (square, wide, tall, larger and
    much smaller than the tower's input)"""
    path = make_clip(pkg, tmp_models, "tiny-clip")
    c, o = pkg.Clip(path), oq.OracleClip(path)
    rgb = photo(w, h, w * 31 + h)
    assert np.array_equal(c.preprocess(rgb), o.preprocess(rgb))
    c.close(); o.close()


@pytest.mark.parametrize("cfg", ["tiny-clip", "tiny-clip-d128", "tiny-clip-gelu", "clip-d128-336"])
def test_encode_matches_oracle(be, pkg, tmp_models, cfg):
    """The tower + projector on three geometries (head size 64 with quick-GELU, 4 heads x 3 blocks, the GELU variant with a 4096-wide projector).  The projections
    round their activations to f16 on both sides; what differs is the f32 summation order and - rarely - an f16 table entry of the GELU picked one step apart:
    held to 2e-3 of the largest output, typical element 1e-4."""
    path = make_clip(pkg, tmp_models, cfg)
    c, o = pkg.Clip(path), oq.OracleClip(path)
    assert (c.n_patches, c.n_embd, c.image_size) == (o.n_patches, o.n_embd, o.image_size)
    for seed in (1, 2):
        img = o.preprocess(photo(90, 60, seed))
        got, ref = c.encode(img), o.encode(img)
        den = float(np.abs(ref).max())
        err = np.abs(got - ref) / den
        if os.environ.get("MI355_TEST_RECORD_FLIPS"):
            with open(os.environ["MI355_TEST_RECORD_FLIPS"], "a") as f:
                f.write(f"clip encode {cfg} seed {seed}: max {err.max():.3g} median {np.median(err):.3g} (|ref| max {den:.3g})\n")
        assert err.max() <= 2e-3 and np.median(err) <= 2e-4, (err.max(), np.median(err))
    c.close(); o.close()


def test_image_bytes_to_embedding_rows(be, pkg, tmp_models):
    """llava_image_embed_make_with_clip_img on encoded bytes: decode + preprocess + encode in one call = the three steps chained; PNG and JPEG"""
    PIL = pytest.importorskip("PIL.Image")
    path = make_clip(pkg, tmp_models, "tiny-clip")
    c = pkg.Clip(path)
    rgb = photo(120, 80, 3)
    for fmt in ("PNG", "JPEG"):
        b = io.BytesIO()
        PIL.fromarray(rgb).save(b, fmt)
        rows = c.embed_bytes(b.getvalue())
        dec = c.load_image(b.getvalue())
        if fmt == "PNG":
            assert np.array_equal(dec, rgb)
        assert rows.shape == (c.n_patches, c.n_embd)
        assert np.array_equal(rows, c.encode(c.preprocess(dec)))
    with pytest.raises(pkg.MI355Error, match="unknown format"):
        c.embed_bytes(b"not an image at all, just some bytes")
    c.close()


@pytest.mark.parametrize("w,h", [(150, 60), (40, 170), (100, 100), (9, 6), (640, 480), (56, 56)])
def test_image_grid_matches_oracle(be, pkg, tmp_models, w, h):
    """LLaVA-1.6: the overview + the tiles of the best canvas are the CPU restatement's images bit for bit, the grid shape is the same, and the rows of the
    picture follow the restatement's (same tolerance as one encode) in the same order"""
    path = make_clip(pkg, tmp_models, "tiny-clip-grid")
    c, o = pkg.Clip(path), oq.OracleClip(path)
    assert c.max_image_rows == o.max_image_rows == 80
    rgb = photo(w, h, w * 7 + h)
    (gi, gw, gh), (oi, ow, oh) = c.preprocess_grid(rgb), o.preprocess_all(rgb)
    assert (gw, gh) == (ow, oh) and np.array_equal(gi, oi)
    PIL = pytest.importorskip("PIL.Image")
    b = io.BytesIO()
    PIL.fromarray(rgb).save(b, "PNG")
    got, ref = c.embed_bytes(b.getvalue()), o.embed(rgb)
    assert got.shape == ref.shape == (16 * (1 + gw * gh), c.n_embd)
    err = np.abs(got - ref) / float(np.abs(ref).max())
    assert err.max() <= 2e-3 and np.median(err) <= 2e-4, (err.max(), np.median(err))
    c.close(); o.close()


def test_image_grid_with_flat_merge_encodes_the_overview_only(be, pkg, tmp_models):
    """a grid in the file but clip.vision.mm_patch_merge_type "flat": one image a picture - a non-square one padded the LLaVA-1.5 way, a square one resized
    with the bicubic filter (clip_image_preprocess pads unless the merge type is "spatial_unpad", and enters the grid branch only when it did not pad)"""
    path = make_clip(pkg, tmp_models, "tiny-clip-grid-flat")
    c, o = pkg.Clip(path), oq.OracleClip(path)
    assert c.max_image_rows == o.max_image_rows == 16
    for w, h, padded in ((150, 60, True), (90, 90, False)):
        rgb = photo(w, h, 5)
        (gi, gw, gh), (oi, ow, oh) = c.preprocess_grid(rgb), o.preprocess_all(rgb)
        assert (gw, gh) == (ow, oh) == (0, 0) and gi.shape[0] == 1 and np.array_equal(gi, oi)
        assert np.array_equal(gi[0], c.preprocess(rgb)) == padded
    c.close(); o.close()


@pytest.mark.parametrize("cfg", ["tiny-clip", "tiny-clip-d128", "clip-d128-336", "clip-vit-l-336"])
def test_tiled_attention_and_small_gemm_tiles_change_no_bit(be, pkg, tmp_models, cfg, monkeypatch):
    """the tower's LDS-tiled attention against the one-wave-per-query kernel (and the matrix-core attention within rounding of them), the LDS-staged f16 GEMM against the direct one and its 64 x 64 workgroup tiles against the 128 x 128 ones, and activation
    rows rounded to f16 once per projection against the rounding inside the GEMM: the same sums in the same order - the embedding rows must be identical (head size 64 and 128; 17, 37 and 577 rows: part tiles, padded key chunks)"""
    path = make_clip(pkg, tmp_models, cfg)
    c = pkg.Clip(path)
    img = c.preprocess(photo(90, 60, 11))
    served = c.encode(img)                                       # head size 64: the attention on the f32 matrix cores
    monkeypatch.setenv("MI355_CLIP_ATTN_MFMA", "0")
    new = c.encode(img)
    # the matrix core adds a dot product's f32 terms in another order than the VALU kernels: last-bit differences in the attention, which the f16 roundings
    # downstream (activation rows, GELU table) turn into an occasional step - the bar of the comparison with the CPU restatement
    err = np.abs(served - new) / float(np.abs(new).max())
    print(f"matrix-core attention vs VALU kernels, {cfg}: max {err.max():.3g} median {np.median(err):.3g}")
    assert err.max() <= 2e-3 and np.median(err) <= 2e-4, (err.max(), np.median(err))
    monkeypatch.setenv("MI355_CLIP_ATTN_TILED", "0")
    assert np.array_equal(c.encode(img), new)
    monkeypatch.setenv("MI355_MMF16_LDS", "0")                   # the projections straight from global memory instead of through LDS
    assert np.array_equal(c.encode(img), new)
    monkeypatch.setenv("MI355_MMF16_TILE", "128")
    assert np.array_equal(c.encode(img), new)
    monkeypatch.setenv("MI355_CLIP_XH", "0")                     # the activation rows rounded to f16 inside the GEMM instead of beforehand
    assert np.array_equal(c.encode(img), new)
    monkeypatch.delenv("MI355_MMF16_TILE")
    assert np.array_equal(c.encode(img), new)
    c.close()


def test_load_refusals(be, pkg, tmp_models):
    """a language-model file is not a projector file; a missing file names itself"""
    lm = os.path.join(tmp_models, "tiny-for-clip.gguf")
    pkg.gguf_synth.write_synthetic_llama(lm, "tiny", "q4_k_m")
    with pytest.raises(pkg.MI355Error, match="not a projector file"):
        pkg.Clip(lm)
    with pytest.raises(pkg.MI355Error):
        pkg.Clip(os.path.join(tmp_models, "does-not-exist.gguf"))


def test_full_size_tower_matches_oracle(be, pkg, tmp_models):
    """CLIP ViT-L/14-336 as LLaVA-1.5-7B uses it: 577 rows of 1024 through 23 blocks, 576 rows of 4096 out"""
    path = make_clip(pkg, tmp_models, "clip-vit-l-336")
    c, o = pkg.Clip(path), oq.OracleClip(path)
    assert (c.n_patches, c.n_embd, c.image_size) == (576, 4096, 336)
    img = o.preprocess(photo(500, 375, 7))
    got, ref = c.encode(img), o.encode(img)
    err = np.abs(got - ref) / float(np.abs(ref).max())
    assert err.max() <= 5e-3 and np.median(err) <= 3e-4, (err.max(), np.median(err))
    c.close(); o.close()
