"""The LLaVA-1.6 image grid against tests/golden/clip_grid_v1.npz (written by tests/golden/make_golden_clip_grid.py: an independent numpy restatement of
bicubic_resize, select_best_resolution, resize_and_pad_image, the tiles and the row order of a picture's embedding).  CPU: the C oracle; GPU: the device path.
Images bit for bit, grid shape and row order exactly."""
import os

import numpy as np
import pytest

import oracle_py as oq

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_grid_v1.npz"))
CASES = sorted({k.split(".")[0] for k in G.files})


@pytest.fixture(scope="module")
def path(pkg, tmp_path_factory):
    p = str(tmp_path_factory.mktemp("clipgrid") / "tiny-clip-grid.gguf")
    pkg.gguf_synth.write_synthetic_clip(p, "tiny-clip-grid")
    return p


def check(enc, preprocess_all, embed, case):
    rgb = G[case + ".rgb"]
    imgs, gw, gh = preprocess_all(rgb)
    assert (gw, gh) == tuple(int(x) for x in G[case + ".grid"]), case
    assert imgs.shape == G[case + ".imgs"].shape and np.array_equal(imgs, G[case + ".imgs"]), case
    rows = embed(rgb)
    assert rows.shape[0] == enc.n_patches * (1 + gw * gh) <= enc.max_image_rows
    cat = np.concatenate([enc.encode(i) for i in imgs], 0)
    assert np.array_equal(rows, cat[G[case + ".order"]]), case


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_the_golden_vectors(path, case):
    o = oq.OracleClip(path)
    assert o.max_image_rows == 16 * 5
    check(o, o.preprocess_all, o.embed, case)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_path_reproduces_the_golden_vectors(pkg, path, case):
    PIL = pytest.importorskip("PIL.Image")
    import io
    pkg.Backend()
    c = pkg.Clip(path)
    assert c.max_image_rows == 16 * 5

    def embed(rgb):                                 # (through the bytes entry point: a lossless PNG of the same pixels)
        b = io.BytesIO()
        PIL.fromarray(rgb).save(b, "PNG")
        return c.embed_bytes(b.getvalue())
    check(c, c.preprocess_grid, embed, case)
    c.close()
