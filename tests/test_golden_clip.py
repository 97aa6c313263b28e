"""The LLaVA image path against tests/golden/clip_v1.npz (written by tests/golden/make_golden_clip.py: an independent numpy restatement - preprocessing in the CPU
path's float32 expression, the encoder in float64 over the operand values the CPU path holds).  CPU: the C oracle; GPU: the device path.  Preprocessing must
match bit for bit; the encoder within 2e-3 of the largest output (f32 sums against float64; a GELU table entry one step apart now and then)."""
import os

import numpy as np
import pytest

import oracle_py as oq

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_v1.npz"))
CASES = sorted({k.rsplit(".", 1)[0] for k in G.files})


@pytest.fixture(scope="module")
def files(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("clipg")
    out = {}
    for cfg in ("tiny-clip", "tiny-clip-gelu"):
        out[cfg] = str(d / (cfg + ".gguf"))
        pkg.gguf_synth.write_synthetic_clip(out[cfg], cfg)
    return out


def check(enc, case, files_):
    cfg, name = case.split(".")
    img = enc.preprocess(G[case + ".rgb"])
    assert np.array_equal(img, G[case + ".img"]), (case, float(np.abs(img - G[case + ".img"]).max()))
    if case + ".emb" in G.files:
        ref = G[case + ".emb"]
        got = enc.encode(img)
        err = np.abs(got - ref) / float(np.abs(ref).max())
        assert err.max() <= 2e-3 and np.median(err) <= 2e-4, (case, float(err.max()), float(np.median(err)))


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_the_golden_vectors(files, case):
    o = oq.OracleClip(files[case.split(".")[0]])
    check(o, case, files)
    o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_device_path_reproduces_the_golden_vectors(pkg, files, case):
    pkg.Backend()
    c = pkg.Clip(files[case.split(".")[0]])
    check(c, case, files)
    c.close()
