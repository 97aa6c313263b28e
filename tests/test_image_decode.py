"""Image bytes -> RGB (host/image_decode.cc behind mi355_clip_image_load_from_bytes; the reference: clip_image_load_from_bytes -> stb_image,
/root/reference/src/llama_server_context.cc:568).  Host code only: runs without a GPU.  Files are written here with Pillow and compared with Pillow's own decode -
bit-exact for the lossless formats, within the spread of two conforming JPEG decoders (inverse transform and chroma upsampling are implementation choices) for JPEG."""
import ctypes as C
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def decode(pkg, data: bytes) -> np.ndarray:
    lib = pkg.load_library()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    nx, ny = C.c_int32(0), C.c_int32(0)
    rc = lib.mi355_clip_image_load_from_bytes(buf, len(data), C.byref(nx), C.byref(ny), None, 0)
    if rc != 0:
        raise RuntimeError(lib.mi355_last_error().decode())
    rgb = np.empty((ny.value, nx.value, 3), np.uint8)
    assert lib.mi355_clip_image_load_from_bytes(buf, len(data), C.byref(nx), C.byref(ny), rgb.ctypes.data, rgb.nbytes) == 0
    return rgb


def picture(w, h, seed=0):
    """smooth gradients + a few hard edges + noise: every PNG filter type and a realistic JPEG spectrum"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 127 // max(w + h - 2, 1))], -1).astype(np.int32)
    img[h // 4:h // 2, w // 3:2 * w // 3] = (250, 20, 60)
    img += rng.integers(-12, 13, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__ as ge
    return ge.load_pkg()


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
@pytest.mark.parametrize("w,h", [(37, 23), (64, 64), (1, 5), (200, 3)])
def test_png_matches_pillow(pkg, mode, w, h):
    im = PIL.fromarray(picture(w, h, w * h)).convert(mode)
    b = io.BytesIO()
    im.save(b, "PNG")
    got = decode(pkg, b.getvalue())
    want = np.asarray(PIL.open(io.BytesIO(b.getvalue())).convert("RGB"))
    if mode in ("RGBA", "LA"):      # alpha is dropped, not blended
        want = np.asarray(im.convert("RGB"))
    assert got.shape == want.shape and np.array_equal(got, want)


def test_png_compression_levels_and_sixteen_bit(pkg):
    a = picture(90, 70, 5)
    for level in (0, 1, 9):         # stored blocks, fixed / dynamic Huffman
        b = io.BytesIO()
        PIL.fromarray(a).save(b, "PNG", compress_level=level)
        assert np.array_equal(decode(pkg, b.getvalue()), a)
    g16 = (np.arange(40 * 30, dtype=np.uint16).reshape(30, 40) * 50)
    b = io.BytesIO()
    PIL.fromarray(g16).save(b, "PNG")
    got = decode(pkg, b.getvalue())
    assert np.array_equal(got[..., 0], (g16 >> 8).astype(np.uint8)) and np.array_equal(got[..., 0], got[..., 2])


@pytest.mark.parametrize("fmt,mode", [("BMP", "RGB"), ("PPM", "RGB"), ("PPM", "L")])
def test_bmp_and_pnm_match_pillow(pkg, fmt, mode):
    im = PIL.fromarray(picture(53, 31, 9)).convert(mode)
    b = io.BytesIO()
    im.save(b, fmt)
    assert np.array_equal(decode(pkg, b.getvalue()), np.asarray(im.convert("RGB")))


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=0), dict(quality=75, subsampling=2), dict(quality=60, subsampling=1), dict(quality=95, subsampling=0, restart_marker_blocks=3),
                                dict(quality=85, subsampling=2, restart_marker_rows=1)])
@pytest.mark.parametrize("w,h,grey", [(64, 48, False), (37, 29, False), (50, 50, True)])
def test_baseline_jpeg_is_close_to_pillow(pkg, kw, w, h, grey):
    a = picture(w, h, w + h)
    im = PIL.fromarray(a).convert("L" if grey else "RGB")
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    got = decode(pkg, b.getvalue()).astype(np.int32)
    want = np.asarray(PIL.open(io.BytesIO(b.getvalue())).convert("RGB")).astype(np.int32)
    assert got.shape == want.shape
    d = np.abs(got - want)
    sub = kw.get("subsampling", 0)
    # same coefficients, two conforming reconstructions: the inverse transform (float here, fixed point there) and the rounding of the colour conversion
    # leave a unit or two; subsampled chroma goes through the same triangle filter as libjpeg's
    assert d.mean() <= 0.6, (d.mean(), d.max())
    assert d.max() <= 4, (d.mean(), d.max())
    # and against the picture itself it is as good a reconstruction as Pillow's
    assert np.abs(got - np.asarray(im.convert("RGB")).astype(np.int32)).mean() <= np.abs(want - np.asarray(im.convert("RGB")).astype(np.int32)).mean() + 1.0


def test_refusals(pkg):
    a = PIL.fromarray(picture(40, 40))
    b = io.BytesIO()
    a.save(b, "JPEG", progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        decode(pkg, b.getvalue())
    b = io.BytesIO()
    a.save(b, "PNG")
    raw = b.getvalue()
    with pytest.raises(RuntimeError):
        decode(pkg, raw[: len(raw) // 2])          # truncated
    with pytest.raises(RuntimeError, match="unknown format"):
        decode(pkg, b"GIF89a" + bytes(64))
    with pytest.raises(RuntimeError):
        decode(pkg, raw[:8] + bytes(100))


def test_decoders_survive_mutated_files_under_sanitizers(tmp_path):
    """The bytes come from requests.  host/image_decode.cc built with AddressSanitizer + UBSan; every sample file and 400 seeded mutations of each (bit flips,
    truncations, garbage runs, 0xff runs over length fields, insertions, zero runs) must come back as an image or as a refusal - no out-of-bounds access, no
    overflow, no crash."""
    import os
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "image_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(root, "cortex.llamacpp_amd", "host"),
                    os.path.join(root, "tests", "host", "image_fuzz.cc"), os.path.join(root, "cortex.llamacpp_amd", "host", "image_decode.cc"), "-o", exe], check=True)
    files = []
    a = picture(61, 47, 3)
    for name, fmt, mode, kw in [("rgb.png", "PNG", "RGB", {}), ("pal.png", "PNG", "P", {}), ("la.png", "PNG", "LA", {}), ("bit.png", "PNG", "1", {}), ("stored.png", "PNG", "RGB", {"compress_level": 0}),
                                ("q90.jpg", "JPEG", "RGB", {"quality": 90, "subsampling": 0}), ("q70_420.jpg", "JPEG", "RGB", {"quality": 70, "subsampling": 2}),
                                ("grey.jpg", "JPEG", "L", {"quality": 80}), ("rst.jpg", "JPEG", "RGB", {"quality": 85, "restart_marker_blocks": 2}),
                                ("refuse_progressive.jpg", "JPEG", "RGB", {"progressive": True}), ("x.bmp", "BMP", "RGB", {}), ("x.ppm", "PPM", "RGB", {}), ("x.pgm", "PPM", "L", {})]:
        p = str(tmp_path / name)
        PIL.fromarray(a).convert(mode).save(p, fmt, **kw)
        files.append(p)
    r = subprocess.run([exe, "400"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "mutations decoded" in r.stdout
