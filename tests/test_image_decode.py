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


def _png_from_zlib_stream(rgb: np.ndarray, idat: bytes) -> bytes:
    """a PNG around a hand-made IDAT stream (filter type 0 rows)"""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    h, w, _ = rgb.shape
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")


@pytest.mark.parametrize("flush", ["sync", "full"])
def test_png_with_flushed_streams_mixing_stored_and_huffman_blocks(pkg, flush):
    """An encoder that flushes (Z_SYNC_FLUSH / Z_FULL_FLUSH) ends a Huffman block and emits an EMPTY STORED block, byte-aligned, in the middle of the stream: the
    stored block's LEN / NLEN may already sit in the Huffman decoder's look-ahead when the end-of-block code is at most 5 bits long (without the
    hand-back in inflate_zlib about a quarter of these files are refused).  200 files per flush mode, plus streams that alternate level-0 (stored) and level-9 sections."""
    import zlib
    mode = zlib.Z_SYNC_FLUSH if flush == "sync" else zlib.Z_FULL_FLUSH
    bad = []
    for i in range(200):
        rng = np.random.default_rng(1000 + i)
        a = picture(64, 64, i)
        a[:48] = a[0, 0]                                     # a flat band: the first section is a few dozen long matches, a code with few symbols, a short end-of-block code
        raw = b"".join(b"\x00" + a[y].tobytes() for y in range(64))
        co = zlib.compressobj(9)
        cut = int(rng.integers(2000, 9000)) if i % 4 else int(rng.integers(1, 400))
        stream = co.compress(raw[:cut]) + co.flush(mode)
        if i % 3 == 0:                                       # a second flush further on, and a switch to stored blocks for the tail
            cut2 = cut + int(rng.integers(1, 3000))
            stream += co.compress(raw[cut:cut2]) + co.flush(mode)
            cut = cut2
        stream += co.compress(raw[cut:]) + co.flush()
        assert zlib.decompress(stream) == raw
        try:
            got = decode(pkg, _png_from_zlib_stream(a, stream))
        except RuntimeError as e:
            bad.append((i, str(e)))
            continue
        assert np.array_equal(got, a), i
    assert not bad, bad[:5]


@pytest.mark.parametrize("fmt,mode", [("BMP", "RGB"), ("PPM", "RGB"), ("PPM", "L")])
def test_bmp_and_pnm_match_pillow(pkg, fmt, mode):
    im = PIL.fromarray(picture(53, 31, 9)).convert(mode)
    b = io.BytesIO()
    im.save(b, fmt)
    assert np.array_equal(decode(pkg, b.getvalue()), np.asarray(im.convert("RGB")))


def adam7_png(a: np.ndarray, depth: int = 8, palette=None) -> bytes:
    """An interlaced PNG of `a` ([h][w] grey / palette indices or [h][w][3|4]) written here (Pillow reads Adam7 but does not write it): seven reduced pictures, rows
    filtered with a type that changes from row to row (none / sub / up / average / Paeth)"""
    import struct
    import zlib
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    ctype = 3 if palette is not None else {1: 0, 3: 2, 4: 6}[ch]
    bpp = max(1, ch * depth // 8)
    raw = b""
    for p, (x0, y0, dx, dy) in enumerate([(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for y in range(sub.shape[0]):
            row = sub[y].reshape(-1)
            if depth < 8:
                bits = "".join(format(int(v), f"0{depth}b") for v in row)
                bits += "0" * (-len(bits) % 8)
                line = np.frombuffer(int(bits, 2).to_bytes(len(bits) // 8, "big"), np.uint8).astype(np.int32)
            else:
                line = row.astype(np.int32)
            up = prev if prev is not None else np.zeros_like(line)
            left = np.concatenate([np.zeros(bpp, np.int32), line[:-bpp]]) if len(line) > bpp else np.zeros_like(line)
            upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if len(line) > bpp else np.zeros_like(line)
            ft = (y + p) % 5
            if ft == 0:
                f = line
            elif ft == 1:
                f = line - left
            elif ft == 2:
                f = line - up
            elif ft == 3:
                f = line - ((left + up) >> 1)
            else:
                pp = left + up - upleft
                pa, pb, pc = np.abs(pp - left), np.abs(pp - up), np.abs(pp - upleft)
                f = line - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
            raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = line

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1))
    if palette is not None:
        out += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    z = zlib.compress(raw, 6)
    return out + chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")


@pytest.mark.parametrize("w,h", [(33, 21), (8, 8), (1, 1), (3, 2), (5, 1), (2, 9), (64, 64)])
def test_interlaced_png_matches_pillow(pkg, w, h):
    """Adam7 (pictures narrower than a pass's offset have empty passes): RGB, RGBA, grey, a 4-bit palette, 1-bit grey"""
    a = picture(max(w, 2), max(h, 2), w * 13 + h)[:h, :w]
    rng = np.random.default_rng(w + h)
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
    cases = [adam7_png(a), adam7_png(np.dstack([a, a[:, :, :1]])), adam7_png(a[:, :, 0]), adam7_png(a[:, :, 1] >> 4, depth=4, palette=pal), adam7_png(a[:, :, 2] >> 7, depth=1)]
    for i, data in enumerate(cases):
        want = np.asarray(PIL.open(io.BytesIO(data)).convert("RGB"))
        assert np.array_equal(decode(pkg, data), want), (i, w, h)
    assert np.array_equal(decode(pkg, cases[0]), a)


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=0), dict(quality=75, subsampling=2), dict(quality=60, subsampling=1), dict(quality=95, subsampling=0, restart_marker_blocks=3),
                                dict(quality=85, subsampling=2, restart_marker_rows=1)])
@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("w,h,grey", [(64, 48, False), (37, 29, False), (50, 50, True)])
def test_jpeg_is_close_to_pillow(pkg, kw, w, h, grey, progressive):
    """sequential (one interleaved scan) and progressive (Pillow's script: DC first, per-component AC bands, successive-approximation refinements - the
    non-interleaved scans, end-of-band runs and refinement passes of annex G) files of the same picture"""
    kw = dict(kw, progressive=True) if progressive else kw
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


def test_progressive_jpeg_decodes_to_the_sequential_files_pixels(pkg):
    """the same quantised coefficients reach the decoder through ten scans instead of one: the pixels must be the sequential file's, bit for bit (a larger picture,
    4:2:0, with restart intervals; and a high-quality one, whose refinement scans carry many non-zero coefficients)"""
    for w, h, kw in [(203, 157, dict(quality=80, subsampling=2)), (96, 64, dict(quality=98, subsampling=0)), (131, 77, dict(quality=50, subsampling=1, restart_marker_rows=1)),
                     (64, 64, dict(quality=100, subsampling=0))]:
        im = PIL.fromarray(picture(w, h, w))
        a, b = io.BytesIO(), io.BytesIO()
        im.save(a, "JPEG", **kw)
        im.save(b, "JPEG", progressive=True, **kw)
        assert b"\xff\xc2" in b.getvalue() and b"\xff\xc2" not in a.getvalue()[:600]
        assert np.array_equal(decode(pkg, a.getvalue()), decode(pkg, b.getvalue())), (w, h, kw)
    g = PIL.fromarray(picture(70, 90, 5)).convert("L")
    a, b = io.BytesIO(), io.BytesIO()
    g.save(a, "JPEG", quality=85)
    g.save(b, "JPEG", quality=85, progressive=True)
    assert np.array_equal(decode(pkg, a.getvalue()), decode(pkg, b.getvalue()))


def test_refusals(pkg):
    a = PIL.fromarray(picture(40, 40))
    b = io.BytesIO()
    a.save(b, "JPEG")
    raw = b.getvalue()
    i = raw.index(b"\xff\xc0")
    with pytest.raises(RuntimeError, match="arithmetic"):
        decode(pkg, raw[:i] + b"\xff\xc9" + raw[i + 2:])          # (a frame header that announces arithmetic coding)
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
                                ("progressive.jpg", "JPEG", "RGB", {"progressive": True, "quality": 85, "subsampling": 2}),
                                ("progressive_rst.jpg", "JPEG", "RGB", {"progressive": True, "quality": 95, "subsampling": 0, "restart_marker_blocks": 3}), ("x.bmp", "BMP", "RGB", {}), ("x.ppm", "PPM", "RGB", {}), ("x.pgm", "PPM", "L", {})]:
        p = str(tmp_path / name)
        PIL.fromarray(a).convert(mode).save(p, fmt, **kw)
        files.append(p)
    for name, data in [("adam7.png", adam7_png(a)), ("adam7_pal.png", adam7_png(a[:, :, 0] >> 4, depth=4, palette=np.arange(48, dtype=np.uint8).reshape(16, 3)))]:
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        files.append(p)
    r = subprocess.run([exe, "400"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "mutations decoded" in r.stdout
