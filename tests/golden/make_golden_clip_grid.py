#!/usr/bin/env python3
"""Generates tests/golden/clip_grid_v1.npz - committed vectors for the LLaVA-1.6 image grid.

Provenance: as make_golden_clip.py - clip.cpp / llava.cpp are not in /root/reference, so the file pins DRIFT and STRUCTURE with a numpy restatement that shares
no code with oracle/oq_clip.c or the product:
  * bicubic_resize in float32, the expression element for element (source index truncated, neighbours clamped, the cubic's coefficients formed in double and
    rounded to float, the polynomial in float, rows first then the column, roundf + clamp to a byte);
  * select_best_resolution (most kept pixels, then least wasted canvas), resize_and_pad_image (aspect-preserving fit, centred on black), the S x S tiles
    row-major, the overview = the whole picture resized to S x S; every image normalised;
  * the row order of the picture's embedding: the overview's rows, then the tiles' rows in the canvas' row-major order - held as an index map
    (row r of the result = row order[r] of the tile-by-tile concatenation), so the encoder (pinned by clip_v1.npz) stays out of this file.
The C oracle and the device path must reproduce the images bit for bit and the order exactly.
usage: python tests/golden/make_golden_clip_grid.py"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

f32, f64 = np.float32, np.float64


def cubic(p0, p1, p2, p3, t):
    d0, d2, d3 = p0 - p1, p2 - p1, p3 - p1                                   # float32
    a1 = (f64(-1.0) / 3 * d0.astype(f64) + d2.astype(f64) - f64(1.0) / 6 * d3.astype(f64)).astype(f32)
    a2 = (f64(1.0) / 2 * d0.astype(f64) + f64(1.0) / 2 * d2.astype(f64)).astype(f32)
    a3 = (f64(-1.0) / 6 * d0.astype(f64) - f64(1.0) / 2 * d2.astype(f64) + f64(1.0) / 6 * d3.astype(f64)).astype(f32)
    return p1 + a1 * t + a2 * t * t + a3 * t * t * t


def bicubic(rgb, tw, th):
    ny, nx, _ = rgb.shape
    tx, ty = f32(nx) / f32(tw), f32(ny) / f32(th)
    fx, fy = tx * np.arange(tw, dtype=f32), ty * np.arange(th, dtype=f32)
    x, y = fx.astype(np.int32), fy.astype(np.int32)
    dx, dy = (fx - x.astype(f32))[None, :, None], (fy - y.astype(f32))[:, None, None]
    src = rgb.astype(f32)
    cols = []
    for jj in range(4):
        rows = src[np.clip(y - 1 + jj, 0, ny - 1)]                         # [th][nx][3]
        p = [rows[:, np.clip(x + o, 0, nx - 1)] for o in (-1, 0, 1, 2)]     # [th][tw][3]
        cols.append(cubic(p[0], p[1], p[2], p[3], dx))
    v = cubic(cols[0], cols[1], cols[2], cols[3], dy)
    r = np.floor(np.abs(v) + f32(0.5)) * np.sign(v)                          # roundf
    return np.clip(r, 0, 255).astype(np.uint8)


def best_canvas(ow, oh, pins):
    best, max_eff, min_waste = None, 0, None
    for w, h in pins:
        scale = min(f32(w) / f32(ow), f32(h) / f32(oh))
        dw, dh = int(f32(ow) * scale), int(f32(oh) * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and (min_waste is None or waste < min_waste)):
            best, max_eff, min_waste = (w, h), eff, waste
    return best


def fit(rgb, tw, th):
    ny, nx, _ = rgb.shape
    sw, sh = f32(tw) / f32(nx), f32(th) / f32(ny)
    if sw < sh:
        nw, nh = tw, min(int(math.ceil(float(f32(ny) * sw))), th)
    else:
        nh, nw = th, min(int(math.ceil(float(f32(nx) * sh))), tw)
    rs = bicubic(rgb, nw, nh)
    canvas = np.zeros((th, tw, 3), np.uint8)
    ox, oy = (tw - nw) // 2, (th - nh) // 2
    canvas[oy:oy + nh, ox:ox + nw] = rs
    return canvas


def planar(tile, mean, std):
    return np.stack([((tile[:, :, k].astype(f32) / f32(255.0)) - f32(mean[k])) / f32(std[k]) for k in range(3)], 0)


def preprocess_all(rgb, S, pins, mean, std, spatial_unpad=True):
    imgs = [planar(bicubic(rgb, S, S), mean, std)]
    if not spatial_unpad:
        return np.stack(imgs), 0, 0
    tw, th = best_canvas(rgb.shape[1], rgb.shape[0], pins)
    canvas = fit(rgb, tw, th)
    gw, gh = tw // S, th // S
    for gy in range(gh):
        for gx in range(gw):
            imgs.append(planar(canvas[gy * S:(gy + 1) * S, gx * S:(gx + 1) * S], mean, std))
    return np.stack(imgs), gw, gh


def row_order(G, gw, gh):
    """row r of a picture's embedding = row order[r] of [overview ; tile 0 ; tile 1 ; ...] (each G x G rows, tiles row-major)"""
    NP = G * G
    order = list(range(NP))
    for Y in range(gh * G):
        for X in range(gw * G):
            tile = (Y // G) * gw + (X // G)
            order.append(NP * (1 + tile) + (Y % G) * G + (X % G))
    return np.array(order, np.int32)


CASES = (("wide", (150, 60)), ("tall", (40, 170)), ("square", (100, 100)), ("small", (9, 6)), ("strip", (400, 90)), ("exact", (112, 56)), ("photo", (301, 211)))


def main():
    import __graft_entry__ as ge
    from make_golden_clip import read_gguf
    import tempfile
    gs = ge.load_pkg().gguf_synth
    out = {}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "g.gguf")
        gs.write_synthetic_clip(path, "tiny-clip-grid")
        _, kv = read_gguf(path)
    S, P = kv["clip.vision.image_size"], kv["clip.vision.patch_size"]
    pp = kv["clip.vision.image_grid_pinpoints"]
    pins = [(pp[i], pp[i + 1]) for i in range(0, len(pp), 2)]
    rng = np.random.default_rng(1606)
    for name, (w, h) in CASES:
        y, x = np.mgrid[0:h, 0:w]
        base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x * 3 + y * 5) % 256)], -1).astype(np.int32)
        rgb = np.clip(base + rng.integers(-60, 61, base.shape), 0, 255).astype(np.uint8)
        imgs, gw, gh = preprocess_all(rgb, S, pins, kv["clip.vision.image_mean"], kv["clip.vision.image_std"])
        out[f"{name}.rgb"] = rgb
        out[f"{name}.imgs"] = imgs
        out[f"{name}.grid"] = np.array([gw, gh], np.int32)
        out[f"{name}.order"] = row_order(S // P, gw, gh)
    np.savez_compressed(os.path.join(HERE, "clip_grid_v1.npz"), **out)
    print("wrote clip_grid_v1.npz:", {k: tuple(int(x) for x in v) for k, v in out.items() if k.endswith(".grid")})


if __name__ == "__main__":
    sys.path.insert(0, HERE)
    main()
