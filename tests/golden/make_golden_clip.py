#!/usr/bin/env python3
"""Generates tests/golden/clip_v1.npz - committed vectors for the LLaVA image path.

Provenance.  clip.cpp (llama.cpp examples/llava) is not in /root/reference, so this path is as unpinned as the rest of the oracle; what the file pins is DRIFT and
STRUCTURE: the vectors are computed HERE by a numpy restatement that shares no code with oracle/oq_clip.c or the HIP kernels -
  * preprocessing (LLaVA-1.5: pad to a square with (122, 116, 104), half-pixel bilinear resample, round to a byte, normalise) in float32, element for element
    the expression of the CPU path: the C oracle and the device path must reproduce it bit for bit;
  * the encoder in float64 over the values the CPU path's operands hold: activations rounded to f16 in front of every f16 weight matrix (ggml's f16 vec_dot),
    attention in the f32 values, quick-GELU / GELU through the f16-table semantics (y = f16(f(f16(x)))), LayerNorm with biased variance.
The projector file is a pure function of (config, seed): gguf_synth.write_synthetic_clip("tiny-clip").
usage: python tests/golden/make_golden_clip.py"""
import os
import struct
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def read_gguf(path):
    """{name: ndarray} + {key: value} of a GGUF v3 file with f32 / f16 tensors (enough for a projector file)"""
    raw = open(path, "rb").read()
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, raw, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal pos
        n = rd("Q")
        s = raw[pos:pos + n].decode()
        pos += n
        return s
    magic, ver, n_t, n_kv = rd("IIQQ")
    assert magic == 0x46554747 and ver == 3
    scal = {0: "B", 1: "b", 2: "H", 3: "h", 4: "I", 5: "i", 6: "f", 7: "?", 10: "Q", 11: "q", 12: "d"}
    kv = {}
    for _ in range(n_kv):
        k = rstr()
        t = rd("I")
        if t == 8:
            kv[k] = rstr()
        elif t == 9:
            et, n = rd("I"), rd("Q")
            kv[k] = [rstr() if et == 8 else rd(scal[et]) for _ in range(n)]
        else:
            kv[k] = rd(scal[t])
    infos = []
    for _ in range(n_t):
        name = rstr()
        nd = rd("I")
        ne = [rd("Q") for _ in range(nd)]
        ty, off = rd("I"), rd("Q")
        infos.append((name, ne, ty, off))
    al = kv.get("general.alignment", 32)
    data0 = (pos + al - 1) // al * al
    out = {}
    for name, ne, ty, off in infos:
        n = int(np.prod(ne))
        dt = np.float32 if ty == 0 else np.float16
        out[name] = np.frombuffer(raw, dt, n, data0 + off).reshape(ne[::-1]).copy()     # numpy order: slowest dimension first
    return out, kv


def preprocess(rgb, S, mean, std):
    ny, nx, _ = rgb.shape
    if nx != ny:
        L = max(nx, ny)
        sq = np.empty((L, L, 3), np.uint8)
        sq[:] = np.array([122, 116, 104], np.uint8)
        sq[:ny, :nx] = rgb
        rgb = sq
    n = rgb.shape[0]
    f32 = np.float32
    scale = f32(n) / f32(S)
    c = (np.arange(S, dtype=f32) + f32(0.5)) * scale - f32(0.5)
    i0 = np.maximum(0, np.floor(c).astype(np.int32))
    i1 = np.minimum(i0 + 1, n - 1)
    d = c - i0.astype(f32)
    src = rgb.astype(f32)
    out = np.empty((3, S, S), f32)
    for k in range(3):
        p = src[:, :, k]
        v00, v01 = p[np.ix_(i0, i0)], p[np.ix_(i0, i1)]
        v10, v11 = p[np.ix_(i1, i0)], p[np.ix_(i1, i1)]
        dx, dy = d[None, :], d[:, None]
        v0 = v00 * (f32(1) - dx) + v01 * dx
        v1 = v10 * (f32(1) - dx) + v11 * dx
        v = v0 * (f32(1) - dy) + v1 * dy
        r = np.floor(np.abs(v) + f32(0.5)) * np.sign(v)                  # roundf: half away from zero
        v2 = np.clip(r, 0, 255).astype(np.uint8)
        out[k] = ((v2.astype(f32) / f32(255.0)) - f32(mean[k])) / f32(std[k])
    return out


def h16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float64)


def gelu_tab(x):
    x32 = x.astype(np.float32)
    xh = x32.astype(np.float16).astype(np.float32)
    g = np.float32(0.5) * xh * (np.float32(1) + np.tanh(np.float32(0.79788456080286535587989211986876) * xh * (np.float32(1) + np.float32(0.044715) * xh * xh)))
    y = g.astype(np.float16).astype(np.float64)
    return np.where(x32 <= -10, 0.0, np.where(x32 >= 10, x32.astype(np.float64), y))


def gelu_quick_tab(x):
    xh = x.astype(np.float32).astype(np.float16).astype(np.float32)
    g = xh * (np.float32(1) / (np.float32(1) + np.exp(np.float32(-1.702) * xh)))
    return g.astype(np.float16).astype(np.float64)


def encode(t, kv, img):
    S, P = kv["clip.vision.image_size"], kv["clip.vision.patch_size"]
    E, H, NL = kv["clip.vision.embedding_length"], kv["clip.vision.attention.head_count"], kv["clip.vision.block_count"]
    eps, use_gelu = kv["clip.vision.attention.layer_norm_epsilon"], kv.get("clip.use_gelu", False)
    G, D = S // P, E // H

    def lin(x, w, b):                                                    # f16 weights [out][in]: the activation row goes through f16
        return h16(x) @ t[w].astype(np.float64).T + t[b].astype(np.float64)

    def ln(x, w, b):
        mu = x.mean(-1, keepdims=True)
        xc = x - mu
        var = (xc * xc).mean(-1, keepdims=True)
        return xc / np.sqrt(var + eps) * t[w].astype(np.float64) + t[b].astype(np.float64)
    patches = img.reshape(3, G, P, G, P).transpose(1, 3, 0, 2, 4).reshape(G * G, 3 * P * P).astype(np.float64)
    wpe = t["v.patch_embd.weight"].reshape(E, 3 * P * P).astype(np.float64)
    emb = np.concatenate([t["v.class_embd"].astype(np.float64)[None, :], h16(patches) @ wpe.T], 0)
    emb = emb + t["v.position_embd.weight"].astype(np.float64)
    emb = ln(emb, "v.pre_ln.weight", "v.pre_ln.bias")
    for il in range(NL - 1):                                             # clip.cpp: block_count - 1 blocks feed a LLaVA projector (the file's last block is unused)
        p = f"v.blk.{il}."
        cur = ln(emb, p + "ln1.weight", p + "ln1.bias")
        q = lin(cur, p + "attn_q.weight", p + "attn_q.bias") * (1.0 / np.sqrt(np.float32(D)))
        k = lin(cur, p + "attn_k.weight", p + "attn_k.bias")
        v = lin(cur, p + "attn_v.weight", p + "attn_v.bias")
        T = emb.shape[0]
        qh, kh, vh = (a.astype(np.float32).astype(np.float64).reshape(T, H, D).transpose(1, 0, 2) for a in (q, k, v))
        s = qh @ kh.transpose(0, 2, 1)
        s = np.exp(s - s.max(-1, keepdims=True))
        s = s / s.sum(-1, keepdims=True)
        att = (s @ vh).transpose(1, 0, 2).reshape(T, E)
        emb = emb + lin(att, p + "attn_out.weight", p + "attn_out.bias")
        cur = ln(emb, p + "ln2.weight", p + "ln2.bias")
        ff = lin(cur, p + "ffn_down.weight", p + "ffn_down.bias")        # (the converter's names: "down" is the first projection)
        ff = gelu_tab(ff) if use_gelu else gelu_quick_tab(ff)
        emb = emb + lin(ff, p + "ffn_up.weight", p + "ffn_up.bias")
    h1 = gelu_tab(lin(emb[1:], "mm.0.weight", "mm.0.bias"))
    return lin(h1, "mm.2.weight", "mm.2.bias")


def main():
    import __graft_entry__ as ge
    gs = ge.load_pkg().gguf_synth
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for cfg in ("tiny-clip", "tiny-clip-gelu"):
            path = os.path.join(d, cfg + ".gguf")
            gs.write_synthetic_clip(path, cfg)
            t, kv = read_gguf(path)
            rng = np.random.default_rng(2025)
            for name, (w, h) in (("wide", (70, 40)), ("tall", (31, 90)), ("square", (56, 56)), ("small", (7, 5))):
                rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
                img = preprocess(rgb, kv["clip.vision.image_size"], kv["clip.vision.image_mean"], kv["clip.vision.image_std"])
                out[f"{cfg}.{name}.rgb"] = rgb
                out[f"{cfg}.{name}.img"] = img
                if name in ("wide", "small"):
                    out[f"{cfg}.{name}.emb"] = encode(t, kv, img).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "clip_v1.npz"), **out)
    print("wrote clip_v1.npz:", {k: v.shape for k, v in out.items() if k.endswith(".emb")})


if __name__ == "__main__":
    main()
