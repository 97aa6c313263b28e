#!/usr/bin/env python3
"""Time of the image side of a LLaVA request at full size (ViT-L/14-336 + projector, synthetic weights): one encode, and a LLaVA-1.6 picture (five encodes +
preprocessing).  usage: python3 tools/clip_time.py"""
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge


def main():
    pkg = ge.load_pkg()
    pkg.Backend()
    d = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (700, 1000, 3)).astype(np.uint8)
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(rgb).save(b, "JPEG", quality=90)
    jpg = b.getvalue()
    for cfg in ("clip-vit-l-336", "clip-vit-l-336-grid"):
        path = os.path.join(d, cfg + ".gguf")
        pkg.gguf_synth.write_synthetic_clip(path, cfg)
        c = pkg.Clip(path)
        img = c.preprocess(rgb)
        c.encode(img)
        t0 = time.perf_counter()
        for _ in range(5):
            c.encode(img)
        t_enc = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        dec = c.load_image(jpg)
        t_dec = time.perf_counter() - t0
        t0 = time.perf_counter()
        c.preprocess_grid(dec)
        t_pre = time.perf_counter() - t0
        c.embed_bytes(jpg)
        t0 = time.perf_counter()
        rows = c.embed_bytes(jpg)
        t_all = time.perf_counter() - t0
        print(f"{cfg}: encode {t_enc * 1e3:.1f} ms/image; 1000x700 JPEG decode {t_dec * 1e3:.1f} ms, preprocess {t_pre * 1e3:.1f} ms, bytes -> {len(rows)} rows {t_all * 1e3:.1f} ms")
        c.close()


if __name__ == "__main__":
    main()
