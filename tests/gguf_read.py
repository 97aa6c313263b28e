"""Minimal GGUF v3 reader for tests: metadata scalars / strings and raw tensor bytes (independent of the package's C++ reader and of the oracle's)."""
from __future__ import annotations

import struct

import numpy as np

_SC = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def read_gguf(path: str):
    """-> (kv: dict, tensors: dict name -> (ne tuple, ggml type, raw uint8 array))"""
    buf = open(path, "rb").read()
    off = 0

    def take(fmt):
        nonlocal off
        v = struct.unpack_from(fmt, buf, off)
        off += struct.calcsize(fmt)
        return v[0] if len(v) == 1 else v

    def string():
        nonlocal off
        n = take("<Q")
        s = buf[off:off + n].decode("utf-8", "replace")
        off += n
        return s

    def value(t):
        if t == 8:
            return string()
        if t == 9:
            et, n = take("<I"), take("<Q")
            return [value(et) for _ in range(n)]
        return take(_SC[t])

    magic, version, n_tensors, n_kv = take("<I"), take("<I"), take("<Q"), take("<Q")
    assert magic == 0x46554747 and version == 3
    kv = {}
    for _ in range(n_kv):
        k = string()
        t = take("<I")
        kv[k] = value(t)
    infos = []
    for _ in range(n_tensors):
        name = string()
        nd = take("<I")
        ne = tuple(take("<Q") for _ in range(nd))
        t, o = take("<I"), take("<Q")
        infos.append((name, ne, t, o))
    al = kv.get("general.alignment", 32)
    base = (off + al - 1) // al * al
    ends = sorted(o for _, _, _, o in infos) + [len(buf) - base]
    tensors = {}
    for name, ne, t, o in infos:
        nxt = min(e for e in ends if e > o)
        tensors[name] = (ne, t, np.frombuffer(buf, np.uint8, nxt - o, base + o))
    return kv, tensors
