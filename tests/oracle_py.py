"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 8, 12, 13, 14, 15
Q5_0, Q2_K, Q3_K, IQ4_NL = 6, 10, 11, 20


def threads(cap: int = 32) -> int:
    """OpenMP threads for the oracle's heavier runs: the cores this process may use, at most `cap` (the restatement is parallel over weight rows / heads: the
    thread count never changes a bit of its results)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def build_oracle(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oq_quants.c", "oq_ops.c", "oq_llama.c", "oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        L = _lib
        vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
        L.oq_block_elems.restype = i32
        L.oq_block_bytes.restype = sz
        L.oq_row_bytes.restype = sz
        L.oq_row_bytes.argtypes = [i32, i64]
        L.oq_fp16_to_fp32.restype = f32
        L.oq_fp16_to_fp32.argtypes = [C.c_uint16]
        L.oq_fp32_to_fp16.restype = C.c_uint16
        L.oq_fp32_to_fp16.argtypes = [f32]
        L.oq_dequantize_row.argtypes = [i32, vp, vp, i64]
        L.oq_quantize_row.argtypes = [i32, vp, vp, i64]
        L.oq_vec_dot_type.restype = i32
        L.oq_vec_dot.restype = f32
        L.oq_vec_dot.argtypes = [i32, i64, vp, vp]
        L.oq_vec_dot_int_partials.argtypes = [i32, i64, vp, vp, vp, vp]
        L.oq_mul_mat.argtypes = [i32, vp, i64, i64, vp, i64, vp, i32]
        L.oq_rms_norm.argtypes = [vp, vp, i64, f32]
        L.oq_silu_f32.argtypes = [vp, vp, i64]
        L.oq_soft_max.argtypes = [vp, vp, vp, i64, f32]
        L.oq_moe_route.argtypes = [vp, i32, i32, vp, vp, vp]
        L.oq_rope_norm.argtypes = [vp, i32, i32, i32, C.c_int32, f32, f32, vp]
        L.oq_rope_neox.argtypes = [vp, i32, i32, i32, C.c_int32, f32, f32, vp]
        L.oq_rope_ext.argtypes = [vp, i32, i32, i32, C.c_int32, f32, f32, vp, i32, vp]
        L.oq_yarn_corr_dims.argtypes = [i32, i32, f32, f32, f32, vp, vp]
        L.oq_get_rows.argtypes = [i32, vp, i64, vp, i64, vp]
        L.oq_flash_attn_ext.argtypes = [vp, i32, i32, i32, i32, i32, vp, sz, sz, i32, vp, sz, sz, vp, i32, f32, vp]
        L.oq_model_load.restype = vp
        L.oq_model_load.argtypes = [C.c_char_p]
        L.oq_model_free.argtypes = [vp]
        for fn in ("oq_model_n_vocab", "oq_model_n_embd", "oq_model_n_layer"):
            getattr(L, fn).restype = i32
            getattr(L, fn).argtypes = [vp]
        L.oq_ctx_new.restype = vp
        L.oq_ctx_new.argtypes = [vp, i32, i32, i32, i32, i32]
        L.oq_ctx_free.argtypes = [vp]
        L.oq_decode.restype = i32
        L.oq_decode.argtypes = [vp, vp, vp, vp, vp, i32, vp]
        L.oq_clip_load.restype = vp
        L.oq_clip_load.argtypes = [C.c_char_p]
        L.oq_clip_free.argtypes = [vp]
        for nm in ("oq_clip_image_size", "oq_clip_n_patches", "oq_clip_n_mmproj_embd"):
            getattr(L, nm).restype = i32
            getattr(L, nm).argtypes = [vp]
        L.oq_clip_preprocess.argtypes = [vp, vp, i32, i32, vp]
        L.oq_clip_max_image_rows.restype = i32
        L.oq_clip_max_image_rows.argtypes = [vp]
        L.oq_clip_preprocess_all.restype = i32
        L.oq_clip_preprocess_all.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp]
        L.oq_clip_embed.restype = i32
        L.oq_clip_embed.argtypes = [vp, vp, i32, i32, vp, i32, i32]
        L.oq_clip_encode.restype = i32
        L.oq_clip_encode.argtypes = [vp, vp, vp, i32]
        L.oq_decode_embd.restype = i32
        L.oq_decode_embd.argtypes = [vp, vp, vp, vp, vp, i32, vp]
        L.oq_kv_clear.argtypes = [vp]
        L.oq_kv_seq_rm.restype = i32
        L.oq_kv_seq_rm.argtypes = [vp, i32, i32, i32]
        L.oq_kv_seq_cp.argtypes = [vp, i32, i32, i32, i32]
        L.oq_kv_seq_add.argtypes = [vp, i32, i32, i32, i32]
        L.oq_debug_layer_out.restype = C.POINTER(C.c_float)
        L.oq_debug_layer_out.argtypes = [vp, i32]
        L.oq_set_assoc_variant.argtypes = [i32]
        L.oq_moe_record_start.argtypes = [C.c_size_t]
        L.oq_moe_record_get.argtypes = [vp, C.c_size_t]
        L.oq_moe_record_get.restype = C.c_size_t
        L.oq_set_fa_v_acc_f32.argtypes = [i32]
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def row_bytes(t: int, n: int) -> int:
    return lib().oq_row_bytes(t, n)


def dequantize(t: int, raw: np.ndarray, n: int) -> np.ndarray:
    raw = np.ascontiguousarray(raw.view(np.uint8).reshape(-1))
    out = np.empty(n, dtype=np.float32)
    lib().oq_dequantize_row(t, _p(raw), _p(out), n)
    return out


def quantize(t: int, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    out = np.zeros(row_bytes(t, x.size), dtype=np.uint8)
    lib().oq_quantize_row(t, _p(x), _p(out), x.size)
    return out


def vec_dot_type(t: int) -> int:
    return lib().oq_vec_dot_type(t)


def vec_dot(t: int, w_row: np.ndarray, act_q: np.ndarray, n: int) -> float:
    w_row = np.ascontiguousarray(w_row.view(np.uint8))
    act_q = np.ascontiguousarray(act_q.view(np.uint8))
    return float(lib().oq_vec_dot(t, n, _p(w_row), _p(act_q)))


def vec_dot_int_partials(t: int, w_row: np.ndarray, act_q: np.ndarray, n: int):
    nb = n // (32 if t in (Q8_0, Q5_0, Q4_0, IQ4_NL) else 256)
    isum = np.zeros(nb, dtype=np.int32)
    msum = np.zeros(nb, dtype=np.int32)
    w_row = np.ascontiguousarray(w_row.view(np.uint8))
    act_q = np.ascontiguousarray(act_q.view(np.uint8))
    lib().oq_vec_dot_int_partials(t, n, _p(w_row), _p(act_q), _p(isum), _p(msum))
    return isum, msum


def mul_mat(t: int, W: np.ndarray, N: int, K: int, x: np.ndarray, nth: int = 4) -> np.ndarray:
    """W raw bytes [N rows of K]; x f32 [T][K]; returns f32 [T][N]."""
    W = np.ascontiguousarray(W.view(np.uint8).reshape(-1))
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, K)
    T = x.shape[0]
    y = np.empty((T, N), dtype=np.float32)
    lib().oq_mul_mat(t, _p(W), N, K, _p(x), T, _p(y), nth)
    return y


def rms_norm(x: np.ndarray, eps: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().oq_rms_norm(_p(x), _p(y), x.size, eps)
    return y


def silu(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().oq_silu_f32(_p(x), _p(y), x.size)
    return y


def soft_max(x: np.ndarray, mask, scale: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    m = None if mask is None else np.ascontiguousarray(mask, dtype=np.float32)
    lib().oq_soft_max(_p(x), None if m is None else _p(m), _p(y), x.size, scale)
    return y


def moe_route(logits: np.ndarray, k: int):
    """Top-k expert selection of one token: (ids [k] int32, weights [k] f32)."""
    x = np.ascontiguousarray(logits, dtype=np.float32)
    probs = np.empty_like(x); ids = np.zeros(k, np.int32); w = np.zeros(k, np.float32)
    lib().oq_moe_route(_p(x), x.size, k, _p(probs), _p(ids), _p(w))
    return ids, w


def rope(x: np.ndarray, n_head: int, head_dim: int, pos: int, base: float, neox: bool = False,
         n_rot: int | None = None, freq_scale: float = 1.0, freq_factors=None) -> np.ndarray:
    y = np.array(x, dtype=np.float32, copy=True).reshape(n_head * head_dim)
    ff = None if freq_factors is None else np.ascontiguousarray(freq_factors, dtype=np.float32)
    fn = lib().oq_rope_neox if neox else lib().oq_rope_norm
    fn(_p(y), n_head, head_dim, n_rot or head_dim, pos, base, freq_scale, None if ff is None else _p(ff))
    return y.reshape(n_head, head_dim)


def yarn_corr_dims(n_rot: int, n_ctx_orig: int, base: float, beta_fast: float = 32.0, beta_slow: float = 1.0) -> tuple[float, float]:
    lo, hi = C.c_float(), C.c_float()
    lib().oq_yarn_corr_dims(n_rot, n_ctx_orig, base, beta_fast, beta_slow, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def rope_yarn(x: np.ndarray, n_head: int, head_dim: int, pos: int, base: float, freq_scale: float, ext_factor: float, attn_factor: float,
              corr_lo: float, corr_hi: float, neox: bool = False, n_rot: int | None = None) -> np.ndarray:
    y = np.array(x, dtype=np.float32, copy=True).reshape(n_head * head_dim)
    par = (C.c_float * 4)(ext_factor, attn_factor, corr_lo, corr_hi)
    lib().oq_rope_ext(_p(y), n_head, head_dim, n_rot or head_dim, pos, base, freq_scale, None, int(neox), par)
    return y.reshape(n_head, head_dim)


def flash_attn(q: np.ndarray, n_head: int, n_head_kv: int, hd: int, type_k: int, k_cache: np.ndarray,
               type_v: int, v_cache: np.ndarray, cells: np.ndarray, scale: float) -> np.ndarray:
    """k_cache / v_cache: raw bytes [n_cells_total][n_head_kv * row_bytes(type, hd)]."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    k_cache = np.ascontiguousarray(k_cache.view(np.uint8))
    v_cache = np.ascontiguousarray(v_cache.view(np.uint8))
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    kh, vh = row_bytes(type_k, hd), row_bytes(type_v, hd)
    out = np.empty((n_head, hd), dtype=np.float32)
    lib().oq_flash_attn_ext(_p(q), n_head, n_head_kv, hd, hd, type_k, _p(k_cache), kh * n_head_kv, kh,
                            type_v, _p(v_cache), vh * n_head_kv, vh, _p(cells), cells.size, scale, _p(out))
    return out


def moe_record_start(cap: int = 1 << 20):
    """Record the expert ids the oracle's router selects from now on (call order: decode call -> layer -> token -> rank)."""
    lib().oq_moe_record_start(cap)


def moe_record_get() -> np.ndarray:
    n = lib().oq_moe_record_get(None, 0)
    out = np.zeros(n, np.int32)
    lib().oq_moe_record_get(_p(out), n)
    return out


def set_assoc_variant(v: int):
    lib().oq_set_assoc_variant(v)


def set_fa_v_acc_f32(v: int):
    lib().oq_set_fa_v_acc_f32(v)


class OracleModel:
    def __init__(self, path: str):
        self.h = lib().oq_model_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path}")
        self.n_vocab = lib().oq_model_n_vocab(self.h)
        self.n_embd = lib().oq_model_n_embd(self.h)
        self.n_layer = lib().oq_model_n_layer(self.h)

    def close(self):
        if self.h:
            lib().oq_model_free(self.h)
            self.h = None


class OracleContext:
    def __init__(self, model: OracleModel, n_ctx: int, type_k: int = F16, type_v: int = F16,
                 flash_attn: bool = True, n_threads: int = 4):
        self.model = model
        self.h = lib().oq_ctx_new(model.h, n_ctx, type_k, type_v, int(flash_attn), n_threads)
        if not self.h:
            raise RuntimeError("oracle: bad context params")

    def decode(self, tokens, pos, seq=None, want_logits=None) -> np.ndarray:
        tokens = np.ascontiguousarray(tokens, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = tokens.size
        seq_a = None if seq is None else np.ascontiguousarray(seq, dtype=np.int32)
        want = None if want_logits is None else np.ascontiguousarray(want_logits, dtype=np.int8)
        n_out = 1 if want is None else int((want != 0).sum())
        out = np.empty((n_out, self.model.n_vocab), dtype=np.float32)
        rc = lib().oq_decode(self.h, _p(tokens), _p(pos), None if seq_a is None else _p(seq_a),
                             None if want is None else _p(want), n, _p(out))
        if rc != 0:
            raise RuntimeError(f"oracle decode rc={rc}")
        return out

    def decode_embd(self, embd, pos, seq=None, want_logits=None) -> np.ndarray:
        embd = np.ascontiguousarray(embd, dtype=np.float32)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        n = pos.size
        assert embd.shape == (n, self.model.n_embd)
        seq_a = None if seq is None else np.ascontiguousarray(seq, dtype=np.int32)
        want = None if want_logits is None else np.ascontiguousarray(want_logits, dtype=np.int8)
        n_out = 1 if want is None else int((want != 0).sum())
        out = np.empty((n_out, self.model.n_vocab), dtype=np.float32)
        rc = lib().oq_decode_embd(self.h, _p(embd), _p(pos), None if seq_a is None else _p(seq_a),
                                  None if want is None else _p(want), n, _p(out))
        if rc != 0:
            raise RuntimeError(f"oracle decode_embd rc={rc}")
        return out

    def layer_out(self, il: int, n_tokens: int) -> np.ndarray:
        p = lib().oq_debug_layer_out(self.h, il)
        return np.ctypeslib.as_array(p, shape=(n_tokens, self.model.n_embd)).copy()

    def kv_clear(self):
        lib().oq_kv_clear(self.h)

    def kv_seq_rm(self, seq, p0, p1):
        return bool(lib().oq_kv_seq_rm(self.h, seq, p0, p1))

    def kv_seq_cp(self, s, d, p0, p1):
        lib().oq_kv_seq_cp(self.h, s, d, p0, p1)

    def kv_seq_add(self, seq, p0, p1, delta):
        lib().oq_kv_seq_add(self.h, seq, p0, p1, delta)

    def close(self):
        if self.h:
            lib().oq_ctx_free(self.h)
            self.h = None


class OracleClip:
    """clip_model_load + clip_image_preprocess + clip_image_encode of the CPU restatement (oracle/oq_clip.c)."""

    def __init__(self, path: str):
        self.h = lib().oq_clip_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load projector file {path}")
        self.image_size = lib().oq_clip_image_size(self.h)
        self.n_patches = lib().oq_clip_n_patches(self.h)
        self.n_embd = lib().oq_clip_n_mmproj_embd(self.h)
        self.max_image_rows = lib().oq_clip_max_image_rows(self.h)

    def preprocess_all(self, rgb: np.ndarray):
        """LLaVA-1.6: ([n, 3, S, S], grid_w, grid_h)."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        ny, nx, ch = rgb.shape
        assert ch == 3
        cap = self.max_image_rows // self.n_patches
        out = np.empty((cap, 3, self.image_size, self.image_size), np.float32)
        gw, gh = C.c_int32(0), C.c_int32(0)
        n = lib().oq_clip_preprocess_all(self.h, _p(rgb), nx, ny, _p(out), cap, C.byref(gw), C.byref(gh))
        if n < 1:
            raise RuntimeError(f"oracle clip preprocess_all rc={n}")
        return out[:n], gw.value, gh.value

    def embed(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        ny, nx, ch = rgb.shape
        assert ch == 3
        out = np.empty((self.max_image_rows, self.n_embd), np.float32)
        n = lib().oq_clip_embed(self.h, _p(rgb), nx, ny, _p(out), self.max_image_rows, threads())
        if n < 1:
            raise RuntimeError(f"oracle clip embed rc={n}")
        return out[:n]

    def preprocess(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        ny, nx, ch = rgb.shape
        assert ch == 3
        out = np.empty((3, self.image_size, self.image_size), np.float32)
        lib().oq_clip_preprocess(self.h, _p(rgb), nx, ny, _p(out))
        return out

    def encode(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, dtype=np.float32)
        assert img.shape == (3, self.image_size, self.image_size)
        out = np.empty((self.n_patches, self.n_embd), np.float32)
        rc = lib().oq_clip_encode(self.h, _p(img), _p(out), threads())
        if rc != 0:
            raise RuntimeError(f"oracle clip encode rc={rc}")
        return out

    def close(self):
        if self.h:
            lib().oq_clip_free(self.h)
            self.h = None
