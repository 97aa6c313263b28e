"""ctypes binding of include/mi355_llama.h (same names, same argument meaning)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 8, 12, 13, 14, 15
Q2_K, Q3_K = 10, 11
Q5_0, IQ4_NL = 6, 20


class MI355Error(RuntimeError):
    pass


def lib_path() -> str:
    # MI355_LLAMA_LIB: an alternative build of the same library (tools/ experiments)
    return os.environ.get("MI355_LLAMA_LIB") or os.path.join(HERE, "lib", "libmi355_llama.so")


class ModelParams(C.Structure):
    _fields_ = [("n_gpu_layers", C.c_int32), ("main_gpu", C.c_int32), ("use_mmap", C.c_int32), ("use_mlock", C.c_int32),
                ("tp_rank", C.c_int32), ("tp_size", C.c_int32), ("prefill_planes", C.c_int32)]


class ContextParams(C.Structure):
    _fields_ = [("n_ctx", C.c_uint32), ("n_batch", C.c_uint32), ("n_ubatch", C.c_uint32), ("n_seq_max", C.c_uint32),
                ("type_k", C.c_int32), ("type_v", C.c_int32), ("flash_attn", C.c_int32), ("embeddings", C.c_int32),
                ("use_graphs", C.c_int32), ("logits_to_host", C.c_int32)]


class Batch(C.Structure):
    _fields_ = [("n_tokens", C.c_int32), ("token", C.POINTER(C.c_int32)), ("embd", C.POINTER(C.c_float)),
                ("pos", C.POINTER(C.c_int32)), ("n_seq_id", C.POINTER(C.c_int32)),
                ("seq_id", C.POINTER(C.POINTER(C.c_int32))), ("logits", C.POINTER(C.c_int8))]


# every symbol include/mi355_llama.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u32, _u64, _f32, _sz, _cp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t, C.c_char_p
ENGINE_CB = C.CFUNCTYPE(None, C.c_char_p, C.c_char_p, C.c_void_p)
SYMBOLS = {
    "mi355_backend_init": (C.c_int, []),
    "mi355_backend_free": (None, []),
    "mi355_device_count": (C.c_int, []),
    "mi355_last_error": (_cp, []),
    "mi355_time_us": (_i64, []),
    "mi355_print_system_info": (_cp, []),
    "mi355_model_default_params": (ModelParams, []),
    "mi355_model_load_from_file": (_vp, [_cp, ModelParams]),
    "mi355_model_free": (None, [_vp]),
    "mi355_model_n_vocab": (_i32, [_vp]),
    "mi355_model_n_embd": (_i32, [_vp]),
    "mi355_model_n_layer": (_i32, [_vp]),
    "mi355_model_n_head": (_i32, [_vp]),
    "mi355_model_n_head_kv": (_i32, [_vp]),
    "mi355_model_n_ctx_train": (_i32, [_vp]),
    "mi355_model_size": (_u64, [_vp]),
    "mi355_model_cpu_buffer": (_u64, [_vp]),
    "mi355_model_other_buffer": (_u64, [_vp]),
    "mi355_model_bytes_per_token": (_u64, [_vp]),
    "mi355_model_planes_bytes": (_u64, [_vp]),
    "mi355_debug_set_option": (C.c_int, [_cp, _i32]),
    "mi355_model_desc": (_cp, [_vp]),
    "mi355_model_meta_str": (C.c_int, [_vp, _cp, _cp, _sz]),
    "mi355_context_default_params": (ContextParams, []),
    "mi355_context_new": (_vp, [_vp, ContextParams]),
    "mi355_context_free": (None, [_vp]),
    "mi355_n_ctx": (_u32, [_vp]),
    "mi355_n_batch": (_u32, [_vp]),
    "mi355_n_ubatch": (_u32, [_vp]),
    "mi355_context_device_bytes": (_u64, [_vp]),
    "mi355_batch_init": (Batch, [_i32, _i32, _i32]),
    "mi355_batch_free": (None, [Batch]),
    "mi355_decode": (_i32, [_vp, Batch]),
    "mi355_greedy_steps": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp]),
    "mi355_get_logits_ith": (C.POINTER(C.c_float), [_vp, _i32]),
    "mi355_get_argmax_ith": (_i32, [_vp, _i32]),
    "mi355_get_topk_ith": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp]),
    "mi355_debug_mega_steps": (C.c_int64, [_vp]),
    "mi355_debug_engine_steps": (C.c_int64, [_vp]),
    "mi355_debug_fused_skipped_steps": (C.c_int64, [_vp]),
    "mi355_debug_qkv_attn_launches": (C.c_int64, [_vp]),
    "mi355_debug_qkv_attn_plan": (C.c_int64, [C.c_int32] * 10 + [C.POINTER(C.c_int32)]),
    "mi355_set_embeddings": (None, [_vp, _i32]),
    "mi355_get_embeddings_ith": (C.POINTER(C.c_float), [_vp, _i32]),
    "mi355_synchronize": (None, [_vp]),
    "mi355_kv_cache_clear": (None, [_vp]),
    "mi355_kv_cache_seq_rm": (_i32, [_vp, _i32, _i32, _i32]),
    "mi355_kv_cache_seq_cp": (None, [_vp, _i32, _i32, _i32, _i32]),
    "mi355_kv_cache_seq_add": (None, [_vp, _i32, _i32, _i32, _i32]),
    "mi355_kv_cache_used_cells": (_i32, [_vp]),
    "mi355_debug_enable_taps": (None, [_vp, _i32]),
    "mi355_debug_layer_out": (_i32, [_vp, _i32, _vp, _sz]),
    "mi355_op_quantize_act": (C.c_int, [_i32, _vp, _i64, _i64, _vp]),
    "mi355_op_mul_mat": (C.c_int, [_i32, _vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp]),
    "mi355_op_ffn_gate_up": (C.c_int, [_i32, _vp, _vp, _i64, _i64, _vp, _i64, _vp]),
    "mi355_op_rms_norm_mul": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp]),
    "mi355_op_rope": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, _f32, _f32, _vp, _i32]),
    "mi355_op_rope_yarn": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i64, _f32, _f32, _vp, _i32, _f32, _f32, _f32, _f32]),
    "mi355_op_get_rows": (C.c_int, [_i32, _vp, _i64, _i64, _vp, _i64, _vp]),
    "mi355_op_swiglu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "mi355_op_soft_max": (C.c_int, [_vp, _vp, _i64, _i64, _f32, _vp]),
    "mi355_op_moe_route": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "mi355_op_flash_attn": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _f32, _vp]),
    "mi355_op_attn_step": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _i32, _f32, _i32, _vp, _i64, _vp, _i32,
                                     _vp, _vp, _vp, _vp]),
    "mi355_bench_hbm_read": (C.c_double, [_sz, C.c_int]),
    "mi355_profile_last_decode": (_i32, [_vp, C.POINTER(_cp), C.POINTER(_f32), _i32]),
    "mi355_profile_enable": (None, [_vp, _i32]),
    "mi355_debug_force_moe_ids": (C.c_int, [_vp, _vp, _i32, _i32, _i32]),
    "mi355_bench_weight_sweep": (C.c_double, [_vp, C.c_int, C.POINTER(_u64)]),
    "mi355_bench_weight_sweep2": (C.c_double, [_vp, C.c_int, C.POINTER(_u64), C.POINTER(C.c_int32)]),
    "mi355_tokenize": (_i32, [_vp, _cp, _i32, C.POINTER(C.c_int32), _i32, _i32, _i32]),
    "mi355_token_to_piece": (_i32, [_vp, _i32, C.c_char_p, _i32, _i32]),
    "mi355_token_bos": (_i32, [_vp]),
    "mi355_token_eos": (_i32, [_vp]),
    "mi355_token_is_eog": (_i32, [_vp, _i32]),
    "mi355_engine_create": (_vp, []),
    "mi355_engine_destroy": (None, [_vp]),
    "mi355_engine_set_release_callback": (None, [_vp, _vp]),
    "mi355_engine_load_model": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_unload_model": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_get_model_status": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_get_models": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_handle_chat_completion": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_handle_embedding": (None, [_vp, _cp, ENGINE_CB, _vp]),
    "mi355_engine_is_supported": (_i32, [_vp, _cp]),
    "mi355_engine_stop_inferencing": (None, [_vp, _cp]),
    "mi355_engine_load": (None, [_vp, _cp, _cp, _i32, _cp, _i32, _i32]),
    "mi355_engine_unload": (None, [_vp]),
    "mi355_engine_set_file_logger": (None, [_vp, _i32, _cp]),
    "mi355_engine_set_log_level": (None, [_vp, _i32]),
    "mi355_engine_set_log_callback": (None, [_vp, _vp, _vp]),
    "mi355_clip_model_load": (_vp, [_cp, _i32]),
    "mi355_clip_free": (None, [_vp]),
    "mi355_clip_n_mmproj_embd": (_i32, [_vp]),
    "mi355_clip_n_patches": (_i32, [_vp]),
    "mi355_clip_image_size": (_i32, [_vp]),
    "mi355_clip_max_image_rows": (_i32, [_vp]),
    "mi355_clip_image_preprocess_grid": (_i32, [_vp, _vp, _i32, _i32, _vp, _sz, _vp, _vp]),
    "mi355_clip_image_load_from_bytes": (_i32, [_vp, _sz, C.POINTER(_i32), C.POINTER(_i32), _vp, _sz]),
    "mi355_clip_image_preprocess": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "mi355_clip_image_encode": (_i32, [_vp, _vp, _vp]),
    "mi355_llava_image_embed_from_bytes": (_i32, [_vp, _vp, _sz, _vp, _sz]),
    "mi355_tp_p2p_local_handle": (C.c_int, [_vp, _sz, _sz]),
    "mi355_tp_p2p_local_handle2": (C.c_int, [_vp, _sz, _sz, _sz]),
    "mi355_tp_p2p_prompt_exchanges": (C.c_int64, []),
    "mi355_tp_p2p_enable": (C.c_int, [_vp, _sz]),
    "mi355_tp_p2p_exchanges": (C.c_int64, []),
    "mi355_tp_unique_id": (C.c_int, [_vp, _sz]),
    "mi355_tp_init": (C.c_int, [_i32, _i32, _i32, _vp, _sz]),
    "mi355_tp_shutdown": (None, []),
    "mi355_tp_worker_main": (C.c_int, [C.c_int]),
    "mi355_tp_rank": (_i32, []),
    "mi355_tp_size": (_i32, []),
    "mi355_tp_set_host_exchange": (C.c_int, [_vp, _vp, _i32, _i32]),
}
TP_HOST_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int32)
TP_ID_BYTES = 128

_lib = None


def load_library(path: str | None = None):
    """dlopen the native library and type every exported entry point.  Raises if it is missing:
    there is no Python / CPU fallback for the compute path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or lib_path()
    if not os.path.exists(p):
        raise MI355Error(f"{p} not found: build it with `python cortex.llamacpp_amd/build.py` (hipcc, gfx950)")
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        if name.startswith("mi355_debug_") and os.environ.get("MI355_LLAMA_LIB") and not hasattr(lib, name):
            continue                     # (tools/ab_libs.sh: an older build of the library under MI355_LLAMA_LIB may lack a diagnosis getter)
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _err(lib) -> str:
    return (lib.mi355_last_error() or b"").decode("utf-8", "replace")


class Backend:
    """mi355_backend_init / per-op entry points."""

    def __init__(self):
        self.lib = load_library()
        rc = self.lib.mi355_backend_init()
        if rc != 0:
            raise MI355Error(f"mi355_backend_init failed ({rc}): {_err(self.lib)}")

    def _chk(self, rc, what):
        if rc != 0:
            raise MI355Error(f"{what} failed ({rc}): {_err(self.lib)}")

    def set_option(self, name: str, value: int) -> None:
        self._chk(self.lib.mi355_debug_set_option(name.encode(), int(value)), f"set_option({name})")

    def system_info(self) -> str:
        return self.lib.mi355_print_system_info().decode()

    def quantize_act(self, act_type: int, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        rows, n = (1, x.size) if x.ndim == 1 else x.shape
        bb = 292 * (n // 256) if act_type == Q8_K else 34 * (n // 32)
        out = np.zeros(rows * bb, np.uint8)
        self._chk(self.lib.mi355_op_quantize_act(act_type, _ptr(x), n, rows, _ptr(out)), "op_quantize_act")
        return out.reshape(rows, bb)

    def mul_mat(self, t: int, W: np.ndarray, N: int, K: int, x: np.ndarray, want_ints: bool = False):
        W = np.ascontiguousarray(W.view(np.uint8).reshape(-1))
        x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
        T = x.shape[0]
        y = np.zeros((T, N), np.float32)
        isum = msum = None
        if want_ints:
            nblk = K // 32 if t in (Q8_0, Q4_0, Q5_0, IQ4_NL) else K // 256
            isum = np.zeros((T, N, nblk), np.int32)
            msum = np.zeros((T, N, nblk), np.int32)
        self._chk(self.lib.mi355_op_mul_mat(t, _ptr(W), N, K, _ptr(x), T, _ptr(y), _ptr(isum), _ptr(msum)), "op_mul_mat")
        return (y, isum, msum) if want_ints else y

    def ffn_gate_up(self, t: int, Wg: np.ndarray, Wu: np.ndarray, N: int, K: int, x: np.ndarray) -> np.ndarray:
        Wg = np.ascontiguousarray(Wg.view(np.uint8).reshape(-1))
        Wu = np.ascontiguousarray(Wu.view(np.uint8).reshape(-1))
        x = np.ascontiguousarray(x, np.float32).reshape(-1, K)
        y = np.zeros((x.shape[0], N), np.float32)
        self._chk(self.lib.mi355_op_ffn_gate_up(t, _ptr(Wg), _ptr(Wu), N, K, _ptr(x), x.shape[0], _ptr(y)), "op_ffn_gate_up")
        return y

    def rms_norm_mul(self, x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        T, n = (1, x.size) if x.ndim == 1 else x.shape
        y = np.zeros_like(x)
        self._chk(self.lib.mi355_op_rms_norm_mul(_ptr(x), _ptr(w), n, T, eps, _ptr(y)), "op_rms_norm_mul")
        return y

    def rope(self, x: np.ndarray, n_head: int, head_dim: int, pos, base: float, neox: bool = False, n_rot=None,
             freq_scale: float = 1.0, freq_factors=None) -> np.ndarray:
        pos = np.ascontiguousarray(pos, np.int32).reshape(-1)
        y = np.array(x, np.float32, copy=True).reshape(pos.size, n_head * head_dim)
        ff = None if freq_factors is None else np.ascontiguousarray(freq_factors, np.float32)
        self._chk(self.lib.mi355_op_rope(_ptr(y), n_head, head_dim, n_rot or head_dim, _ptr(pos), pos.size, base, freq_scale,
                                         _ptr(ff), int(neox)), "op_rope")
        return y.reshape(pos.size, n_head, head_dim)

    def rope_yarn(self, x: np.ndarray, n_head: int, head_dim: int, pos, base: float, freq_scale: float, ext_factor: float, attn_factor: float,
                  corr_lo: float, corr_hi: float, neox: bool = False, n_rot=None) -> np.ndarray:
        pos = np.ascontiguousarray(pos, np.int32).reshape(-1)
        y = np.array(x, np.float32, copy=True).reshape(pos.size, n_head * head_dim)
        self._chk(self.lib.mi355_op_rope_yarn(_ptr(y), n_head, head_dim, n_rot or head_dim, _ptr(pos), pos.size, base, freq_scale, None, int(neox),
                                              ext_factor, attn_factor, corr_lo, corr_hi), "op_rope_yarn")
        return y.reshape(pos.size, n_head, head_dim)

    def get_rows(self, t: int, table: np.ndarray, K: int, n_rows: int, ids) -> np.ndarray:
        table = np.ascontiguousarray(table.view(np.uint8).reshape(-1))
        ids = np.ascontiguousarray(ids, np.int32)
        out = np.zeros((ids.size, K), np.float32)
        self._chk(self.lib.mi355_op_get_rows(t, _ptr(table), K, n_rows, _ptr(ids), ids.size, _ptr(out)), "op_get_rows")
        return out

    def swiglu(self, g: np.ndarray, u: np.ndarray) -> np.ndarray:
        g = np.ascontiguousarray(g, np.float32)
        u = np.ascontiguousarray(u, np.float32)
        y = np.zeros_like(g)
        self._chk(self.lib.mi355_op_swiglu(_ptr(g), _ptr(u), g.size, _ptr(y)), "op_swiglu")
        return y

    def soft_max(self, x: np.ndarray, mask, scale: float) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        rows, n = (1, x.size) if x.ndim == 1 else x.shape
        m = None if mask is None else np.ascontiguousarray(mask, np.float32)
        y = np.zeros_like(x)
        self._chk(self.lib.mi355_op_soft_max(_ptr(x), _ptr(m), n, rows, scale, _ptr(y)), "op_soft_max")
        return y

    def moe_route(self, logits: np.ndarray, k: int):
        x = np.ascontiguousarray(logits, np.float32)
        T, ne = x.shape
        ids = np.zeros((T, k), np.int32); w = np.zeros((T, k), np.float32)
        self._chk(self.lib.mi355_op_moe_route(_ptr(x), T, ne, k, _ptr(ids), _ptr(w)), "op_moe_route")
        return ids, w

    def flash_attn(self, q: np.ndarray, n_head: int, n_head_kv: int, hd: int, type_k: int, k_rows: np.ndarray, type_v: int,
                   v_rows: np.ndarray, cell_pos, q_pos, scale: float) -> np.ndarray:
        q = np.ascontiguousarray(q, np.float32)
        q_pos = np.ascontiguousarray(q_pos, np.int32).reshape(-1)
        cell_pos = np.ascontiguousarray(cell_pos, np.int32)
        k_rows = np.ascontiguousarray(k_rows.view(np.uint8))
        v_rows = np.ascontiguousarray(v_rows.view(np.uint8))
        out = np.zeros((q_pos.size, n_head, hd), np.float32)
        self._chk(self.lib.mi355_op_flash_attn(_ptr(q), q_pos.size, n_head, n_head_kv, hd, type_k, _ptr(k_rows), type_v,
                                               _ptr(v_rows), cell_pos.size, _ptr(cell_pos), _ptr(q_pos), scale, _ptr(out)),
                  "op_flash_attn")
        return out

    def attn_step(self, q, k_new, v_new, n_head: int, n_head_kv: int, hd: int, type_k: int, k_rows, type_v: int, v_rows, cell_pos, tok_pos: int,
                  tok_cell: int, rope_base: float, n_rot: int, scale: float, type_o: int, W_o, n_embd: int, resid, fused: bool, k_row_bytes: int, v_row_bytes: int):
        """One single-token attention block (mi355_op_attn_step): returns (att [H * D], out [n_embd], k_row, v_row)."""
        q = np.ascontiguousarray(q, np.float32); k_new = np.ascontiguousarray(k_new, np.float32); v_new = np.ascontiguousarray(v_new, np.float32)
        cell_pos = np.ascontiguousarray(cell_pos, np.int32)
        k_rows = np.ascontiguousarray(k_rows.view(np.uint8)); v_rows = np.ascontiguousarray(v_rows.view(np.uint8))
        W_o = np.ascontiguousarray(W_o.view(np.uint8)); resid = np.ascontiguousarray(resid, np.float32)
        att = np.zeros(n_head * hd, np.float32); out = np.zeros(n_embd, np.float32)
        kr = np.zeros(k_row_bytes, np.uint8); vr = np.zeros(v_row_bytes, np.uint8)
        self._chk(self.lib.mi355_op_attn_step(_ptr(q), _ptr(k_new), _ptr(v_new), n_head, n_head_kv, hd, type_k, _ptr(k_rows), type_v, _ptr(v_rows), cell_pos.size,
                                              _ptr(cell_pos), int(tok_pos), int(tok_cell), float(rope_base), int(n_rot), float(scale), type_o, _ptr(W_o), n_embd,
                                              _ptr(resid), int(bool(fused)), _ptr(att), _ptr(out), _ptr(kr), _ptr(vr)), "op_attn_step")
        return att, out, kr, vr

    def hbm_read_gbps(self, nbytes: int = 1 << 30, iters: int = 10) -> float:
        return float(self.lib.mi355_bench_hbm_read(nbytes, iters))


def gloo_exchange(group=None):
    """The host exchange of mi355_tp_set_host_exchange over a torch.distributed (gloo) group: op 0 sums the buffer in
    place over the ranks, op 1 gathers `n` floats per rank (the caller's part already at buf + rank * n)."""
    import torch
    import torch.distributed as dist

    def fn(_user, buf, n, op):
        try:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            total = n if op == 0 else n * world
            t = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(total,)))
            if op == 0:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                parts = [torch.empty(n, dtype=torch.float32) for _ in range(world)]
                dist.all_gather(parts, t[rank * n:(rank + 1) * n].clone(), group=group)
                for r in range(world):
                    t[r * n:(r + 1) * n] = parts[r]
            return 0
        except Exception:   # noqa: BLE001 - a Python exception must not unwind through the C caller
            import traceback
            traceback.print_exc()
            return 1
    return fn


_tp_keepalive = []


def tp_p2p_enable(rank: int, size: int, device: int, max_floats: int, group=None, prompt_floats: int = 0):
    """One-shot peer-to-peer all-reduce for messages of up to max_floats floats (mi355_tp_p2p_*): every rank exports its exchange buffer, the 64-byte
    IPC handles are all-gathered over torch.distributed, every rank maps its peers' buffers.  prompt_floats > max_floats: messages up to that size
    take the reduce-scatter + all-gather kernel (mi355_tp_p2p_local_handle2)."""
    import torch
    import torch.distributed as dist
    lib = load_library()
    hb = (C.c_uint8 * 64)()
    if lib.mi355_tp_p2p_local_handle2(hb, 64, max_floats, prompt_floats) != 64:
        raise MI355Error(f"mi355_tp_p2p_local_handle2 failed: {_err(lib)}")
    backend = dist.get_backend(group)
    dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
    mine = torch.tensor(list(bytes(hb)), dtype=torch.uint8, device=dev)
    parts = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(size)]
    dist.all_gather(parts, mine, group=group)
    allh = (C.c_uint8 * (64 * size))(*[int(v) for p_ in parts for v in p_.cpu().tolist()])
    if lib.mi355_tp_p2p_enable(allh, 64 * size) != 0:
        raise MI355Error(f"mi355_tp_p2p_enable failed: {_err(lib)}")


def tp_p2p_exchanges() -> int:
    return int(load_library().mi355_tp_p2p_exchanges())


def tp_p2p_prompt_exchanges() -> int:
    return int(load_library().mi355_tp_p2p_prompt_exchanges())


def tp_init(rank: int, size: int, device: int = 0, transport: str = "rccl", group=None, p2p_floats: int = 0, p2p_prompt_floats: int = 0):
    """Forms the process's row-split group (include/mi355_llama.h, mi355_tp_*).  transport "rccl": rank 0 makes the RCCL
    id, torch.distributed (any backend) carries it to the others, every rank calls mi355_tp_init on its device.
    transport "host": the exchange goes through gloo_exchange(group) (ranks sharing one GPU; validation only).
    p2p_floats > 0: all-reduces of up to that many floats (the decode steps' exchanges) take the one-shot peer-to-peer kernel;
    p2p_prompt_floats > p2p_floats: larger ones up to that size (prompt batches) the reduce-scatter + all-gather kernel."""
    lib = load_library()
    if transport == "host":
        cb = TP_HOST_EXCHANGE(gloo_exchange(group))
        _tp_keepalive.append(cb)
        if lib.mi355_tp_set_host_exchange(C.cast(cb, C.c_void_p), None, rank, size) != 0:
            raise MI355Error(f"mi355_tp_set_host_exchange failed: {_err(lib)}")
        if p2p_floats > 0 and size > 1:
            tp_p2p_enable(rank, size, device, p2p_floats, group, p2p_prompt_floats)
        return
    buf = (C.c_uint8 * TP_ID_BYTES)()
    if size > 1:
        import torch
        import torch.distributed as dist
        if rank == 0 and lib.mi355_tp_unique_id(buf, TP_ID_BYTES) != TP_ID_BYTES:
            raise MI355Error(f"mi355_tp_unique_id failed: {_err(lib)}")
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
        backend = dist.get_backend(group)
        dev = torch.device("cuda", device) if backend == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.broadcast(t, src=0, group=group)
        buf = (C.c_uint8 * TP_ID_BYTES)(*t.cpu().tolist())
    elif lib.mi355_tp_unique_id(buf, TP_ID_BYTES) != TP_ID_BYTES:
        raise MI355Error(f"mi355_tp_unique_id failed: {_err(lib)}")
    if lib.mi355_tp_init(device, rank, size, buf, TP_ID_BYTES) != 0:
        raise MI355Error(f"mi355_tp_init failed: {_err(lib)}")
    if p2p_floats > 0 and size > 1:
        tp_p2p_enable(rank, size, device, p2p_floats, group, p2p_prompt_floats)


def tp_shutdown():
    load_library().mi355_tp_shutdown()
    _tp_keepalive.clear()


class Clip:
    """The image side of a LLaVA request (clip_model_load / clip_image_load_from_bytes / clip_image_preprocess / clip_image_encode)."""

    def __init__(self, path: str, main_gpu: int = 0):
        self.lib = load_library()
        self.h = self.lib.mi355_clip_model_load(path.encode(), main_gpu)
        if not self.h:
            raise MI355Error(f"mi355_clip_model_load({path}) failed: {_err(self.lib)}")
        self.n_embd = self.lib.mi355_clip_n_mmproj_embd(self.h)
        self.n_patches = self.lib.mi355_clip_n_patches(self.h)
        self.image_size = self.lib.mi355_clip_image_size(self.h)
        self.max_image_rows = self.lib.mi355_clip_max_image_rows(self.h)

    def load_image(self, data: bytes) -> np.ndarray:
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        nx, ny = C.c_int32(0), C.c_int32(0)
        if self.lib.mi355_clip_image_load_from_bytes(buf, len(data), C.byref(nx), C.byref(ny), None, 0) != 0:
            raise MI355Error(f"image: {_err(self.lib)}")
        rgb = np.empty((ny.value, nx.value, 3), np.uint8)
        if self.lib.mi355_clip_image_load_from_bytes(buf, len(data), C.byref(nx), C.byref(ny), rgb.ctypes.data, rgb.nbytes) != 0:
            raise MI355Error(f"image: {_err(self.lib)}")
        return rgb

    def preprocess(self, rgb: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        ny, nx, _ = rgb.shape
        out = np.empty((3, self.image_size, self.image_size), np.float32)
        if self.lib.mi355_clip_image_preprocess(self.h, rgb.ctypes.data, nx, ny, out.ctypes.data) != 0:
            raise MI355Error(f"preprocess: {_err(self.lib)}")
        return out

    def preprocess_grid(self, rgb: np.ndarray):
        """Every image the encoder sees for one picture (LLaVA-1.6: overview + tiles): ([n, 3, S, S], grid_w, grid_h)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        ny, nx, _ = rgb.shape
        cap = self.max_image_rows // self.n_patches
        out = np.empty((cap, 3, self.image_size, self.image_size), np.float32)
        gw, gh = C.c_int32(0), C.c_int32(0)
        n = self.lib.mi355_clip_image_preprocess_grid(self.h, rgb.ctypes.data, nx, ny, out.ctypes.data, out.size, C.byref(gw), C.byref(gh))
        if n < 0:
            raise MI355Error(f"preprocess_grid: {_err(self.lib)}")
        return out[:n], gw.value, gh.value

    def encode(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, np.float32)
        out = np.empty((self.n_patches, self.n_embd), np.float32)
        if self.lib.mi355_clip_image_encode(self.h, img.ctypes.data, out.ctypes.data) != 0:
            raise MI355Error(f"encode: {_err(self.lib)}")
        return out

    def embed_bytes(self, data: bytes) -> np.ndarray:
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        out = np.empty((self.max_image_rows, self.n_embd), np.float32)
        n = self.lib.mi355_llava_image_embed_from_bytes(self.h, buf, len(data), out.ctypes.data, out.size)
        if n < 0:
            raise MI355Error(f"llava_image_embed: {_err(self.lib)}")
        return out[:n]

    def close(self):
        if self.h:
            self.lib.mi355_clip_free(self.h)
            self.h = None


class Model:
    def __init__(self, path: str, n_gpu_layers: int = 300, main_gpu: int = 0, prefill_planes: int = -1, tp_rank: int = 0, tp_size: int = 1):
        self.lib = load_library()
        mp = self.lib.mi355_model_default_params()
        mp.n_gpu_layers = n_gpu_layers
        mp.main_gpu = main_gpu
        mp.prefill_planes = prefill_planes
        mp.tp_rank, mp.tp_size = tp_rank, tp_size
        self.h = self.lib.mi355_model_load_from_file(path.encode(), mp)
        if not self.h:
            raise MI355Error(f"mi355_model_load_from_file({path}) failed: {_err(self.lib)}")
        L = self.lib
        self.n_vocab = L.mi355_model_n_vocab(self.h)
        self.n_embd = L.mi355_model_n_embd(self.h)
        self.n_layer = L.mi355_model_n_layer(self.h)
        self.n_head = L.mi355_model_n_head(self.h)
        self.n_head_kv = L.mi355_model_n_head_kv(self.h)
        self.bytes_per_token = L.mi355_model_bytes_per_token(self.h)
        self.planes_bytes = L.mi355_model_planes_bytes(self.h)
        self.size = L.mi355_model_size(self.h)
        self.vram = L.mi355_model_other_buffer(self.h)
        self.ram = L.mi355_model_cpu_buffer(self.h)
        self.desc = L.mi355_model_desc(self.h).decode()

    def meta(self, key: str):
        buf = C.create_string_buffer(512)
        if self.lib.mi355_model_meta_str(self.h, key.encode(), buf, 512):
            return buf.value.decode()
        return None

    def tokenize(self, text: str, add_special: bool = True, parse_special: bool = False) -> list[int]:
        raw = text.encode()
        cap = len(raw) + 8
        out = (C.c_int32 * cap)()
        n = self.lib.mi355_tokenize(self.h, raw, len(raw), out, cap, int(add_special), int(parse_special))
        if n < 0:
            raise MI355Error(f"mi355_tokenize failed: {_err(self.lib)}")
        return list(out[:n])

    def token_to_piece(self, tok: int, special: bool = True) -> bytes:
        buf = C.create_string_buffer(256)
        n = self.lib.mi355_token_to_piece(self.h, tok, buf, 256, int(special))
        if n < 0:
            raise MI355Error(f"mi355_token_to_piece failed: {_err(self.lib)}")
        return buf.raw[:n]

    def close(self):
        if self.h:
            self.lib.mi355_model_free(self.h)
            self.h = None


class Context:
    def __init__(self, model: Model, n_ctx: int = 512, n_batch: int = 2048, n_ubatch: int = 512, n_seq_max: int = 1,
                 type_k: int = F16, type_v: int = F16, flash_attn: bool = True, use_graphs: bool = True,
                 logits_to_host: bool = True):
        self.lib = model.lib
        self.model = model
        cp = self.lib.mi355_context_default_params()
        cp.n_ctx, cp.n_batch, cp.n_ubatch, cp.n_seq_max = n_ctx, n_batch, n_ubatch, n_seq_max
        cp.type_k, cp.type_v, cp.flash_attn, cp.use_graphs = type_k, type_v, int(flash_attn), int(use_graphs)
        cp.logits_to_host = int(logits_to_host)
        self.h = self.lib.mi355_context_new(model.h, cp)
        if not self.h:
            raise MI355Error(f"mi355_context_new failed: {_err(self.lib)}")
        self.n_ctx = n_ctx
        self._batch_cap = 0
        self._b = None

    def _batch(self, n: int) -> Batch:
        if n > self._batch_cap:
            if self._b is not None:
                self.lib.mi355_batch_free(self._b)
            self._batch_cap = max(n, 64)
            self._b = self.lib.mi355_batch_init(self._batch_cap, 0, 1)
        return self._b

    def decode(self, tokens, pos, seq=None, logits=None) -> int:
        """llama_decode on one batch; logits=None flags only the last token."""
        tokens = np.asarray(tokens, np.int32).reshape(-1)
        pos = np.asarray(pos, np.int32).reshape(-1)
        n = tokens.size
        b = self._batch(n)
        b.n_tokens = n
        C.memmove(b.token, tokens.ctypes.data, 4 * n)
        C.memmove(b.pos, pos.ctypes.data, 4 * n)
        for i in range(n):
            b.n_seq_id[i] = 1
            b.seq_id[i][0] = 0 if seq is None else int(seq[i] if np.ndim(seq) else seq)
            b.logits[i] = (1 if i == n - 1 else 0) if logits is None else int(bool(logits[i]))
        rc = self.lib.mi355_decode(self.h, b)
        if rc < 0:
            raise MI355Error(f"mi355_decode failed ({rc}): {_err(self.lib)}")
        return rc

    def decode_embd(self, embd, pos, seq=None, logits=None) -> int:
        """llama_decode on a batch of embedding rows [n][n_embd] (llama_batch.embd: how image embeddings reach the model)."""
        embd = np.ascontiguousarray(embd, np.float32)
        pos = np.asarray(pos, np.int32).reshape(-1)
        n = pos.size
        if embd.shape != (n, self.model.n_embd):
            raise ValueError(f"embd must be [{n}][{self.model.n_embd}]")
        b = self.lib.mi355_batch_init(n, self.model.n_embd, 1)
        try:
            b.n_tokens = n
            C.memmove(b.embd, embd.ctypes.data, embd.nbytes)
            C.memmove(b.pos, pos.ctypes.data, 4 * n)
            for i in range(n):
                b.n_seq_id[i] = 1
                b.seq_id[i][0] = 0 if seq is None else int(seq[i] if np.ndim(seq) else seq)
                b.logits[i] = (1 if i == n - 1 else 0) if logits is None else int(bool(logits[i]))
            rc = self.lib.mi355_decode(self.h, b)
        finally:
            self.lib.mi355_batch_free(b)
        if rc < 0:
            raise MI355Error(f"mi355_decode failed ({rc}): {_err(self.lib)}")
        return rc

    def greedy_steps(self, first: int, pos0: int, n: int, seq: int = 0) -> np.ndarray:
        """n greedy single-token steps (decode, logits row host-visible, arg-max fed back) in one C call; returns the n tokens"""
        out = np.empty(n, np.int32)
        done = self.lib.mi355_greedy_steps(self.h, int(first), int(pos0), int(seq), int(n), out.ctypes.data)
        if done != n:
            raise MI355Error(f"mi355_greedy_steps stopped after {done} of {n}: {_err(self.lib)}")
        return out

    def logits(self, i: int = -1) -> np.ndarray:
        p = self.lib.mi355_get_logits_ith(self.h, i)
        if not p:
            raise MI355Error(f"no logits for that batch row: {_err(self.lib)}")
        return np.ctypeslib.as_array(p, shape=(self.model.n_vocab,)).copy()

    def logits_ready(self, i: int = -1) -> None:
        """Block until the logits row is host-visible (no numpy copy)."""
        if not self.lib.mi355_get_logits_ith(self.h, i):
            raise MI355Error(f"no logits for that batch row: {_err(self.lib)}")

    def set_embeddings(self, on: bool = True) -> None:
        self.lib.mi355_set_embeddings(self.h, int(on))

    def embeddings(self, i: int = -1) -> np.ndarray:
        p = self.lib.mi355_get_embeddings_ith(self.h, i)
        if not p:
            raise MI355Error("no embeddings for that row")
        return np.ctypeslib.as_array(p, shape=(self.model.n_embd,)).copy()

    def argmax(self, i: int = -1) -> int:
        return int(self.lib.mi355_get_argmax_ith(self.h, i))

    def mega_steps(self) -> int:
        """Single-token steps that ran as one whole-step launch (diagnosis)."""
        return int(self.lib.mi355_debug_mega_steps(self.h))

    def topk(self, k: int, i: int = -1, adj_tok=(), adj_bias=(), adj_count=(), repeat: float = 1.0, freq: float = 0.0, present: float = 0.0):
        """Device-side sampling front end: (tokens [k] int32, logits [k] f32) of row i after the adjustments, best first."""
        at = np.ascontiguousarray(adj_tok, np.int32); ab = np.ascontiguousarray(adj_bias, np.float32); ac = np.ascontiguousarray(adj_count, np.int32)
        toks = np.zeros(k, np.int32); lg = np.zeros(k, np.float32)
        r = self.lib.mi355_get_topk_ith(self.h, i, k, at.size, _ptr(at), _ptr(ab), _ptr(ac), repeat, freq, present, _ptr(toks), _ptr(lg))
        if r != k:
            raise MI355Error(f"mi355_get_topk_ith failed: {_err(self.lib)}")
        return toks, lg

    def fused_skipped_steps(self) -> int:
        return int(self.lib.mi355_debug_fused_skipped_steps(self.h))

    def qkv_attn_launches(self) -> int:
        return int(self.lib.mi355_debug_qkv_attn_launches(self.h))

    def engine_steps(self) -> int:
        """Single-token steps that ran through the layer engine (one persistent launch per layer; diagnosis)."""
        return int(self.lib.mi355_debug_engine_steps(self.h))

    def synchronize(self):
        self.lib.mi355_synchronize(self.h)

    def kv_clear(self):
        self.lib.mi355_kv_cache_clear(self.h)

    def kv_seq_rm(self, seq, p0, p1) -> bool:
        return bool(self.lib.mi355_kv_cache_seq_rm(self.h, seq, p0, p1))

    def kv_seq_cp(self, s, d, p0, p1):
        self.lib.mi355_kv_cache_seq_cp(self.h, s, d, p0, p1)

    def kv_seq_add(self, seq, p0, p1, delta):
        self.lib.mi355_kv_cache_seq_add(self.h, seq, p0, p1, delta)

    def kv_used(self) -> int:
        return int(self.lib.mi355_kv_cache_used_cells(self.h))

    def enable_taps(self, on: bool = True):
        self.lib.mi355_debug_enable_taps(self.h, int(on))

    def layer_out(self, il: int, n_tokens: int) -> np.ndarray:
        out = np.zeros((n_tokens, self.model.n_embd), np.float32)
        n = self.lib.mi355_debug_layer_out(self.h, il, _ptr(out), out.size)
        if n < 0:
            raise MI355Error("debug taps not enabled")
        return out[:n]

    def profile(self, on: bool = True):
        self.lib.mi355_profile_enable(self.h, int(on))

    def last_profile(self) -> dict:
        names = (_cp * 64)()
        us = (_f32 * 64)()
        n = self.lib.mi355_profile_last_decode(self.h, names, us, 64)
        return {names[i].decode(): float(us[i]) for i in range(n)}

    def force_moe_ids(self, ids: np.ndarray) -> None:
        """ids [n_layer][T][k]: the experts the NEXT decode call takes instead of its router's selection (test hook)."""
        ids = np.ascontiguousarray(ids, np.int32)
        rc = self.lib.mi355_debug_force_moe_ids(self.h, _ptr(ids), ids.shape[0], ids.shape[1], ids.shape[2])
        if rc != 0:
            raise MI355Error(f"mi355_debug_force_moe_ids failed ({rc}): {_err(self.lib)}")

    def weight_sweep_us(self, iters: int = 5):
        b = _u64(0)
        us = self.lib.mi355_bench_weight_sweep(self.h, iters, C.byref(b))
        return float(us), int(b.value)

    def weight_sweep(self, iters: int = 5):
        """(us per sweep, weight bytes per sweep, mat-vec launches per sweep) of the step's own weight-stream launches."""
        b, n = _u64(0), C.c_int32(0)
        us = self.lib.mi355_bench_weight_sweep2(self.h, iters, C.byref(b), C.byref(n))
        return float(us), int(b.value), int(n.value)

    def close(self):
        if self._b is not None:
            self.lib.mi355_batch_free(self._b)
            self._b = None
        if self.h:
            self.lib.mi355_context_free(self.h)
            self.h = None


class Engine:
    """ctypes view of mi355_engine_* — the reference's EngineI surface (base/cortex-common/enginei.h:13-74) with JSON text
    bodies.  Every call returns the list of (status, body) pairs the callback received; a streaming chat completion
    returns one pair per SSE chunk."""

    def __init__(self):
        import json
        import threading
        self._json, self._threading = json, threading
        self.lib = load_library()
        self.h = self.lib.mi355_engine_create()
        if not self.h:
            raise MI355Error(f"mi355_engine_create failed: {_err(self.lib)}")

    def _call(self, fn, body: dict, wait_done: bool = True, timeout: float = 120.0):
        out, done = [], self._threading.Event()

        def on(status, payload, _user):
            st, bd = self._json.loads(status.decode("utf-8", "replace")), self._json.loads(payload.decode("utf-8", "replace"))
            out.append((st, bd))
            if st.get("is_done", True) or st.get("has_error", False):
                done.set()

        cb = ENGINE_CB(on)
        fn(self.h, self._json.dumps(body).encode(), cb, None)
        if wait_done and not done.wait(timeout):
            raise MI355Error("engine call timed out")
        self._keep = cb   # the callback object must outlive the native call
        return out

    def load_model(self, **body):
        return self._call(self.lib.mi355_engine_load_model, body)[-1]

    def unload_model(self, **body):
        return self._call(self.lib.mi355_engine_unload_model, body)[-1]

    def get_model_status(self, **body):
        return self._call(self.lib.mi355_engine_get_model_status, body)[-1]

    def get_models(self):
        return self._call(self.lib.mi355_engine_get_models, {})[-1]

    def chat_completion(self, **body):
        return self._call(self.lib.mi355_engine_handle_chat_completion, body)

    def embedding(self, **body):
        return self._call(self.lib.mi355_engine_handle_embedding, body)[-1]

    def is_supported(self, feature: str) -> bool:
        return bool(self.lib.mi355_engine_is_supported(self.h, feature.encode()))

    def stop_inferencing(self, model_id: str) -> None:
        self.lib.mi355_engine_stop_inferencing(self.h, model_id.encode())

    def close(self):
        if self.h:
            self.lib.mi355_engine_destroy(self.h)
            self.h = None
