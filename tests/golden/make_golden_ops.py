#!/usr/bin/env python3
"""Generates tests/golden/ops_v1.npz, e2e_v1.npz and api_shapes_v1.json — committed vectors for the ops and the whole decode path.

Provenance.  The reference (`/root/reference`) cannot produce vectors for this path (its arithmetic is an un-vendored submodule, its tests hold no
numbers: SURVEY.md §4), so parity stays "unpinned"; what these files pin is DRIFT: an oracle regression and a matching HIP regression can no
longer pass together, because both are compared with bytes that were written once.
  * ops_v1.npz — op-level vectors computed HERE by restatements that share no code with oracle/*.c or the HIP kernels:
      rope (NORM pairing, f32 angle recurrence theta *= theta_scale with glibc powf / cosf / sinf, base 1e4 and 5e5, positions 0, 1, 4095),
      masked softmax, SwiGLU, top-2 expert routing, attention with GQA 4:1 over a q8_0 and an f16 cache (float64 over the values the CPU path's operands
      hold: q quantised to Q8_0 / rounded to f16 as ggml's vec_dot types prescribe, K / V dequantised) — float64 unless the f32 order itself is the
      definition (rope).  The C oracle (CPU test) and the HIP path (GPU test) are both checked against them, each at its stated tolerance.
  * e2e_v1.npz — end to end: for three synthetic GGUF files (tiny / q4_k_m + q8_0 cache, tiny-d128 / q5_k_m + f16 cache, tiny-moe / q4_k_m + q8_0 cache; the
      file is a pure function of (config, ftype, seed): gguf_synth) a 12-token prompt and 32 greedy steps run through the C ORACLE: token ids, the logits
      row of every step and the gap between the two largest logits.  No independent twin exists for the whole model, so this one is a regression pin of the
      oracle (CPU test: the oracle still reproduces it bit for bit) and the yardstick of the HIP path (GPU test: ids equal wherever the recorded top-2
      gap exceeds the tolerance, logits within FLIP_TOL).
  * api_shapes_v1.json — the status / body JSON shapes of the engine surface, transcribed from the reference's source (src/llama_engine.cc:180-270 response
      bodies, :363-500 load / unload / status / models), since nothing of the reference can be run here.  Checked against host/engine.cc by the CPU tests.

usage: python tests/golden/make_golden_ops.py          (deterministic: fixed seeds; rewrites the three files; needs oracle/ built for e2e_v1)"""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import np_twin as tw  # noqa: E402

_libm = ctypes.CDLL("libm.so.6")
for _n in ("powf", "cosf", "sinf", "expf"):
    getattr(_libm, _n).restype = ctypes.c_float
_libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.cosf.argtypes = [ctypes.c_float]
_libm.sinf.argtypes = [ctypes.c_float]
_libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32


def rope_norm(x, pos, base):
    """ggml rope, mode NORM: pairs (2i, 2i+1); theta_0 = pos, theta_{i+1} = theta_i * theta_scale, all in f32."""
    H, D = x.shape
    theta_scale = f32(_libm.powf(f32(base), f32(-2.0) / f32(D)))
    y = x.copy()
    theta = f32(pos)
    cs = []
    for _ in range(0, D, 2):
        cs.append((f32(_libm.cosf(theta)), f32(_libm.sinf(theta))))
        theta = f32(theta * theta_scale)
    for h in range(H):
        for j, (c, s) in enumerate(cs):
            x0, x1 = x[h, 2 * j], x[h, 2 * j + 1]
            y[h, 2 * j] = f32(f32(x0 * c) - f32(x1 * s))
            y[h, 2 * j + 1] = f32(f32(x0 * s) + f32(x1 * c))
    return y


def softmax64(x, mask, scale):
    z = x.astype(np.float64) * scale + (mask.astype(np.float64) if mask is not None else 0.0)
    z = z - z[np.isfinite(z)].max()
    e = np.where(np.isfinite(z), np.exp(z), 0.0)
    return e / e.sum()


def attention64(q, H, G, D, tk, k_rows, tv, v_rows, cells, scale):
    """flash_attn_ext for one query token in float64: q in K's vec_dot type (q8_0 blocks for a q8_0 cache, f16 for an f16 cache), K / V dequantised."""
    R = H // G
    out = np.zeros((H, D), np.float64)
    for h in range(H):
        g = h // R
        qh = q[h]
        if tk == tw.Q8_0:
            qb = tw.quantize_q8_0(qh).view(tw.DT[tw.Q8_0])
            qd = qb["d"].astype(np.float64); qc = qb["qs"].astype(np.int64)
        else:
            q16 = qh.astype(np.float16).astype(np.float64)
        s = np.empty(len(cells), np.float64)
        for n, c in enumerate(cells):
            if tk == tw.Q8_0:
                kb = k_rows[c].view(tw.DT[tw.Q8_0])[g * (D // 32):(g + 1) * (D // 32)]
                isum = (kb["qs"].astype(np.int64) * qc).sum(axis=1)
                s[n] = float((isum * (kb["d"].astype(np.float64) * qd)).sum())
            else:
                kr = k_rows[c].view(np.float16)[g * D:(g + 1) * D].astype(np.float64)
                s[n] = float((kr * q16).sum())
        p = softmax64(s, None, scale)
        for n, c in enumerate(cells):
            if tv == tw.Q8_0:
                vb = v_rows[c].view(tw.DT[tw.Q8_0])[g * (D // 32):(g + 1) * (D // 32)]
                vr = (vb["qs"].astype(np.float64) * vb["d"].astype(np.float64)[:, None]).reshape(-1)
            else:
                vr = v_rows[c].view(np.float16)[g * D:(g + 1) * D].astype(np.float64)
            out[h] += p[n] * vr
    return out


def make_ops():
    rng = np.random.default_rng(20250405)
    out = {}
    # ---- rope
    H, D = 4, 128
    xr = rng.standard_normal((H, D)).astype(np.float32)
    out["rope_x"] = xr
    out["rope_pos"] = np.array([0, 1, 4095], np.int32)
    out["rope_base"] = np.array([1e4, 5e5], np.float32)
    out["rope_y"] = np.stack([np.stack([rope_norm(xr, int(p), float(b)) for p in out["rope_pos"]]) for b in out["rope_base"]])
    # ---- masked softmax (scale 0.25)
    sx = (rng.standard_normal((3, 200)) * 5).astype(np.float32)
    sm = np.where(rng.random((3, 200)) < 0.3, -np.inf, 0).astype(np.float32)
    sm[:, 0] = 0
    out["softmax_x"], out["softmax_mask"] = sx, sm
    out["softmax_scale"] = np.array([0.25], np.float32)
    out["softmax_y"] = np.stack([softmax64(sx[r], sm[r], 0.25) for r in range(3)])
    # ---- SwiGLU
    g = (rng.standard_normal(2048) * 3).astype(np.float32)
    u = rng.standard_normal(2048).astype(np.float32)
    out["swiglu_g"], out["swiglu_u"] = g, u
    g64 = g.astype(np.float64)
    out["swiglu_y"] = g64 / (1.0 + np.exp(-g64)) * u.astype(np.float64)
    # ---- top-2 routing over 8 experts (softmax, two largest with the lower index winning ties, renormalised)
    lg = rng.standard_normal((6, 8)).astype(np.float32) * 2
    lg[5, 3] = lg[5, 6] = lg[5].max() + 1.0                     # an exact tie for the first place: expert 3 first
    out["route_logits"] = lg
    ids = np.zeros((6, 2), np.int32); w = np.zeros((6, 2), np.float64)
    for t in range(6):
        p = softmax64(lg[t], None, 1.0)
        first = int(np.argmax(p)); p2 = p.copy(); p2[first] = -1.0; second = int(np.argmax(p2))
        ids[t] = [first, second]
        w[t] = np.array([p[first], p[second]]) / (p[first] + p[second])
    out["route_ids"], out["route_w"] = ids, w
    # ---- attention, GQA 4:1, head_dim 128, 96 cells with holes; q8_0 and f16 caches
    H, G, D, NC = 8, 2, 128, 96
    kf = rng.standard_normal((NC, G * D)).astype(np.float32)
    vf = (rng.standard_normal((NC, G * D)) * rng.uniform(0.2, 3.0, (NC, 1))).astype(np.float32)
    cell_pos = np.arange(NC, dtype=np.int32)
    cell_pos[[5, 17, 40, 41, 77]] = -1
    q_pos = np.array([NC - 1, 50, 3], np.int32)
    q = rng.standard_normal((q_pos.size, H, D)).astype(np.float32)
    out["attn_q"], out["attn_cell_pos"], out["attn_q_pos"] = q, cell_pos, q_pos
    out["attn_shape_H_G_D"] = np.array([H, G, D], np.int32)
    for name, t in (("q8_0", tw.Q8_0), ("f16", 1)):
        if t == tw.Q8_0:
            kc = np.stack([tw.quantize_q8_0(r) for r in kf]); vc = np.stack([tw.quantize_q8_0(r) for r in vf])
        else:
            kc = np.stack([r.astype(np.float16).view(np.uint8) for r in kf]); vc = np.stack([r.astype(np.float16).view(np.uint8) for r in vf])
        out[f"attn_k_{name}"], out[f"attn_v_{name}"] = kc, vc
        ys = []
        for i, qp in enumerate(q_pos):
            cells = np.nonzero((cell_pos >= 0) & (cell_pos <= qp))[0]
            ys.append(attention64(q[i], H, G, D, t, kc, t, vc, cells, 1.0 / np.sqrt(D)))
        out[f"attn_y_{name}"] = np.stack(ys)
    np.savez_compressed(os.path.join(HERE, "ops_v1.npz"), **out)
    print("wrote ops_v1.npz", {k: v.shape for k, v in out.items()})


E2E_CASES = [("tiny", "q4_k_m", "q8_0", 7), ("tiny-d128", "q5_k_m", "f16", 11), ("tiny-moe", "q4_k_m", "q8_0", 13),
             # round 3: general.architecture qwen2 (NEOX pairing, Q / K / V biases) and YaRN rope scaling from the file's metadata
             ("tiny-qwen2", "q4_k_m", "q8_0", 17), ("tiny-yarn", "q4_k_m", "q8_0", 19)]
# round 4 (e2e_v2.npz; v1 stays frozen): the type mix of the reference's smoke model (a Q2_K file of TinyLlama's layer geometry, /root/reference Makefile:5-6),
# a Q8_0 file with a q8_0 cache (BASELINE config 1's file type), a q4_0 cache (cache_type "q4_0", src/llama_engine.cc:272-285)
E2E_CASES_V2 = [("tiny-tl-2l", "q2_k", "f16", 23), ("tiny", "q8_0", "q8_0", 29), ("tiny-gqa4", "q4_k_m", "q4_0", 31)]
# ... and the encoder graph of the reference's embedding smoke model (nomic-bert): the last layer's hidden states of one 24-token sequence
ENC_CASE = ("tiny-nomic", "f16", 37, 24)
N_PROMPT, N_STEPS = 12, 32


def _load_synth():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gguf_synth", os.path.join(ROOT, "cortex.llamacpp_amd", "gguf_synth.py"))
    gs = importlib.util.module_from_spec(spec); sys.modules["gguf_synth"] = gs; spec.loader.exec_module(gs)
    return gs


def make_e2e(cases=None, fname="e2e_v1.npz", with_encoder=False):
    import oracle_py as oq
    gs = _load_synth()
    out = {}
    for cfg, ftype, kv, seed in (E2E_CASES if cases is None else cases):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.gguf")
            gs.write_synthetic_llama(path, cfg, ftype, seed=seed)
            om = oq.OracleModel(path)
            tkv = {"q8_0": oq.Q8_0, "f16": oq.F16, "q4_0": oq.Q4_0}[kv]
            oc = oq.OracleContext(om, 64, tkv, tkv, True, 2)
            n_vocab = gs.CONFIGS[cfg].n_vocab
            prompt = np.random.default_rng(seed).integers(0, n_vocab, N_PROMPT).astype(np.int32)
            rows = [oc.decode(prompt, np.arange(N_PROMPT))[0]]
            ids = []
            for s in range(N_STEPS):
                tok = int(rows[-1].argmax()); ids.append(tok)
                rows.append(oc.decode([tok], [N_PROMPT + s])[0])
            logits = np.stack(rows[:N_STEPS]).astype(np.float32)      # row s chose ids[s]
            top2 = np.sort(logits, axis=1)[:, -2:]
            key = f"{cfg}.{ftype}.{kv}"
            out[f"{key}.prompt"] = prompt
            out[f"{key}.ids"] = np.array(ids, np.int32)
            out[f"{key}.logits"] = logits
            out[f"{key}.top2_gap"] = (top2[:, 1] - top2[:, 0]).astype(np.float32)
            out[f"{key}.seed"] = np.array([seed], np.int32)
            oc.close(); om.close()
    if with_encoder:
        cfg, ftype, seed, n = ENC_CASE
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "m.gguf")
            gs.write_synthetic_llama(path, cfg, ftype, seed=seed)
            om = oq.OracleModel(path)
            oq.set_fa_v_acc_f32(1)         # (the f32-accumulating form of the restatement: what the HIP kernels are held to tightly)
            try:
                oc = oq.OracleContext(om, 64, oq.F16, oq.F16, True, 2)
                toks = np.random.default_rng(seed).integers(5, gs.CONFIGS[cfg].n_vocab, n).astype(np.int32)
                oc.decode(toks, np.arange(n), [0] * n, np.ones(n, np.int8))
                n_layer = gs.CONFIGS[cfg].n_layer
                out[f"{cfg}.{ftype}.enc.tokens"] = toks
                out[f"{cfg}.{ftype}.enc.hidden"] = oc.layer_out(n_layer - 1, n).reshape(n, -1).astype(np.float32)
                out[f"{cfg}.{ftype}.enc.seed"] = np.array([seed], np.int32)
                oc.close()
            finally:
                oq.set_fa_v_acc_f32(0)
            om.close()
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print("wrote", fname, {k: v.shape for k, v in out.items()})


def make_api_shapes():
    """Transcribed from the reference source (file:line beside each entry); values that vary per request are given as type names."""
    shapes = {
        "_source": "janhq/cortex.llamacpp src/llama_engine.cc",
        "status_keys": ["is_done", "has_error", "is_stream", "status_code"],                         # :19-22, 376-384
        "load_model_ok": {"status": {"is_done": True, "has_error": False, "is_stream": False, "status_code": 200},
                          "body": {"message": "Model loaded successfully"}},                         # :407-417
        "load_model_no_id": {"status": {"is_done": False, "has_error": True, "is_stream": False, "status_code": 400},
                             "body": {"message": "No model id found in request body"}},             # :374-385
        "load_model_again": {"status": {"is_done": True, "has_error": False, "is_stream": False, "status_code": 409},
                             "body": {"message": "Model already loaded"}},                          # :387-398
        "load_model_failed": {"status": {"is_done": False, "has_error": True, "is_stream": False, "status_code": 500},
                              "body": {"message": "Failed to load model"}},                         # :400-406
        "unload_model_ok": {"status": {"is_done": True, "has_error": False, "is_stream": False, "status_code": 200},
                            "body": {"message": "Model unloaded successfully"}},                    # :433-440
        "model_not_loaded": {"status": {"is_done": False, "has_error": True, "is_stream": False, "status_code": 409},
                             "body": {"message": "Model has not been loaded, please load model into cortex.llamacpp"}},   # :1226-1241
        "get_model_status_ok": {"status": {"is_done": True, "has_error": False, "is_stream": False, "status_code": 200},
                                "body": {"model_loaded": True, "model_data": "str"}},               # :447-466
        "get_models_body": {"object": "list", "data": [{"id": "str", "engine": "cortex.llamacpp", "start_time": "int", "model_size": "int", "vram": "int",
                                                        "ram": "int", "object": "model"}]},           # :468-500
        # CreateFullReturnJson :180-218 as called at :1081-1085 (model "_", system_fingerprint "_", finish_reason "stop")
        "chat_completion_body": {"id": "str", "object": "chat.completion", "created": "int", "model": "_", "system_fingerprint": "_",
                                 "choices": [{"index": 0, "message": {"role": "assistant", "content": "str"}, "finish_reason": "stop"}],
                                 "usage": {"prompt_tokens": "int", "completion_tokens": "int", "total_tokens": "int"}},
        "chat_completion_status": {"is_done": True, "has_error": False, "is_stream": False, "status_code": 200},          # :1103-1108
        # CreateReturnJson :220-270 as called at :969-974 (running chunk: model "_", finish_reason "") and :997-1001 (last chunk: content "", "stop")
        "chat_chunk_body": {"id": "str", "object": "chat.completion.chunk", "created": "int", "model": "_",
                            "choices": [{"index": 0, "delta": {"content": "str", "role": "assistant"}, "finish_reason": ""}]},
        "chat_chunk_last_body": {"id": "str", "object": "chat.completion.chunk", "created": "int", "model": "_",
                                 "choices": [{"index": 0, "delta": {"content": "", "role": "assistant"}, "finish_reason": "stop"}]},
        # include_usage: every chunk carries "usage": null, the last one has an empty choices array and the counts (:236-262, :985-1001)
        "chat_chunk_usage_body": {"id": "str", "object": "chat.completion.chunk", "created": "int", "model": "_", "choices": [],
                                  "usage": {"prompt_tokens": "int", "completion_tokens": "int", "total_tokens": "int",
                                            "completion_tokens_details": {"reasoning_tokens": 0}}},
        "stream_frame": {"data_prefix": "data: ", "data_suffix": "\n\n", "done": "data: [DONE]\n\n"},                  # :969-1009
        "stream_status": {"running": {"is_done": False, "has_error": False, "is_stream": True, "status_code": 200},       # :976-981
                          "final": {"is_done": True, "has_error": False, "is_stream": True, "status_code": 200},          # :1003-1008
                          "error": {"is_done": False, "has_error": True, "is_stream": True, "status_code": 200}},         # :1017-1024 (body: {"data": ""})
    }
    with open(os.path.join(HERE, "api_shapes_v1.json"), "w", encoding="utf-8") as f:
        json.dump(shapes, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote api_shapes_v1.json")


if __name__ == "__main__":
    which = sys.argv[1:] or ["ops", "e2e", "api"]
    if "ops" in which: make_ops()
    if "e2e" in which: make_e2e()
    if "e2e2" in which: make_e2e(E2E_CASES_V2, "e2e_v2.npz", with_encoder=True)
    if "api" in which: make_api_shapes()
