"""The CPU restatement's encoder graph (oracle/oq_llama.c bert_decode: general.architecture nomic-bert) against an independent float64 forward written here from the
published graph (llama.cpp llm_build_bert, NOMIC_BERT branches): token + type-0 embeddings -> LayerNorm; per layer fused Q | K | V, NEOX rope, bidirectional
softmax attention, attn_output, residual, LayerNorm, SwiGLU, residual, LayerNorm.  No golden vector for this graph exists in the mount (llama.cpp is an absent
submodule): this pins the restatement to the formula, not to the reference's bits ("parity unpinned" for the encoder)."""
import numpy as np
import pytest

import oracle_py as oq
from gguf_read import read_gguf


def _f64(t):
    ne, ty, raw = t
    n = int(np.prod(ne))
    if ty == 0:
        return raw[: n * 4].view(np.float32).astype(np.float64).reshape(ne[::-1])
    if ty == 1:
        return raw[: n * 2].view(np.float16).astype(np.float64).reshape(ne[::-1])
    return oq.dequantize(ty, raw[: oq.row_bytes(ty, ne[0]) * (n // ne[0])], n).astype(np.float64).reshape(ne[::-1])


def _layer_norm(x, w, b, eps):
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def _rope_neox(x, pos, base):          # x: [T, H, D]
    T, H, D = x.shape
    i = np.arange(D // 2)
    th = pos[:, None] * base ** (-2.0 * i / D)          # [T, D/2]
    c, s = np.cos(th)[:, None, :], np.sin(th)[:, None, :]
    a, b = x[..., : D // 2], x[..., D // 2:]
    return np.concatenate([a * c - b * s, a * s + b * c], axis=-1)


def bert_forward_f64(path, tokens, pos):
    kv, t = read_gguf(path)
    a = kv["general.architecture"]
    E, L, H = kv[f"{a}.embedding_length"], kv[f"{a}.block_count"], kv[f"{a}.attention.head_count"]
    eps, base = kv[f"{a}.attention.layer_norm_epsilon"], kv[f"{a}.rope.freq_base"]
    D = E // H
    x = _f64(t["token_embd.weight"])[tokens] + _f64(t["token_types.weight"])[0]
    x = _layer_norm(x, _f64(t["token_embd_norm.weight"]), _f64(t["token_embd_norm.bias"]), eps)
    outs = []
    T = len(tokens)
    for il in range(L):
        p = f"blk.{il}."
        qkv = x @ _f64(t[p + "attn_qkv.weight"]).T
        q, k, v = qkv[:, :E].reshape(T, H, D), qkv[:, E:2 * E].reshape(T, H, D), qkv[:, 2 * E:].reshape(T, H, D)
        q, k = _rope_neox(q, pos, base), _rope_neox(k, pos, base)
        s = np.einsum("thd,uhd->htu", q, k) / np.sqrt(D)            # every token sees every token
        w = np.exp(s - s.max(axis=-1, keepdims=True))
        w /= w.sum(axis=-1, keepdims=True)
        att = np.einsum("htu,uhd->thd", w, v).reshape(T, E)
        cur = att @ _f64(t[p + "attn_output.weight"]).T + x
        cur = _layer_norm(cur, _f64(t[p + "attn_output_norm.weight"]), _f64(t[p + "attn_output_norm.bias"]), eps)
        g, u = cur @ _f64(t[p + "ffn_gate.weight"]).T, cur @ _f64(t[p + "ffn_up.weight"]).T
        ff = (g / (1.0 + np.exp(-g)) * u) @ _f64(t[p + "ffn_down.weight"]).T
        x = _layer_norm(ff + cur, _f64(t[p + "layer_output_norm.weight"]), _f64(t[p + "layer_output_norm.bias"]), eps)
        outs.append(x)
    return outs


@pytest.mark.parametrize("cfg,ftype,tol", [("tiny-nomic", "f16", 2e-2), ("tiny-nomic", "q8_0", 6e-2)])
def test_encoder_restatement_matches_float64_graph(pkg, tmp_path, cfg, ftype, tol):
    path = str(tmp_path / "enc.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=5)
    rng = np.random.default_rng(2)
    n = 19
    tokens = rng.integers(5, pkg.gguf_synth.CONFIGS[cfg].n_vocab, n)
    pos = np.arange(n)
    want = bert_forward_f64(path, tokens, pos.astype(np.float64))
    om = oq.OracleModel(path)
    oc = oq.OracleContext(om, 64, oq.F16, oq.F16, True, 2)
    oc.decode(tokens, pos)
    for il, w in enumerate(want):
        got = oc.layer_out(il, n).reshape(n, -1)
        # f16 weights contract against f16-rounded activations and the K / V rows are f16 (Q8_0: int8 activations): LayerNorm'd rows of unit scale agree to ~1e-2
        assert np.abs(got - w).max() <= tol, (il, np.abs(got - w).max())
    # bidirectional: the FIRST token's row depends on the last token (a causal graph would not)
    tokens2 = tokens.copy(); tokens2[-1] = (tokens2[-1] + 7) % 500 + 5
    oc.kv_clear(); oc.decode(tokens2, pos)
    assert np.abs(oc.layer_out(0, n).reshape(n, -1)[0] - want[0][0]).max() > 10 * tol / 100
    oc.close(); om.close()
