"""GPU parity end to end through the C-ABI: synthetic GGUF -> mi355_model_load_from_file -> mi355_decode,
against the CPU oracle on the same file and tokens: per-layer residual stream, logits, greedy token ids, and
the KV-cache sequence operations the reference's slot loop uses."""
import numpy as np
import pytest

import oracle_py as oq

pytestmark = pytest.mark.gpu

KV = {"f16": 1, "q8_0": 8, "q4_0": 2}


@pytest.fixture(scope="module")
def be(pkg):
    return pkg.Backend()


def make(pkg, tmp_models, cfg, ftype, seed=11):
    path = str(tmp_models / f"{cfg}-{ftype}-{seed}.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    return path


def open_pair(pkg, path, n_ctx, kv, use_graphs=True, n_ubatch=512):
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=n_ctx, type_k=KV[kv], type_v=KV[kv], use_graphs=use_graphs, n_ubatch=n_ubatch)
    om = oq.OracleModel(path)
    oc = oq.OracleContext(om, n_ctx, KV[kv], KV[kv], True, 4)
    return m, c, om, oc


# logits tolerance: identical integer arithmetic; f32 reductions are re-associated and cosf/sinf/expf differ by ulps.
# f16 KV additionally differs by the CPU path's fp16 V accumulation (see test_gpu_ops.test_flash_attn).
TOL = {"q8_0": 1e-3, "q4_0": 1e-3, "f16": 1e-2}


@pytest.mark.parametrize("cfg,ftype,kv", [("tiny", "q4_k_m", "q8_0"), ("tiny", "q5_k_m", "f16"), ("tiny", "q8_0", "q8_0"),
                                          ("tiny-gqa4", "q4_k_m", "q8_0"), ("tiny-gqa4", "q4_k_m", "q4_0"), ("tiny", "f16", "f16"),
                                          ("tiny-moe", "q4_k_m", "q8_0"), ("tiny-moe", "q5_k_m", "f16")])
def test_prefill_layers_logits_and_greedy_ids(be, pkg, tmp_models, cfg, ftype, kv):
    path = make(pkg, tmp_models, cfg, ftype)
    m, c, om, oc = open_pair(pkg, path, 128, kv)
    rng = np.random.default_rng(5)
    n_prompt = 21
    prompt = rng.integers(0, m.n_vocab, n_prompt)
    c.enable_taps(True)
    c.decode(prompt, np.arange(n_prompt))
    ref = oc.decode(prompt, np.arange(n_prompt))[0]
    for il in range(m.n_layer):
        a, b = c.layer_out(il, n_prompt), oc.layer_out(il, n_prompt)
        assert np.abs(a - b).max() <= TOL[kv] * max(1.0, np.abs(b).max()), (il, np.abs(a - b).max())
    got = c.logits()
    assert np.abs(got - ref).max() <= TOL[kv] * max(1.0, np.abs(ref).max())
    c.enable_taps(False)
    tok = int(ref.argmax())
    assert int(got.argmax()) == tok == c.argmax()
    ids_g, ids_r = [], []
    tg = tr = tok
    for step in range(24):          # single-token steps run through the captured hipGraph
        c.decode([tg], [n_prompt + step])
        r = oc.decode([tr], [n_prompt + step])[0]
        g = c.logits()
        assert np.abs(g - r).max() <= TOL[kv] * max(1.0, np.abs(r).max()), step
        tg, tr = c.argmax(), int(r.argmax())
        ids_g.append(tg); ids_r.append(tr)
    assert ids_g == ids_r
    c.close(); m.close(); oc.close(); om.close()


def test_graph_and_eager_agree_bitwise(be, pkg, tmp_models):
    path = make(pkg, tmp_models, "tiny-gqa4", "q4_k_m")
    m = pkg.Model(path)
    outs = []
    for graphs in (True, False):
        c = pkg.Context(m, n_ctx=64, type_k=8, type_v=8, use_graphs=graphs)
        c.decode([1, 2, 3, 4, 5], np.arange(5))
        seq = []
        t = c.argmax()
        for s in range(8):
            c.decode([t], [5 + s])
            seq.append(c.logits().copy())
            t = c.argmax()
        outs.append(np.stack(seq))
        c.close()
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all()
    m.close()


def test_ubatch_split_and_multi_sequence(be, pkg, tmp_models):
    """n_tokens > n_ubatch is processed in micro-batches; two sequences share one batch (continuous batching)."""
    path = make(pkg, tmp_models, "tiny", "q4_k_m")
    m, c, om, oc = open_pair(pkg, path, 256, "q8_0", n_ubatch=16)
    rng = np.random.default_rng(9)
    a = rng.integers(0, m.n_vocab, 37)
    b = rng.integers(0, m.n_vocab, 11)
    toks = np.concatenate([a, b])
    pos = np.concatenate([np.arange(37), np.arange(11)])
    seq = np.array([0] * 37 + [1] * 11)
    flags = np.zeros(48, np.int8); flags[36] = 1; flags[47] = 1
    assert c.decode(toks, pos, seq, flags) == 0
    ref = oc.decode(toks, pos, seq, flags)
    for j, i in enumerate((36, 47)):
        g = c.logits(i)
        assert np.abs(g - ref[j]).max() <= 1e-3 * max(1.0, np.abs(ref[j]).max())
    # one decode step for both sequences in one batch
    t0, t1 = int(ref[0].argmax()), int(ref[1].argmax())
    assert c.decode([t0, t1], [37, 11], [0, 1], [1, 1]) == 0
    r2 = oc.decode([t0, t1], [37, 11], [0, 1], [1, 1])
    for j in range(2):
        assert np.abs(c.logits(j) - r2[j]).max() <= 1e-3 * max(1.0, np.abs(r2[j]).max())
    c.close(); m.close(); oc.close(); om.close()


@pytest.mark.parametrize("kv", ["f16", "q8_0"])
def test_kv_seq_ops_match_oracle(be, pkg, tmp_models, kv):
    """prompt-prefix reuse (seq_rm), seq_cp, and context shift (seq_rm + seq_add => K re-rotation), as
    LlamaServerContext::UpdateSlots drives them (reference llama_server_context.cc:1288-1291,1540-1547)."""
    path = make(pkg, tmp_models, "tiny-gqa4", "q4_k_m")
    m, c, om, oc = open_pair(pkg, path, 96, kv)
    rng = np.random.default_rng(3)
    p = rng.integers(0, m.n_vocab, 40)
    for x in (c, oc):
        x.decode(p, np.arange(40))
    # 1. drop the tail and re-evaluate a different continuation (prompt cache)
    assert c.kv_seq_rm(0, 25, -1) and oc.kv_seq_rm(0, 25, -1)
    q = rng.integers(0, m.n_vocab, 9)
    c.decode(q, 25 + np.arange(9)); r = oc.decode(q, 25 + np.arange(9))[0]
    assert np.abs(c.logits() - r).max() <= TOL[kv] * max(1.0, np.abs(r).max())
    assert c.kv_used() == 34
    # 2. context shift: discard positions [4, 14), slide the rest down by 10
    for x in (c, oc):
        x.kv_seq_rm(0, 4, 14)
        x.kv_seq_add(0, 14, 34, -10)
    c.decode([7], [24]); r = oc.decode([7], [24])[0]
    # K rows are re-rotated through a dequantise/requantise round trip on both sides
    assert np.abs(c.logits() - r).max() <= 3 * TOL[kv] * max(1.0, np.abs(r).max())
    # 3. fork the sequence and continue the copy
    for x in (c, oc):
        x.kv_seq_cp(0, 1, 0, -1)
    c.decode([9], [25], [1]); r = oc.decode([9], [25], [1])[0]
    assert np.abs(c.logits() - r).max() <= 3 * TOL[kv] * max(1.0, np.abs(r).max())
    # 4. cache full -> llama_decode returns 1 (caller halves n_batch), state unchanged
    used = c.kv_used()
    assert c.decode(rng.integers(0, m.n_vocab, 96), np.arange(96) + 100) == 1
    assert c.kv_used() == used
    c.kv_clear()
    assert c.kv_used() == 0
    c.close(); m.close(); oc.close(); om.close()


def test_load_errors(be, pkg, tmp_path):
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"NOPE" + b"\0" * 64)
    with pytest.raises(pkg.MI355Error):
        pkg.Model(str(bad))
    with pytest.raises(pkg.MI355Error):
        pkg.Model(str(tmp_path / "missing.gguf"))
