"""GPU parity end to end through the C-ABI: synthetic GGUF -> mi355_model_load_from_file -> mi355_decode,
against the CPU oracle on the same file and tokens: per-layer residual stream, logits, greedy token ids, and
the KV-cache sequence operations the reference's slot loop uses."""
import os

import numpy as np
import pytest

import oracle_py as oq

pytestmark = pytest.mark.gpu

KV = {"f16": 1, "q8_0": 8, "q4_0": 2}


@pytest.fixture(scope="module")
def be(pkg):
    return pkg.Backend()


# test_prefill_layers_logits_and_greedy_ids: the cases whose 24 teacher-forced steps hold TWO near ties of the CPU logits that the HIP path resolves the other
# way (every other case: at most one) -> the largest relative top-2 gap such a flip may have (measured gaps in the comment; FLIP_TOL bounds any flip at 6e-2)
TWO_FLIP_CASES = {
    ("tiny-qwen2-1.5b-2l:70", "q4_k_m", "f16"): 1e-2,      # steps 6 and 15: gaps 7.8e-3 and 4.4e-3 of the logit scale (round 5, gpurun_out/r5_flips.txt)
    ("tiny-r3:40", "q4_k_m", "q8_0"): 1e-2,                # steps 5 and 7: gaps 2.5e-4 and 4.8e-3
}


def make(pkg, tmp_models, cfg, ftype, seed=11):
    path = str(tmp_models / f"{cfg}-{ftype}-{seed}.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    return path


def open_pair(pkg, path, n_ctx, kv, use_graphs=True, n_ubatch=512):
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=n_ctx, type_k=KV[kv], type_v=KV[kv], use_graphs=use_graphs, n_ubatch=n_ubatch)
    om = oq.OracleModel(path)
    oc = oq.OracleContext(om, n_ctx, KV[kv], KV[kv], True, oq.threads())
    return m, c, om, oc


# End-to-end tolerance, calibrated on the CPU side (tests/test_oracle_sensitivity.py): re-associating the f32 sums of
# the CPU restatement itself leaves the logits unchanged to ~1e-6 on most steps, but whenever a 1-ulp difference flips
# one int8 rounding (activation or KV code) the logits of these tiny models jump by up to ~1e-2 relative.  So: every
# layer and step within FLIP_TOL (a flip that lands in the KV cache persists for the rest of the run), and the best-agreeing
# full forward within TIGHT_TOL (before the first flip the two implementations agree to f32 round-off).  The tight
# evidence is per op (tests/test_gpu_ops.py: integers exact, floats <= 2e-5).  An f16 V cache is accumulated in fp16 by
# the CPU path, which makes the CPU result itself noisy at 1e-2; it is therefore compared with that switched off, and
# loosely with it on.
FLIP_TOL = 3e-2
TIGHT_TOL = 2e-5


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("cfg,ftype,kv", [("tiny", "q4_k_m", "q8_0"), ("tiny", "q5_k_m", "f16"), ("tiny", "q8_0", "q8_0"),
                                          ("tiny-gqa4", "q4_k_m", "q8_0"), ("tiny-gqa4", "q4_k_m", "q4_0"), ("tiny", "f16", "f16"),
                                          ("tiny-gqa4", "q5_k_m", "f16"),
                                          ("tiny-d128", "q4_k_m", "q8_0"), ("tiny-d128", "q4_k_m", "f16"),
                                          ("tiny-d128-mha", "q5_k_m", "q8_0"), ("tiny-d128", "q8_0", "f16"),
                                          ("tiny-moe", "q4_k_m", "q8_0"), ("tiny-moe", "q5_k_m", "f16"),
                                          # E = 2048: persistent mat-vec with fused RMSNorm / quantise prologues on the decode
                                          # steps; 40-token prompts go through the MFMA prefill contraction (planes form)
                                          ("tiny-e2048", "q4_k_m", "q8_0"), ("tiny-e2048", "q5_k_m", "q8_0"),
                                          ("tiny-e2048:40", "q4_k_m", "q8_0"), ("tiny-d128:40", "q4_k_m", "q8_0"),
                                          ("tiny-gqa4:40", "q5_k_m", "f16"),
                                          # 8 x 128 KV heads: single-launch decode attention (rope + store + merge fused)
                                          ("tiny-g8", "q4_k_m", "q8_0"), ("tiny-g8", "q4_k_m", "f16"), ("tiny-g8:70", "q4_k_m", "q8_0"),
                                          # Llama-3-8B's layer geometry (4096 / 14336, 32 heads over 8 KV heads)
                                          ("tiny-8b-2l", "q4_k_m", "q8_0"), ("tiny-8b-2l:70", "q5_k_m", "f16"),
                                          # Llama-3-70B's feed-forward width (28672): the single-token ffn_down as two column halves on the weight stream (Q4_K and Q6_K halves)
                                          ("tiny-ff28k", "q4_k_m", "q8_0"), ("tiny-ff28k:40", "q6_k", "q8_0"),
                                          # f16 cache, 40-token prompts: the matrix-core prompt attention with f16 K / V (GQA 4:1 and MHA)
                                          ("tiny-d128:40", "q4_k_m", "f16"), ("tiny-d128-mha:40", "q5_k_m", "f16"),
                                          # Q2_K / Q3_K_M files (Q2_K + Q3_K + Q4_K / Q5_K + Q6_K tensors side by side): the type mix of the reference's smoke model
                                          ("tiny-gqa4", "q2_k", "q8_0"), ("tiny-gqa4", "q3_k_m", "f16"), ("tiny-e2048", "q2_k", "q8_0"), ("tiny-d128:40", "q3_k_m", "q8_0"),
                                          ("tiny-d128-mha", "q2_k", "f16"),
                                          # the rest of the quantisation mixes the reference publishes (convert-model-all-quant.yml:106-152)
                                          ("tiny-gqa4", "q3_k_s", "q8_0"), ("tiny-d128", "q3_k_l", "q8_0"), ("tiny-gqa4", "q4_k_s", "f16"), ("tiny-d128", "q5_k_s", "q8_0"),
                                          ("tiny-gqa4", "q6_k", "q8_0"), ("tiny-g8", "q3_k_l", "q8_0"), ("tiny-8b-2l", "q6_k", "q8_0"), ("tiny-8b-2l", "q3_k_s", "q8_0"),
                                          # 32-element weight formats: Q4_0, Q5_0, IQ4_NL files (Q8_0 activations; IQ4_NL with its Q5_K promotions)
                                          ("tiny-gqa4", "q4_0", "q8_0"), ("tiny-d128", "q5_0", "q8_0"), ("tiny-gqa4", "iq4_nl", "f16"), ("tiny-d128:40", "iq4_nl", "q8_0"),
                                          ("tiny-e2048", "q4_0", "q8_0"), ("tiny-8b-2l", "q5_0", "q8_0"),
                                          # TinyLlama-1.1B's layer geometry in the type mix of the reference's smoke model (a Q2_K file, Makefile:5) and neighbours
                                          ("tiny-tl-2l", "q2_k", "f16"), ("tiny-tl-2l", "q2_k", "q8_0"), ("tiny-tl-2l", "q3_k_m", "q8_0"), ("tiny-tl-2l", "q4_k_m", "f16"),
                                          ("tiny-tl-2l", "q8_0", "f16"), ("tiny-tl-2l:40", "q5_0", "q8_0"),
                                          # general.architecture "qwen2" (a weekend-test family of the reference): NEOX rope pairing, Q / K / V biases; head_dim 64 and the
                                          # layer geometries of Qwen2-1.5B (6 query heads per kv head) and Qwen2-7B (7; hidden 3584 and feed-forward 18944: no multiple of 1024)
                                          ("tiny-qwen2", "q4_k_m", "q8_0"), ("tiny-qwen2:40", "q5_k_m", "f16"), ("tiny-qwen2", "q8_0", "q8_0"),
                                          ("tiny-qwen2-1.5b-2l", "q4_k_m", "q8_0"), ("tiny-qwen2-7b-2l:40", "q4_k_m", "q8_0"), ("tiny-qwen2-7b-2l", "q4_0", "f16"),
                                          ("tiny-qwen2-1.5b-2l:70", "q4_k_m", "f16"),
                                          # 3 and 5 query heads per kv head: matrix-core prompt attention (40 / 70 tokens) and the single-launch decode attention, both cache types
                                          # q4_0 cache (cache_type "q4_0"): the matrix-core prompt attention unpacks the nibbles while staging (head_dim 128; ratios 4, 2, 1)
                                          ("tiny-d128:40", "q4_k_m", "q4_0"), ("tiny-g8:70", "q4_k_m", "q4_0"), ("tiny-d128-mha:40", "q5_k_m", "q4_0"), ("tiny-8b-2l:70", "q4_k_m", "q4_0"),
                                          # two query heads of 64 per kv head on the matrix-core prompt attention (its R = 2, head_dim 64 form), both cache types
                                          ("tiny:40", "q4_k_m", "q8_0"), ("tiny:70", "q5_k_m", "f16"),
                                          ("tiny-r3:40", "q4_k_m", "q8_0"), ("tiny-r3:70", "q5_k_m", "f16"), ("tiny-r5:40", "q4_k_m", "f16"), ("tiny-r5:50", "q4_k_m", "q8_0")])
def test_prefill_layers_logits_and_greedy_ids(be, pkg, tmp_models, cfg, ftype, kv):
    cfg, _, np_s = cfg.partition(":")
    path = make(pkg, tmp_models, cfg, ftype)
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)     # see the note above; the stock fp16 path is checked loosely below
    try:
        m, c, om, oc = open_pair(pkg, path, 128, kv)
        rng = np.random.default_rng(5)
        n_prompt = int(np_s) if np_s else 21
        prompt = rng.integers(0, m.n_vocab, n_prompt)
        c.enable_taps(True)
        c.decode(prompt, np.arange(n_prompt))
        ref = oc.decode(prompt, np.arange(n_prompt))[0]
        errs = []
        for il in range(m.n_layer):
            errs.append(rel_err(c.layer_out(il, n_prompt), oc.layer_out(il, n_prompt)))
        # per-token view of the first layer: a rounding flip hits single tokens (and, through the KV cache, the tokens
        # after it in deeper layers); everything else must agree to f32 round-off
        a0 = c.layer_out(0, n_prompt).reshape(n_prompt, -1)
        b0 = oc.layer_out(0, n_prompt).reshape(n_prompt, -1)
        tok_err0 = np.abs(a0 - b0).max(axis=1) / max(1.0, float(np.abs(b0).max()))
        got = c.logits()
        errs.append(rel_err(got, ref))
        assert max(errs) <= FLIP_TOL, errs
        c.enable_taps(False)
        tok = int(ref.argmax())
        step_err = [errs[-1]]
        mism = 0
        gaps = []
        for step in range(24):          # single-token steps run through the captured hipGraph; teacher-forced with the CPU token
            c.decode([tok], [n_prompt + step])
            r = oc.decode([tok], [n_prompt + step])[0]
            g = c.logits()
            step_err.append(rel_err(g, r))
            tok = int(r.argmax())
            if c.argmax() != tok:        # only acceptable at a near tie of the CPU logits
                top2 = np.sort(r)[-2:]
                assert top2[1] - top2[0] <= 2 * FLIP_TOL * max(1.0, np.abs(r).max()), (step, top2)
                mism += 1
                gaps.append((step, float((top2[1] - top2[0]) / max(1.0, np.abs(r).max()))))
            assert int(g.argmax()) == c.argmax()
        assert max(step_err) <= FLIP_TOL, step_err
        if kv != "f16" and ftype != "f16":   # f16 weights / f16 K rows add an f16 rounding per element: never flip-free
            # before the first rounding flip HIP == CPU to f32 round-off: (i) the typical token of the first layer (median over the prompt's tokens) and
            # (ii) at least two thirds of its tokens individually (measured over the 38 cases: 0.71 - 1.0; medians 6e-8 - 2e-7) - both, not either
            if ftype in ("q8_0", "q4_0", "q5_0", "iq4_nl"):
                # Q8_0 activation blocks: 8x as many blocks per vector as Q8_K, each with an f16-rounded scale - 8x as many roundings that a 1-ulp
                # difference can flip, and 8x as many f32 terms to re-associate (measured at E = 4096 / FF = 14336: single token 4e-7 .. 1.5e-5,
                # 21 tokens: medians 1e-5 .. 1.4e-3, Q8_0 files included): the typical token is held to a tenth of the flip tolerance
                assert float(np.median(tok_err0)) <= FLIP_TOL / 10, (errs, step_err, tok_err0)
            else:
                assert float(np.median(tok_err0)) <= TIGHT_TOL, (errs, step_err, tok_err0)
                assert int((tok_err0 <= TIGHT_TOL).sum()) * 3 >= 2 * n_prompt, (errs, step_err, tok_err0)
        # every mismatch was checked above to be a near tie of the CPU logits (gap <= 2 FLIP_TOL); with 512-entry random vocabularies such ties are common,
        # and which of them flip depends on the f32 association.  At most ONE per run - except the cases named in TWO_FLIP_CASES, each with the steps and
        # the relative top-2 gaps of the CPU logits measured when it was listed (a case that starts to flip elsewhere, or more often, fails here)
        if os.environ.get("MI355_TEST_RECORD_FLIPS") and mism:
            with open(os.environ["MI355_TEST_RECORD_FLIPS"], "a") as f:
                f.write(f"{cfg}:{np_s} {ftype} {kv} mism={mism} gaps={gaps}\n")
        allowed = TWO_FLIP_CASES.get((f"{cfg}:{np_s}" if np_s else cfg, ftype, kv))
        if allowed is None:
            assert mism <= 1, (mism, gaps)
        else:
            assert mism <= 2 and all(g <= allowed for _, g in gaps), (mism, gaps)
        c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


def test_greedy_steps_in_one_call_equal_the_step_by_step_loop(be, pkg, tmp_models):
    """mi355_greedy_steps (decode, logits row host-visible, arg-max fed back, n times on the C side) names the tokens of the same loop driven call by call"""
    path = make(pkg, tmp_models, "tiny-d128", "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(3).integers(0, m.n_vocab, 30)
    outs = []
    for mode in ("calls", "one"):
        c = pkg.Context(m, n_ctx=256, type_k=KV["q8_0"], type_v=KV["q8_0"])
        assert c.decode(prompt, np.arange(30)) == 0
        tok = c.argmax()
        if mode == "one":
            toks = [int(t) for t in c.greedy_steps(tok, 30, 40)]
        else:
            toks = []
            for i in range(40):
                assert c.decode([tok], [30 + i]) == 0
                c.logits_ready()
                tok = c.argmax()
                toks.append(int(tok))
        outs.append(toks)
        c.close()
    assert outs[0] == outs[1]
    m.close()


@pytest.mark.parametrize("cfg,ftype,kv,n_img,ubatch", [("tiny-d128", "q4_k_m", "q8_0", 70, 512), ("tiny-8b-2l", "q4_k_m", "q8_0", 300, 128), ("tiny-gqa4", "q5_k_m", "f16", 33, 512),
                                                       ("tiny-d128", "q4_k_m", "q8_0", 1, 512)])
def test_embeddings_batch_matches_oracle(be, pkg, tmp_models, cfg, ftype, kv, n_img, ubatch):
    """llama_batch.embd: a batch whose rows are embeddings instead of token ids - how the reference hands image embeddings to the model between the prefix and
    the suffix of a LLaVA prompt (llama_server_context.cc:1093-1107: llava_embd_batch at positions n_past ..).  A token prefix, n_img embedding rows (one
    micro-batch, several micro-batches of 128, a single row), a token suffix and teacher-forced steps, against the oracle fed the same rows."""
    path = make(pkg, tmp_models, cfg, ftype)
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    try:
        m, c, om, oc = open_pair(pkg, path, 512, kv, n_ubatch=ubatch)
        rng = np.random.default_rng(n_img)
        prefix = rng.integers(0, m.n_vocab, 9)
        suffix = rng.integers(0, m.n_vocab, 6)
        img = (rng.standard_normal((n_img, m.n_embd)) * 0.05).astype(np.float32)
        c.decode(prefix, np.arange(9)); oc.decode(prefix, np.arange(9))
        pos = np.arange(9, 9 + n_img)
        c.decode_embd(img, pos)
        ref = oc.decode_embd(img, pos)[0]
        errs = [rel_err(c.logits(), ref)]
        p0 = 9 + n_img
        c.decode(suffix, np.arange(p0, p0 + 6))
        ref = oc.decode(suffix, np.arange(p0, p0 + 6))[0]
        errs.append(rel_err(c.logits(), ref))
        tok = int(ref.argmax())
        for step in range(6):
            c.decode([tok], [p0 + 6 + step])
            r = oc.decode([tok], [p0 + 6 + step])[0]
            errs.append(rel_err(c.logits(), r))
            tok = int(r.argmax())
        assert max(errs) <= FLIP_TOL, errs
        if n_img == 1:                                                        # few enough roundings for a flip-free pass: f32 round-off
            assert min(errs) <= TIGHT_TOL, errs
        # a batch is token ids or embeddings; an encoder takes ids only; rows must be n_embd wide
        with pytest.raises(ValueError):
            c.decode_embd(img[:, :-1], pos)
        c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


@pytest.mark.parametrize("cfg,ftype,n_prompt", [("tiny-d128", "q4_k_m", 40), ("tiny-8b-2l", "q4_k_m", 70), ("tiny-tl-2l", "q8_0", 21), ("tiny-d128-mha", "q5_k_m", 300)])
def test_f16_cache_parity_mode_follows_the_stock_cpu_path(be, pkg, tmp_models, cfg, ftype, n_prompt):
    """The reference's default cache is f16 (llama_engine.cc:628-637) and its CPU path accumulates V in FP16, cell by cell.  Every other f16-cache test compares
    with the oracle's f32-accumulating restatement (tight) and with the stock mode loosely; this one switches the HIP side to the cell-by-cell kernel (option
    "fa_v_acc_f16", attn.hip flash_attn_v16_kernel) and compares with the oracle in its STOCK mode: the prompt's logits and 12 teacher-forced steps, through
    the context API (prompt batches and graph-captured single-token steps both take the kernel).  The tight evidence for the kernel is per op
    (test_flash_attn_f16_cache_parity_mode: <= 8e-5 where the parallel kernels sit at 2e-3 .. 4e-3): end to end, an attention output that is off by 1e-5 still
    flips Q8_K codes of the attn_output input, and a flip moves these tiny models' logits by ~1e-2 whatever its cause - so the logits are held to the flip
    tolerance and must be no further from the stock CPU path than the default kernels' (measured medians 0.0075 - 0.011 against 0.012 - 0.013)."""
    path = make(pkg, tmp_models, cfg, ftype)
    oq.set_fa_v_acc_f32(0)
    prompt = np.random.default_rng(11).integers(0, 512, n_prompt)

    def run(mode):
        be.set_option("fa_v_acc_f16", mode)
        try:
            m, c, om, oc = open_pair(pkg, path, 512, "f16")
            errs = []
            c.decode(prompt, np.arange(n_prompt))
            ref = oc.decode(prompt, np.arange(n_prompt))[0]
            errs.append(rel_err(c.logits(), ref))
            tok = int(ref.argmax())
            for step in range(12):
                c.decode([tok], [n_prompt + step])
                r = oc.decode([tok], [n_prompt + step])[0]
                errs.append(rel_err(c.logits(), r))
                tok = int(r.argmax())
            c.close(); m.close(); oc.close(); om.close()
            return errs
        finally:
            be.set_option("fa_v_acc_f16", -1)
    e_par = run(1)
    e_def = run(0)
    if os.environ.get("MI355_TEST_RECORD_FLIPS"):
        with open(os.environ["MI355_TEST_RECORD_FLIPS"], "a") as f:
            f.write(f"f16 parity mode {cfg} {ftype} {n_prompt}: parity max {max(e_par):.3g} median {float(np.median(e_par)):.3g}   default max {max(e_def):.3g} median {float(np.median(e_def)):.3g}\n")
    assert max(e_par) <= FLIP_TOL and max(e_def) <= FLIP_TOL, (e_par, e_def)
    assert float(np.median(e_par)) <= 1.1 * float(np.median(e_def)), (e_par, e_def)


@pytest.mark.parametrize("cfg,ftype,kv,n_prompt,ubatch", [("tiny-d128", "q4_k_m", "q8_0", 300, 512), ("tiny-d128-mha", "q5_k_m", "f16", 300, 512),
                                                          ("tiny-g8", "q4_k_m", "q8_0", 330, 512), ("tiny-d128", "q4_k_m", "q8_0", 300, 128),
                                                          ("tiny-d128-mha", "q4_k_m", "q8_0", 200, 64),
                                                          # Llama-3-8B's layer shapes at 130 tokens: gate | up (>= 8192 rows) on the LDS-form SwiGLU launch,
                                                          # Q | K | V, attn_output and ffn_down on the K-split kernel (kernel choice by shape)
                                                          ("tiny-8b-2l", "q4_k_m", "q8_0", 130, 512),
                                                          # and at 512 tokens, the headline prompt: every contraction on the LDS-form kernels (both operands
                                                          # through LDS, SwiGLU launch, split K for ffn_down), the prompt attention with 16 query tiles
                                                          ("tiny-8b-2l", "q4_k_m", "q8_0", 512, 512),
                                                          # Q2_K / Q3_K files: prompts on the matrix cores through their plane sets (TinyLlama's geometry: the reference's
                                                          # smoke model, and the 8B layer shapes)
                                                          ("tiny-tl-2l", "q2_k", "f16", 200, 512), ("tiny-tl-2l", "q3_k_m", "q8_0", 130, 512), ("tiny-8b-2l", "q2_k", "q8_0", 96, 512),
                                                          ("tiny-8b-2l", "q3_k_s", "q8_0", 160, 512),
                                                          # Q4_0 / IQ4_NL files: prompts on the Q8_0 matrix-core kernel through exact Q8_0-layout copies
                                                          ("tiny-tl-2l", "q4_0", "q8_0", 200, 512), ("tiny-d128", "iq4_nl", "f16", 130, 512)])
def test_long_prompt_logits_match_oracle(be, pkg, tmp_models, cfg, ftype, kv, n_prompt, ubatch):
    """Prompts of a few hundred tokens against the CPU restatement: the matrix-core prompt attention with several query tiles,
    its two key halves per workgroup, the query sub-tiles of one- and two-head kv groups (R = 1: four, R = 2: two), key splits across
    workgroups, and micro-batches that attend to the cells of earlier ones; the last-layer output of every token and the logits."""
    path = make(pkg, tmp_models, cfg, ftype)
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    try:
        m, c, om, oc = open_pair(pkg, path, 512, kv, n_ubatch=ubatch)
        prompt = np.random.default_rng(31).integers(0, m.n_vocab, n_prompt)
        c.enable_taps(True)
        assert c.decode(prompt, np.arange(n_prompt)) == 0
        ref = oc.decode(prompt, np.arange(n_prompt))[0]
        if ubatch >= n_prompt:                                # (the taps hold one micro-batch)
            # first layer, per token: a rounding flip of one K / V code reaches the tokens after it through the attention, so the
            # bulk of the tokens - not every one - agrees to f32 round-off; the last layer within the flip tolerance
            for il, tight in ((0, kv != "f16"), (m.n_layer - 1, False)):
                a0 = c.layer_out(il, n_prompt).reshape(n_prompt, -1)
                b0 = oc.layer_out(il, n_prompt).reshape(n_prompt, -1)
                tok_err = np.abs(a0 - b0).max(axis=1) / max(1.0, float(np.abs(b0).max()))
                assert float(tok_err.max()) <= FLIP_TOL, (il, int(tok_err.argmax()), float(tok_err.max()))
                if tight and ftype.startswith(("q2_k", "q3_k")):
                    # a Q2_K / Q3_K embedding table holds few distinct values per row, so the normalised inputs sit on or near rounding ties far more often than
                    # a 4-bit table's do and a 1-ulp difference in the RMSNorm scale flips more codes (measured: ~half the tokens of a 96-token prompt carry a
                    # flip at E = 4096, and a flipped K / V code reaches every later token through the attention: 24 % exact at 300 tokens): the tokens are
                    # either exact to f32 round-off or flips away - at least a fifth must be exact
                    assert int((tok_err <= TIGHT_TOL).sum()) * 5 >= n_prompt, (il, float(np.median(tok_err)), tok_err[:40])
                elif tight and ftype in ("q8_0", "q4_0", "q5_0", "iq4_nl"):
                    # (Q8_0 activation blocks: 8x the blocks and f16 scales per vector - see test_prefill_layers_logits_and_greedy_ids)
                    assert float(np.median(tok_err)) <= FLIP_TOL / 10, (il, float(np.median(tok_err)), tok_err[:40])
                elif tight:
                    assert float(np.median(tok_err)) <= TIGHT_TOL, (il, float(np.median(tok_err)), tok_err[:40])
        assert rel_err(c.logits(), ref) <= FLIP_TOL
        c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


def test_f16_cache_vs_stock_fp16_accumulation(be, pkg, tmp_models):
    """Against the unmodified CPU restatement (V accumulated in fp16) the f16-cache path agrees to the noise level of
    that fp16 accumulation (measured CPU-vs-CPU: ~1e-2)."""
    path = make(pkg, tmp_models, "tiny-gqa4", "q4_k_m")
    m, c, om, oc = open_pair(pkg, path, 128, "f16")
    rng = np.random.default_rng(5)
    prompt = rng.integers(0, m.n_vocab, 21)
    c.decode(prompt, np.arange(21))
    ref = oc.decode(prompt, np.arange(21))[0]
    assert rel_err(c.logits(), ref) <= 5e-2
    c.close(); m.close(); oc.close(); om.close()


@pytest.mark.parametrize("cfg", ["tiny-d128", "tiny-g8"])
@pytest.mark.parametrize("graphs", [True, False])
def test_logits_do_not_depend_on_the_sequence_region(be, pkg, tmp_models, cfg, graphs):
    """With n_seq_max > 1 every sequence owns a 64-aligned region of the cache and the single-token steps walk per-token
    chunk lists: a sequence must produce bit-identical logits whichever region it lives in, and the same as a
    one-sequence context (same chunk partition, same merge order).  70 steps cross a chunk boundary."""
    path = make(pkg, tmp_models, cfg, "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(3).integers(0, m.n_vocab, 30)

    def run(n_seq_max, seq):
        c = pkg.Context(m, n_ctx=1024, n_seq_max=n_seq_max, type_k=8, type_v=8, use_graphs=graphs)
        assert c.decode(prompt, np.arange(30), seq=seq) == 0
        rows = [c.logits().copy()]
        for s in range(70):
            assert c.decode([int(rows[-1].argmax())], [30 + s], seq=seq) == 0
            rows.append(c.logits().copy())
        c.close()
        return np.stack(rows)

    base = run(1, 0)
    for seq in (0, 1, 3):
        got = run(4, seq)
        if cfg == "tiny-g8":
            assert np.array_equal(got, base), (seq, int(np.argmax(np.abs(got - base).max(axis=1) > 0)))
        else:   # 2 x 128 KV heads take the general split-by-length kernel: the split count follows the highest occupied cell
            assert rel_err(got, base) <= FLIP_TOL, (seq, rel_err(got, base))
    m.close()


@pytest.mark.parametrize("cfg,kv", [("tiny-g8", 8), ("tiny-g8", 1), ("tiny-8b-2l", 8)])
def test_prompt_rope_and_kv_store_vectorised(be, pkg, tmp_models, cfg, kv):
    """Prompt batches rotate q in place and rotate / quantise / store K and V four elements per thread (a grid row per 1024
    elements of a token) instead of one workgroup per token with byte stores: the same bits in q and in the cache - logits of the
    prompt and of ten decode steps over that cache are identical."""
    path = make(pkg, tmp_models, cfg, "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(23).integers(0, m.n_vocab, 150)

    def run(fast):
        be.set_option("rope_fast", fast)
        try:
            c = pkg.Context(m, n_ctx=512, type_k=kv, type_v=kv)
            assert c.decode(prompt, np.arange(150)) == 0
            rows = [c.logits().copy()]
            for s_ in range(10):
                assert c.decode([int(rows[-1].argmax())], [150 + s_]) == 0
                rows.append(c.logits().copy())
            c.close()
        finally:
            be.set_option("rope_fast", 1)
        return np.stack(rows)

    a, b = run(1), run(0)
    assert np.isfinite(a).all()
    assert np.array_equal(a, b)
    m.close()


@pytest.mark.parametrize("kv", [8, 1])
def test_batched_step_stores_kv_inside_the_attention_launch(be, pkg, tmp_models, kv):
    """A continuous-batching step (one token each of different sequences) rotates and stores its K / V rows inside the attention
    launch - the workgroup whose chunk holds a token's new cell writes it - instead of in a launch of its own before it.  Five
    sequences with prompts of different lengths, 70 batched steps (the cells cross chunk boundaries), q8_0 and f16 cache: the
    same bits as with the separate store; a step with two tokens of ONE sequence keeps the separate store (the second token
    reads the first one's cell) and agrees with two single-token steps' cache content."""
    path = make(pkg, tmp_models, "tiny-g8", "q4_k_m")
    m = pkg.Model(path)
    rng = np.random.default_rng(21)
    lens = [30, 5, 64, 17, 41]
    prompts = [rng.integers(0, m.n_vocab, n) for n in lens]

    def run(fuse):
        be.set_option("attn_store_fuse", fuse)
        try:
            c = pkg.Context(m, n_ctx=2048, n_seq_max=8, type_k=kv, type_v=kv)
            for sq, p in enumerate(prompts):
                assert c.decode(p, np.arange(len(p)), seq=[sq] * len(p)) == 0
            toks = [int(p[-1]) for p in prompts]
            rows = []
            for step in range(70):
                assert c.decode(toks, [n + step for n in lens], seq=list(range(5)), logits=[1] * 5) == 0
                lg = np.stack([c.logits(i).copy() for i in range(5)])
                rows.append(lg)
                toks = [int(r.argmax()) for r in lg]
            # two tokens of one sequence in one step (not a distinct-sequence batch)
            assert c.decode([toks[0], toks[1]], [lens[0] + 70, lens[0] + 71], seq=[0, 0], logits=[1, 1]) == 0
            rows.append(np.stack([c.logits(0).copy(), c.logits(1).copy()]))
            c.close()
        finally:
            be.set_option("attn_store_fuse", 1)
        return rows

    a, b = run(1), run(0)
    for x, y in zip(a, b):
        assert np.isfinite(x).all()
        assert np.array_equal(x, y)
    m.close()


def _need_experiments(be, option):
    """The whole-step kernel and the layer engine are experiments: the product library is built without them (cortex.llamacpp_amd/build.py,
    MI355_BUILD_EXPERIMENTS=1 puts them back) and refuses to switch them on."""
    try:
        be.set_option(option, 1)
    except Exception as e:                      # the library says which build holds them
        assert "MI355_BUILD_EXPERIMENTS" in str(e), e
        pytest.skip("experiment kernels are not in this build")
    be.set_option(option, 0 if option == "decode_mega" else -1)


@pytest.mark.parametrize("kv", ["q8_0", "f16"])
@pytest.mark.parametrize("graphs", [True, False])
def test_mega_step_matches_per_launch_bitwise(be, pkg, tmp_models, kv, graphs):
    """The whole-step kernel (all layers of a single-token step in one launch, device-wide barriers between the phases)
    runs the same arithmetic as the per-launch path: logits must agree bit for bit, step after step, with one sequence
    and with sequence regions / chunk lists in use.  100 steps cross a 64-cell chunk boundary."""
    _need_experiments(be, "decode_mega")
    path = make(pkg, tmp_models, "tiny-8b-2l", "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(9).integers(0, m.n_vocab, 40)

    def run(mega, n_seq_max, seq):
        be.set_option("decode_mega", 1 if mega else 0)
        be.set_option("attn_out_fused", 0)      # the whole-step kernel holds the two-launch attention + attn_output arithmetic: compare like with like
        try:
            c = pkg.Context(m, n_ctx=512, n_seq_max=n_seq_max, type_k=KV[kv], type_v=KV[kv], use_graphs=graphs)
            assert c.decode(prompt, np.arange(40), seq=seq) == 0
            rows = [c.logits().copy()]
            for s in range(100):
                assert c.decode([int(rows[-1].argmax())], [40 + s], seq=seq) == 0
                rows.append(c.logits().copy())
            assert c.mega_steps() == (100 if mega else 0)      # the path under test really ran
            c.close()
        finally:
            be.set_option("decode_mega", 0)
            be.set_option("attn_out_fused", -1)
        return np.stack(rows)

    for n_seq_max, seq in ((1, 0), (4, 2)):
        a, b = run(True, n_seq_max, seq), run(False, n_seq_max, seq)
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), (n_seq_max, int(np.argmax(np.abs(a - b).max(axis=1) > 0)))
    m.close()


@pytest.mark.parametrize("cfg,ftype,kv,graphs", [("tiny-8b-3l", "q4_k_m", "q8_0", True), ("tiny-8b-3l", "q4_k_m", "q8_0", False), ("tiny-8b-2l", "q5_k_m", "f16", True),
                                                 ("tiny-8b-3l", "q5_k_m", "q8_0", True)])
def test_layer_engine_matches_per_launch_bitwise(be, pkg, tmp_models, cfg, ftype, kv, graphs):
    """The layer engine (decode_engine.hip: attn_output -> gate | up -> down -> next Q | K | V of a single-token step in ONE persistent launch per layer,
    results handed between the CUs as tagged 8-byte granules, the ffn_down activation quantised by the CUs that own its 256-blocks) runs the arithmetic
    of the per-launch kernels: logits must agree bit for bit, step after step - Llama-3-8B's layer geometry (the one the engine has a form for: an ffn_down
    whose K is a multiple of 1024), Q4_K / Q5_K / Q6_K tensors, two and three layers, from a graph and eagerly, across a 64-cell chunk boundary, and again after the cache was cleared (hand-over tags keep counting)."""
    _need_experiments(be, "decode_engine")
    path = make(pkg, tmp_models, cfg, ftype)
    m = pkg.Model(path)
    prompt = np.random.default_rng(5).integers(0, m.n_vocab, 40)

    def run(engine):
        be.set_option("decode_engine", 1 if engine else 0)
        be.set_option("attn_out_fused", 0)      # the engine's launch starts at attn_output: the per-launch arm must run the two-launch attention it follows
        try:
            c = pkg.Context(m, n_ctx=256, type_k=KV[kv], type_v=KV[kv], use_graphs=graphs)
            rows = []
            for rep in range(2):
                assert c.decode(prompt, np.arange(40)) == 0
                rows.append(c.logits().copy())
                for s in range(40):
                    assert c.decode([int(rows[-1].argmax())], [40 + s]) == 0
                    rows.append(c.logits().copy())
                c.kv_clear()
            assert c.engine_steps() == (80 if engine else 0)      # the path under test really ran
            c.close()
        finally:
            be.set_option("decode_engine", -1)
            be.set_option("attn_out_fused", -1)
        return np.stack(rows)

    a, b = run(True), run(False)
    assert np.isfinite(a).all()
    assert np.array_equal(a, b), int(np.argmax(np.abs(a - b).max(axis=1) > 0))
    m.close()


@pytest.mark.parametrize("cfg,ftype,kv", [("tiny-8b-2l", "q4_k_m", "q8_0"), ("tiny-8b-2l", "q5_k_m", "f16"), ("tiny-e2048", "q4_k_m", "q8_0"),
                                          ("tiny-d128", "q4_k_m", "q8_0"), ("tiny-g8", "q8_0", "q8_0"),
                                          # mixture of experts: the token's two selected experts share a launch (gate | up with SwiGLU, then down), expert index read on the device
                                          ("tiny-moe-e2048", "q4_k_m", "q8_0"), ("tiny-moe-e2048", "q5_k_m", "f16"),
                                          # contraction lengths that end inside a 2048-wide pass: TinyLlama's feed-forward width 5632, Qwen2-7B's hidden size 3584
                                          ("tiny-tl-2l", "q4_k_m", "f16"), ("tiny-qwen2-7b-2l", "q4_k_m", "q8_0"), ("tiny-qwen2-1.5b-2l", "q5_k_m", "q8_0"),
                                          # round 4: Q2_K / Q3_K tensors (the smoke model's mix at TinyLlama's geometry: 672- and 880-byte rows; Llama-3-8B's: row pairs
                                          # at K = 14336), the 32-element formats (Q8_0 activation blocks in the prologue) and launches of Q8_0 tensors only
                                          ("tiny-tl-2l", "q2_k", "f16"), ("tiny-tl-2l", "q3_k_m", "q8_0"), ("tiny-8b-2l", "q2_k", "q8_0"), ("tiny-8b-2l", "q3_k_s", "q8_0"),
                                          ("tiny-e2048", "q4_0", "q8_0"), ("tiny-8b-2l", "q5_0", "q8_0"), ("tiny-8b-2l", "iq4_nl", "f16"), ("tiny-8b-2l", "q8_0", "q8_0")])
def test_weight_stream_matvec_matches_register_ring_bitwise(be, pkg, tmp_models, cfg, ftype, kv):
    """The single-token mat-vecs run as an LDS-DMA weight stream (mmvq_stream.hip: loader waves + consumer waves per CU);
    the register-ring kernel (mmvq_fast.hip) stays as the form for shapes the stream has none for.  Same arithmetic, same
    lane roles, same summation order: logits must agree bit for bit, step after step (Llama-3-8B's layer geometry with
    Q4_K / Q6_K and Q5_K tensors, K = 2048 and 1024 models, a Q8_0 model where only the fused-prologue launches stream)."""
    path = make(pkg, tmp_models, cfg, ftype)
    m = pkg.Model(path)
    prompt = np.random.default_rng(5).integers(0, m.n_vocab, 24)

    def run(stream):
        be.set_option("mmvq_stream", 1 if stream else 0)
        try:
            c = pkg.Context(m, n_ctx=256, type_k=KV[kv], type_v=KV[kv])
            assert c.decode(prompt, np.arange(24)) == 0
            rows = [c.logits().copy()]
            for s in range(40):
                assert c.decode([int(rows[-1].argmax())], [24 + s]) == 0
                rows.append(c.logits().copy())
            c.close()
        finally:
            be.set_option("mmvq_stream", 1)
        return np.stack(rows)

    a, b = run(True), run(False)
    assert np.isfinite(a).all()
    assert np.array_equal(a, b), int(np.argmax(np.abs(a - b).max(axis=1) > 0))
    m.close()


@pytest.mark.parametrize("ftype", ["q4_k_m", "q5_k_m"])
def test_prompt_lds_kernel_matches_per_lane_kernels(be, pkg, tmp_models, ftype):
    """A 400-token prompt (longer than the K-split kernel takes) on Llama-3-8B's layer geometry (2 layers; the second one carries the Q6_K attn_v / ffn_down of the
    *_K_M mixes): with the 128 x 256 LDS kernel in play gate | up run on it, FFN down splits K, and Q | K | V of the mixed-type
    layer go out as ONE launch whose row tiles are Q4_K / Q5_K or Q6_K by segment.  Same integer sums as the per-lane planes
    kernels; only the f32 order over super-blocks differs where K is split."""
    path = make(pkg, tmp_models, "tiny-8b-2l", ftype)
    m = pkg.Model(path)
    prompt = np.random.default_rng(8).integers(0, m.n_vocab, 400)

    def run(lds, split):
        be.set_option("mmq_lds_form", 1 if lds else 0)
        be.set_option("mmq_split", split)
        try:
            c = pkg.Context(m, n_ctx=512, type_k=8, type_v=8)
            assert c.decode(prompt, np.arange(400)) == 0
            out = c.logits().copy()
            c.close()
        finally:
            be.set_option("mmq_lds_form", -1)
            be.set_option("mmq_split", 0)
        return out

    old, lds, lds_split = run(False, 0), run(True, 1), run(True, 0)
    assert np.isfinite(lds).all()
    assert np.array_equal(lds, old)                            # unsplit: same sums in the same order, mixed-type launch included
    # split K: the f32 order over super-blocks changes; this random-weight model amplifies such last-bit changes through the
    # re-quantisation of the activations (a different attention split count moves its logits by as much)
    assert not np.array_equal(lds_split, old)
    assert np.abs(lds_split - old).max() <= 3e-2 * max(1.0, float(np.abs(old).max()))
    m.close()


def test_prompt_path_is_deterministic_across_micro_batches(be, pkg, tmp_models):
    """A 700-token prompt in micro-batches of 512 (the second one attends to the first one's cells): the default path - K split
    with its fixed-order reduction, attention key splits merged in split order - gives the same bits twice, and the unsplit LDS
    form equals the per-lane kernels bit for bit there too.  (tools/stress_prefill.py runs the same check over random lengths.)"""
    path = make(pkg, tmp_models, "tiny-8b-2l", "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(10).integers(0, m.n_vocab, 700)

    def run(lds, split):
        be.set_option("mmq_lds_form", lds)
        be.set_option("mmq_split", split)
        try:
            c = pkg.Context(m, n_ctx=1024, n_batch=2048, n_ubatch=512, type_k=8, type_v=8)
            assert c.decode(prompt, np.arange(700)) == 0
            out = c.logits().copy()
            c.close()
        finally:
            be.set_option("mmq_lds_form", -1)
            be.set_option("mmq_split", 0)
        return out

    a, b = run(-1, 0), run(-1, 0)
    assert np.isfinite(a).all() and np.array_equal(a, b)
    assert np.array_equal(run(1, 1), run(0, 0))
    m.close()


@pytest.mark.parametrize("cfg", ["tiny-gqa4", "tiny-d128", "tiny-g8"])
def test_graph_and_eager_agree_bitwise(be, pkg, tmp_models, cfg):
    path = make(pkg, tmp_models, cfg, "q4_k_m")
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


def test_a_timed_out_wait_fails_the_step_in_flight_and_nothing_else(be, pkg, tmp_models):
    """The sticky error word of the cross-workgroup kernels (a bounded in-kernel wait that gives up raises it; here the test's hook does): the context whose
    step was in flight discards that step ONCE, says why, leaves the kernels that wait for other workgroups (the one-launch attention + attn_output, the layer
    engine) and carries on with the same results from the wait-free launches; a context of the same process that had nothing in flight is not touched."""
    path = make(pkg, tmp_models, "tiny-d128", "q4_k_m")
    m = pkg.Model(path)
    a = pkg.Context(m, n_ctx=64, type_k=8, type_v=8)
    b = pkg.Context(m, n_ctx=64, type_k=8, type_v=8)
    prompt = [5, 6, 7, 8, 9]
    for c in (a, b):
        assert c.decode(prompt, np.arange(5)) == 0
    ref0 = b.logits().copy()                                  # (b is checked and idle from here on)
    assert np.array_equal(a.logits(), ref0)
    t = int(ref0.argmax())
    assert b.decode([t], [5]) == 0
    ref1 = b.logits().copy()
    t2 = int(ref1.argmax())
    assert a.decode([t], [5]) == 0                           # a's step is in flight ...
    be.set_option("raise_stream_error", 1)                   # ... when some kernel's wait "gives up"
    with pytest.raises(pkg.binding.MI355Error):
        a.logits()
    # the bystander: nothing launched since its last check - it takes note and carries on
    assert b.decode([t2], [6]) == 0
    ref2 = b.logits().copy()
    # the context that answered: the step is repeated (its cell is overwritten), now on the two-launch attention path - the same arithmetic up to the order in
    # which a kv head's partial records are merged
    assert a.kv_seq_rm(0, 5, -1)
    assert a.decode([t], [5]) == 0
    got1 = a.logits().copy()
    assert rel_err(got1, ref1) <= 1e-5, rel_err(got1, ref1)
    assert a.decode([t2], [6]) == 0
    assert rel_err(a.logits(), ref2) <= 1e-5
    a.close(); b.close(); m.close()


def test_two_contexts_decoding_at_once_never_run_the_waiting_kernel_side_by_side(be, pkg, tmp_models):
    """Two models of one server stepping at the same time (two contexts, two host threads, their own streams): the one-launch attention + attn_output kernel is
    built of workgroups that wait for each other, and two copies placed on the same CUs at the same moment can hold each other's item workgroups out until the
    waits run into their bound (round 6 saw exactly that between processes, profiles/r6_tp_shared_device_trace.txt).  A context that finds the device's
    waiting kernel taken runs that step on the wait-free launches: no step fails, every row stays at the solo run's values up to the order in which a kv
    head's partial records are merged, and the counter shows that steps did meet."""
    import threading
    path = make(pkg, tmp_models, "tiny-8b-2l", "q4_k_m")
    m = pkg.Model(path)
    prompt = np.random.default_rng(3).integers(0, m.n_vocab, 2300)      # > 2048 cells: 128-cell chunks, one attention item per workgroup and more

    def run(c, n_steps, out, errs):
        try:
            for i0 in range(0, prompt.size, 512):
                assert c.decode(prompt[i0:i0 + 512], np.arange(i0, min(i0 + 512, prompt.size))) == 0
            rows = [c.logits().copy()]
            for s in range(n_steps):
                assert c.decode([int(rows[-1].argmax())], [prompt.size + s]) == 0
                rows.append(c.logits().copy())
            out.append(np.stack(rows))
        except Exception as e:      # noqa: BLE001 - reported by the main thread
            errs.append(repr(e))

    solo, errs = [], []
    c0 = pkg.Context(m, n_ctx=2560, type_k=8, type_v=8)
    run(c0, 48, solo, errs)
    assert not errs, errs
    assert c0.fused_skipped_steps() == 0
    c0.close()
    a, b = pkg.Context(m, n_ctx=2560, type_k=8, type_v=8), pkg.Context(m, n_ctx=2560, type_k=8, type_v=8)
    ra, rb = [], []
    ta, tb = threading.Thread(target=run, args=(a, 48, ra, errs)), threading.Thread(target=run, args=(b, 48, rb, errs))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for r in (ra[0], rb[0]):
        assert r.shape == solo[0].shape
        # teacher-forced on its own arg-max: rows agree while the paths pick the same tokens; compare the common prefix of identical choices
        same = 0
        while same < r.shape[0] and int(r[same].argmax()) == int(solo[0][same].argmax()):
            same += 1
        assert same >= 8, same
        assert max(rel_err(r[i], solo[0][i]) for i in range(same)) <= FLIP_TOL
    assert a.fused_skipped_steps() + b.fused_skipped_steps() > 0         # the two did step at the same time
    a.close(); b.close(); m.close()


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
        assert rel_err(g, ref[j]) <= FLIP_TOL
    # one decode step for both sequences in one batch
    t0, t1 = int(ref[0].argmax()), int(ref[1].argmax())
    assert c.decode([t0, t1], [37, 11], [0, 1], [1, 1]) == 0
    r2 = oc.decode([t0, t1], [37, 11], [0, 1], [1, 1])
    for j in range(2):
        assert rel_err(c.logits(j), r2[j]) <= FLIP_TOL
    c.close(); m.close(); oc.close(); om.close()


@pytest.mark.parametrize("cfg,kv", [("tiny-qwen2-7b-2l", "q8_0"), ("tiny-qwen2", "f16"), ("tiny-qwen2-1.5b-2l", "q8_0"),
                                    # (a llama-pairing file with a q4_0 cache: the same step kinds through the nibble cache - new rows quantised inside the attention launch)
                                    ("tiny-g8", "q4_0"), ("tiny-8b-2l", "q4_0")])
def test_neox_batched_steps_match_oracle(be, pkg, tmp_models, cfg, kv):
    """qwen2-type files through the continuous-batching step: three sequences of different lengths advance together (every token of another sequence: K is
    rotated with the NEOX pairing and stored inside the attention launch), then one sequence takes two tokens in one step (the general path); against the
    CPU restatement, sequence by sequence.  Also a context shift on one of them (K re-rotation with the NEOX pairing)."""
    path = make(pkg, tmp_models, cfg, "q4_k_m")
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    try:
        m = pkg.Model(path)
        c = pkg.Context(m, n_ctx=512, n_seq_max=4, type_k=KV[kv], type_v=KV[kv])
        om = oq.OracleModel(path)
        oc = oq.OracleContext(om, 512, KV[kv], KV[kv], True, oq.threads())
        rng = np.random.default_rng(17)
        lens = [70, 9, 33]
        for sq, n in enumerate(lens):
            p = rng.integers(0, m.n_vocab, n)
            fl = np.zeros(n, np.int8); fl[-1] = 1
            assert c.decode(p, np.arange(n), [sq] * n, fl) == 0
            r = oc.decode(p, np.arange(n), [sq] * n, fl)
            assert rel_err(c.logits(n - 1), r[0]) <= FLIP_TOL
        toks = [3, 5, 7]
        for step in range(8):
            pos = [n + step for n in lens]
            assert c.decode(toks, pos, [0, 1, 2], [1, 1, 1]) == 0
            r = oc.decode(toks, pos, [0, 1, 2], [1, 1, 1])
            for j in range(3):
                assert rel_err(c.logits(j), r[j]) <= FLIP_TOL, (step, j)
            toks = [int(x.argmax()) for x in r]
        for x in (c, oc):                       # context shift on sequence 0: drop [4, 20), slide the rest down
            x.kv_seq_rm(0, 4, 20)
            x.kv_seq_add(0, 20, lens[0] + 8, -16)
        p0 = lens[0] + 8 - 16
        assert c.decode([toks[0], 11], [p0, p0 + 1], [0, 0], [1, 1]) == 0
        r = oc.decode([toks[0], 11], [p0, p0 + 1], [0, 0], [1, 1])
        for j in range(2):
            assert rel_err(c.logits(j), r[j]) <= FLIP_TOL
        c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


@pytest.mark.parametrize("kv", ["f16", "q8_0"])
def test_kv_seq_ops_match_oracle(be, pkg, tmp_models, kv):
    """prompt-prefix reuse (seq_rm), seq_cp, and context shift (seq_rm + seq_add => K re-rotation), as
    LlamaServerContext::UpdateSlots drives them (reference llama_server_context.cc:1288-1291,1540-1547)."""
    path = make(pkg, tmp_models, "tiny-gqa4", "q4_k_m")
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    m, c, om, oc = open_pair(pkg, path, 96, kv)
    rng = np.random.default_rng(3)
    p = rng.integers(0, m.n_vocab, 40)
    for x in (c, oc):
        x.decode(p, np.arange(40))
    # 1. drop the tail and re-evaluate a different continuation (prompt cache)
    assert c.kv_seq_rm(0, 25, -1) and oc.kv_seq_rm(0, 25, -1)
    q = rng.integers(0, m.n_vocab, 9)
    c.decode(q, 25 + np.arange(9)); r = oc.decode(q, 25 + np.arange(9))[0]
    assert rel_err(c.logits(), r) <= FLIP_TOL
    assert c.kv_used() == 34
    # 2. context shift: discard positions [4, 14), slide the rest down by 10
    for x in (c, oc):
        x.kv_seq_rm(0, 4, 14)
        x.kv_seq_add(0, 14, 34, -10)
    c.decode([7], [24]); r = oc.decode([7], [24])[0]
    # K rows are re-rotated through a dequantise/requantise round trip on both sides
    assert rel_err(c.logits(), r) <= FLIP_TOL
    # 3. fork the sequence and continue the copy
    for x in (c, oc):
        x.kv_seq_cp(0, 1, 0, -1)
    c.decode([9], [25], [1]); r = oc.decode([9], [25], [1])[0]
    assert rel_err(c.logits(), r) <= FLIP_TOL
    # 4. cache full -> llama_decode returns 1 (caller halves n_batch), state unchanged
    used = c.kv_used()
    assert c.decode(rng.integers(0, m.n_vocab, 96), np.arange(96) + 100) == 1
    assert c.kv_used() == used
    c.kv_clear()
    assert c.kv_used() == 0
    c.close(); m.close(); oc.close(); om.close()
    oq.set_fa_v_acc_f32(0)


@pytest.mark.parametrize("cfg,kv", [("tiny-gqa4", "q8_0"), ("tiny-g8", "q8_0"), ("tiny-tl-2l", "f16")])
@pytest.mark.parametrize("scaling", ["linear", "yarn"])
def test_rope_scaling_from_the_file(be, pkg, tmp_path, cfg, kv, scaling):
    """{arch}.rope.scaling.{type, factor, original_context_length, attn_factor} (read by llama.cpp when the model loads - the reference passes no rope options
    of its own): linear and YaRN scaling through the prompt kernels, the single-token steps (cos / sin table of the step) and a context shift
    (K rows re-rotated by the position delta), against the CPU restatement; and the scaling is really in effect (the same weights without it differ)."""
    import dataclasses
    base = pkg.gguf_synth.CONFIGS[cfg]
    extra = {"rope.scaling.type": scaling, "rope.scaling.factor": 4.0}
    if scaling == "yarn":
        extra.update({"rope.scaling.original_context_length": 32, "rope.scaling.attn_factor": 0.9})
    path = str(tmp_path / f"{cfg}-{scaling}.gguf"); plain = str(tmp_path / f"{cfg}-plain.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, dataclasses.replace(base, extra=extra), "q4_k_m", seed=11)
    pkg.gguf_synth.write_synthetic_llama(plain, base, "q4_k_m", seed=11)
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    try:
        m, c, om, oc = open_pair(pkg, path, 160, kv)
        rng = np.random.default_rng(9)
        n_prompt = 70
        prompt = rng.integers(0, m.n_vocab, n_prompt)
        c.decode(prompt, np.arange(n_prompt)); ref = oc.decode(prompt, np.arange(n_prompt))[0]
        scaled = c.logits().copy()
        assert rel_err(scaled, ref) <= FLIP_TOL
        tok = int(ref.argmax())
        for step in range(12):
            c.decode([tok], [n_prompt + step]); r = oc.decode([tok], [n_prompt + step])[0]
            assert rel_err(c.logits(), r) <= FLIP_TOL, step
            tok = int(r.argmax())
        for x in (c, oc):                     # context shift: drop [8, 40), slide the rest down
            x.kv_seq_rm(0, 8, 40)
            x.kv_seq_add(0, 40, n_prompt + 12, -32)
        c.decode([tok], [n_prompt + 12 - 32]); r = oc.decode([tok], [n_prompt + 12 - 32])[0]
        assert rel_err(c.logits(), r) <= FLIP_TOL
        c.close(); m.close(); oc.close(); om.close()
        m2 = pkg.Model(plain); c2 = pkg.Context(m2, n_ctx=160, type_k=KV[kv], type_v=KV[kv])
        c2.decode(prompt, np.arange(n_prompt))
        assert rel_err(c2.logits(), scaled) > 10 * FLIP_TOL      # the scaling is not a no-op on these weights
        c2.close(); m2.close()
    finally:
        oq.set_fa_v_acc_f32(0)


def test_load_errors(be, pkg, tmp_path):
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(b"NOPE" + b"\0" * 64)
    with pytest.raises(pkg.MI355Error):
        pkg.Model(str(bad))
    with pytest.raises(pkg.MI355Error):
        pkg.Model(str(tmp_path / "missing.gguf"))


@pytest.mark.parametrize("arch", ["gemma2", "phi3", "bert"])
def test_other_graphs_are_refused_by_name(be, pkg, tmp_path, arch):
    """Only llama-graph files and the nomic-bert encoder run (general.architecture llama / qwen2 / nomic-bert): a file of another architecture is refused when it loads, with its name in the
    message - never evaluated as if it were a llama graph (wrong logits with status 200 is the failure this guards against)."""
    import dataclasses
    path = str(tmp_path / "other.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, dataclasses.replace(pkg.gguf_synth.CONFIGS["tiny"], arch=arch), "q8_0", seed=1)
    with pytest.raises(pkg.MI355Error, match=arch):
        pkg.Model(path)


@pytest.mark.parametrize("cfg,ftype,n", [("tiny-nomic", "f16", 19), ("tiny-nomic", "q8_0", 40), ("tiny-nomic", "q4_k_m", 70), ("nomic-embed-2l", "f16", 33)])
def test_encoder_hidden_states_match_oracle(be, pkg, tmp_models, cfg, ftype, n):
    """general.architecture nomic-bert (the reference's embedding smoke model, Makefile:6): the bidirectional encoder graph (run_layers_encoder: LayerNorms, fused
    Q | K | V, NEOX rope, attention over the whole sequence, SwiGLU) layer by layer against the CPU restatement, the embeddings rows (= the last layer's rows),
    two sequences in one batch that must not see each other, and the dependence of the first token on the last (not causal)."""
    path = make(pkg, tmp_models, cfg, ftype, seed=5)
    oq.set_fa_v_acc_f32(1)            # f16 cache: compare with f32 accumulation of V (as for the llama graph's f16 cases)
    try:
        m = pkg.Model(path)
        c = pkg.Context(m, n_ctx=256, n_seq_max=4, type_k=KV["f16"], type_v=KV["f16"])
        om = oq.OracleModel(path); oc = oq.OracleContext(om, 256, KV["f16"], KV["f16"], True, oq.threads())
        rng = np.random.default_rng(6)
        toks = rng.integers(5, m.n_vocab, n)
        c.enable_taps(True)
        flags = np.ones(n, np.int8)
        assert c.decode(toks, np.arange(n), [0] * n, flags) == 0
        oc.decode(toks, np.arange(n), [0] * n, flags)
        for il in range(m.n_layer):
            a, b = c.layer_out(il, n).reshape(n, -1), oc.layer_out(il, n).reshape(n, -1)
            # LayerNorm'd rows of unit scale.  f16 files: f16 products re-associated; quantised files: a 1-ulp difference can flip one int8 rounding of an
            # activation block (the llama graph's FLIP_TOL case) - everything else agrees to f32 round-off, which the median shows
            assert rel_err(a, b) <= (1e-2 if ftype == "f16" else FLIP_TOL), (il, rel_err(a, b))
            # (f16 files: the CPU rounds the activations to f16 for vec_dot_f16, the device contracts f32 activations with the f16 weights - the llama graph's f16 case)
            # - in the FIRST layer: a flipped code there reaches every token of the next layer through the bidirectional attention
            if il == 0:
                assert float(np.median(np.abs(a - b))) <= (1e-3 if ftype == "f16" else 1e-4), float(np.median(np.abs(a - b)))
        last = oc.layer_out(m.n_layer - 1, n).reshape(n, -1)
        emb = np.stack([c.embeddings(i).copy() for i in range(n)])
        assert rel_err(emb, last) <= (1e-2 if ftype == "f16" else FLIP_TOL)
        assert float(np.median(np.abs(emb - last))) <= 1e-2
        first = emb[0].copy()
        c.enable_taps(False)
        # a second sequence beside the first: same tokens but the last one -> its rows differ from sequence 0's, sequence 0's rows do not move
        t2 = toks.copy(); t2[-1] = (t2[-1] + 9) % (m.n_vocab - 5) + 5
        assert c.decode(t2, np.arange(n), [1] * n, flags) == 0
        e2 = np.stack([c.embeddings(i).copy() for i in range(n)])
        oc.decode(t2, np.arange(n), [1] * n, flags)
        assert rel_err(e2, oc.layer_out(m.n_layer - 1, n).reshape(n, -1)) <= (1e-2 if ftype == "f16" else FLIP_TOL)
        assert np.abs(e2[0] - first).max() > 1e-3            # bidirectional: token 0 saw the changed last token
        c.kv_clear()
        assert c.decode(toks, np.arange(n), [2] * n, flags) == 0
        assert np.abs(c.embeddings(0) - first).max() <= 1e-5 # ... and only tokens of its own sequence
        # one- and two-token sequences (the single-row launches), and a q8_0 cache under the same graph
        for kvn, nn in (("f16", 1), ("f16", 2), ("q8_0", n)):
            c3 = pkg.Context(m, n_ctx=256, type_k=KV[kvn], type_v=KV[kvn])
            oc3 = oq.OracleContext(om, 256, KV[kvn], KV[kvn], True, oq.threads())
            oq.set_fa_v_acc_f32(1 if kvn == "f16" else 0)
            assert c3.decode(toks[:nn], np.arange(nn), [0] * nn, np.ones(nn, np.int8)) == 0
            oc3.decode(toks[:nn], np.arange(nn), [0] * nn, np.ones(nn, np.int8))
            e3 = np.stack([c3.embeddings(i).copy() for i in range(nn)])
            assert rel_err(e3, oc3.layer_out(m.n_layer - 1, nn).reshape(nn, -1)) <= (1e-2 if ftype == "f16" and kvn == "f16" else FLIP_TOL), (kvn, nn)
            c3.close(); oc3.close()
        oq.set_fa_v_acc_f32(1)
        # a batch longer than the micro-batch cannot be cut (every token needs all the others): refused
        c2 = pkg.Context(m, n_ctx=256, n_ubatch=16, type_k=KV["f16"], type_v=KV["f16"])
        with pytest.raises(pkg.MI355Error, match="micro-batch"):
            c2.decode(toks[:17], np.arange(17))
        c2.close(); c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


def _patch_u32(path, key, value):
    """Overwrite a u32 metadata value of a GGUF in place (key\0-less string, then type u32 = 4, then the value)."""
    import struct
    blob = bytearray(open(path, "rb").read())
    k = key.encode()
    at = blob.find(struct.pack("<Q", len(k)) + k + struct.pack("<I", 4))
    assert at >= 0, key
    off = at + 8 + len(k) + 4
    blob[off:off + 4] = struct.pack("<I", value)
    open(path, "wb").write(bytes(blob))


@pytest.mark.parametrize("key,value,needle", [
    ("llama.attention.head_count_kv", 0, "head_count_kv"),        # would divide by zero in the group size
    ("llama.attention.head_count_kv", 3, "head_count_kv"),        # does not divide head_count
    ("llama.attention.head_count", 2, "shape"),                   # head_dim 128 is fine, but attn_q has twice the rows it implies
    ("llama.embedding_length", 512, "token_embd"),                # tensors are 256 wide
    ("llama.feed_forward_length", 768, "feed_forward_length"),
    ("llama.expert_used_count", 9, "expert"),                     # more experts used than the file has (tiny-moe: 8)
])
def test_load_rejects_tensors_that_contradict_the_metadata(be, pkg, tmp_path, key, value, needle):
    """buffers are sized from the hyper-parameters, the kernels write one value per tensor row: a file whose
    tensors disagree with its own metadata must fail at load (upstream create_tensor checks every shape), not write out
    of bounds at the first decode."""
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "tiny-moe" if "expert" in key else "tiny", "q4_k_m", seed=3)
    m = pkg.Model(path); m.close()                                 # the untouched file loads
    _patch_u32(path, key, value)
    with pytest.raises(pkg.MI355Error) as ei:
        pkg.Model(path)
    assert needle in str(ei.value), str(ei.value)


def test_kv_entry_points_refuse_out_of_range_sequence_ids(be, pkg, tmp_models):
    """sequence ids index a 64-bit mask (1 << seq); positions below zero would free the cell just written."""
    path = make(pkg, tmp_models, "tiny", "q4_k_m")
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=64, n_seq_max=2)
    lib = pkg.load_library()
    c.decode([1, 2, 3], [0, 1, 2])
    used = c.kv_used()
    lib.mi355_kv_cache_seq_cp(c.h, 0, 64, -1, -1)                  # UB before: 1ull << 64
    lib.mi355_kv_cache_seq_cp(c.h, -3, 1, -1, -1)
    lib.mi355_kv_cache_seq_add(c.h, 200, 0, -1, 1)
    assert lib.mi355_kv_cache_seq_rm(c.h, 99, -1, -1) == 0
    assert c.kv_used() == used
    with pytest.raises(pkg.MI355Error):
        c.decode([4], [-1])                                        # negative position
    with pytest.raises(pkg.MI355Error):
        c.decode([4], [3], seq=[64])                               # beyond the 64-bit cell mask
    c.decode([4], [3])
    c.close(); m.close()


def test_embeddings_match_oracle_hidden_state(be, pkg, tmp_models):
    """llama_set_embeddings + llama_get_embeddings_ith: the final-norm hidden state of the flagged row (pooling NONE on
    this architecture), against the oracle's last-layer residual stream normalised in numpy with the file's own
    output_norm weights."""
    cfg, ftype, seed = "tiny-d128", "q4_k_m", 11
    path = make(pkg, tmp_models, cfg, ftype, seed)
    m, c, om, oc = open_pair(pkg, path, 128, "q8_0")
    gs = pkg.gguf_synth
    names = [t[0] for t in gs.model_tensors(gs.CONFIGS[cfg], ftype)]
    idx = names.index("output_norm.weight")
    w = np.random.default_rng([seed, idx]).uniform(0.9, 1.1, size=m.n_embd).astype(np.float32)
    prompt = np.random.default_rng(9).integers(0, m.n_vocab, 13)
    c.set_embeddings(True)
    assert c.decode(prompt, np.arange(13)) == 0
    got = c.embeddings(-1)
    oc.decode(prompt, np.arange(13))
    x = oc.layer_out(m.n_layer - 1, 13).reshape(13, -1)[-1].astype(np.float32)
    want = oq.rms_norm(x, gs.CONFIGS[cfg].eps) * w
    assert np.abs(got - want).max() <= FLIP_TOL * max(1.0, float(np.abs(want).max()))
    with pytest.raises(pkg.MI355Error):
        c.logits()                                   # embeddings mode computes no logits
    c.set_embeddings(False)
    assert c.decode([3], [13]) == 0
    assert c.logits().shape == (m.n_vocab,)
    with pytest.raises(pkg.MI355Error):
        c.embeddings()
    c.close(); m.close(); oc.close(); om.close()


@pytest.mark.parametrize("ftype,kv,n_prompt", [("q4_k_m", "q8_0", 300), ("q5_k_m", "q8_0", 96), ("q8_0", "q8_0", 40)])
def test_moe_prompt_batches_grouped_by_expert(be, pkg, tmp_models, ftype, kv, n_prompt):
    """ggml_mul_mat_id on a prompt batch (build_moe_ffn): the (token, rank) pairs are grouped by expert and every expert
    runs ONE batched contraction over its tokens (300 tokens / 8 experts top-2: ~75 rows per expert -> the MFMA kernels;
    96 -> the K-split small-batch kernel; Q8_0 experts -> the tiled mat-vec).  Checked against the CPU oracle per layer
    and on the logits, and against the per-(token, expert) mat-vec loop the single-token steps use (same integer sums)."""
    path = make(pkg, tmp_models, "tiny-moe", ftype)
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, 512, n_prompt)
    om = oq.OracleModel(path)
    oc = oq.OracleContext(om, 512, KV[kv], KV[kv], True, oq.threads())
    ref = oc.decode(prompt, np.arange(n_prompt))[0]
    ref_layers = [oc.layer_out(il, n_prompt) for il in range(2)]
    outs = {}
    for mode, gmin in (("grouped", 8), ("loop", 1 << 20)):
        be.set_option("moe_group_min", gmin)
        m = pkg.Model(path)
        c = pkg.Context(m, n_ctx=512, type_k=KV[kv], type_v=KV[kv], n_ubatch=512)
        c.enable_taps(True)
        c.decode(prompt, np.arange(n_prompt))
        outs[mode] = (c.logits(), [c.layer_out(il, n_prompt) for il in range(m.n_layer)])
        c.close(); m.close()
    be.set_option("moe_group_min", 8)
    for mode, (lg, layers) in outs.items():
        # per token: besides the int8 rounding flips of the dense models, a token whose router probabilities are a near tie
        # may pick another expert pair than the CPU did (its output then moves by more than FLIP_TOL): a few of 300 tokens
        # may, the rest must hold the usual bound, and most must agree with the oracle to f32 round-off after layer 0
        for il, (a, b) in enumerate(zip(layers, ref_layers)):
            a2, b2 = a.reshape(n_prompt, -1), b.reshape(n_prompt, -1)
            tok_err = np.abs(a2 - b2).max(axis=1) / max(1.0, float(np.abs(b2).max()))
            assert float((tok_err > FLIP_TOL).mean()) <= 0.02, (mode, il, np.sort(tok_err)[-8:])
            if il == 0:
                assert float(np.median(tok_err)) <= TIGHT_TOL, (mode, tok_err)
        assert rel_err(lg, ref) <= FLIP_TOL, (mode, rel_err(lg, ref))
    g0, l0 = outs["grouped"][1][0].reshape(n_prompt, -1), outs["loop"][1][0].reshape(n_prompt, -1)
    tok = np.abs(g0 - l0).max(axis=1) / max(1.0, float(np.abs(l0).max()))
    assert float(np.median(tok)) <= TIGHT_TOL and rel_err(outs["grouped"][0], outs["loop"][0]) <= FLIP_TOL, tok
    oc.close(); om.close()


@pytest.mark.parametrize("cfg", ["tiny-gqa4", "tiny-e2048"])
def test_device_topk_front_end_matches_host_order(be, pkg, tmp_models, cfg):
    """Device-side head of the sampler chain (mi355_get_topk_ith): the k best (token, logit) pairs of a row after logit_bias and the
    repetition / frequency / presence penalties, in the host sampler's order (higher logit first, lower id on ties) - ids and float bits exactly those
    of the same f32 operations done on the host copy of the row; k = 1 .. 128, with and without adjustments, a -inf bias, single-token steps and a
    flagged prompt row."""
    path = make(pkg, tmp_models, cfg, "q4_k_m")
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=128, type_k=KV["q8_0"], type_v=KV["q8_0"])
    rng = np.random.default_rng(3)
    prompt = rng.integers(0, m.n_vocab, 20)
    assert c.decode(prompt, np.arange(20)) == 0
    pos = 20
    for trial in range(6):
        row = c.logits().copy()
        for k in (1, 7, 40, 128):
            n_adj = [0, 5, 64, 190][trial % 4]
            tok = rng.choice(m.n_vocab, n_adj, replace=False).astype(np.int32)
            bias = np.where(rng.random(n_adj) < 0.5, rng.standard_normal(n_adj) * 3, 0).astype(np.float32)
            if n_adj:
                bias[0] = -np.inf
            cnt = np.where(rng.random(n_adj) < 0.6, rng.integers(1, 5, n_adj), 0).astype(np.int32)
            rp, fr, pr = np.float32(1.1), np.float32(0.05), np.float32(0.3)
            want = row.copy()
            for t, b, n in zip(tok, bias, cnt):
                l = np.float32(want[t] + b)
                if n > 0:
                    l = np.float32(l * rp) if l <= 0 else np.float32(l / rp)
                    l = np.float32(l - np.float32(np.float32(np.float32(n) * fr) + pr))
                want[t] = l
            order = np.lexsort((np.arange(m.n_vocab), -want.astype(np.float64)))[:k]      # logit descending, id ascending on ties
            got_t, got_l = c.topk(k, adj_tok=tok, adj_bias=bias, adj_count=cnt, repeat=float(rp), freq=float(fr), present=float(pr))
            assert got_t.tolist() == order.tolist(), (trial, k)
            assert got_l.view(np.uint32).tolist() == want[order].view(np.uint32).tolist(), (trial, k)
        assert c.decode([int(row.argmax())], [pos]) == 0
        pos += 1
    c.close(); m.close()


# Layer geometries of llama-architecture checkpoints in the wild (hidden size, heads, kv heads, feed-forward width) x the file types this backend loads: every
# combination must LOAD, run a prompt and three captured single-token steps, and land on the CPU restatement's logits.  (The kernels a tensor takes depend on
# its shape and type together - a feed-forward width that is not a multiple of 1024 under a Q3_K ffn_down once refused its first decode step.)
SHAPES = {
    "d64-mha": (1024, 16, 16, 2816),          # head_dim 64, one query head per kv head
    "tinyllama": (2048, 32, 4, 5632),
    "llama-3.2-1b": (2048, 32, 8, 8192),
    "llama-3.2-3b": (3072, 24, 8, 8192),       # three query heads per kv head: the general attention kernel
    "yi-34b": (7168, 56, 8, 20480),            # seven
    "llama-2-7b": (4096, 32, 32, 11008),
    "llama-3-8b": (4096, 32, 8, 14336),
    "llama-2-13b": (5120, 40, 40, 13824),
    "llama-30b": (6656, 52, 52, 17920),
    "codellama-34b": (8192, 64, 8, 22016),
    "gqa-5": (5120, 40, 8, 13824),             # five query heads per kv head
    "llama-3-70b": (8192, 64, 8, 28672),
}


@pytest.mark.parametrize("ftype", ["q4_k_m", "q2_k", "q3_k_s", "q6_k", "q8_0", "q4_0", "iq4_nl"])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_real_layer_geometries_by_file_type(be, pkg, tmp_path, shape, ftype):
    E, H, G, FF = SHAPES[shape]
    cfg = pkg.gguf_synth.LlamaConfig(f"sweep-{shape}", E, 1, H, G, FF, 512, 10000.0, 1e-5, 512, big_model=(shape == "llama-3-70b"))
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=3)
    m, c, om, oc = open_pair(pkg, path, 64, "q8_0")
    prompt = np.random.default_rng(9).integers(0, m.n_vocab, 9)
    assert c.decode(prompt, np.arange(9)) == 0
    ref = oc.decode(prompt, np.arange(9))[0]
    errs = [rel_err(c.logits(), ref)]
    tok = int(ref.argmax())
    for s in range(3):
        assert c.decode([tok], [9 + s]) == 0
        r = oc.decode([tok], [9 + s])[0]
        errs.append(rel_err(c.logits(), r))
        assert int(c.logits().argmax()) == c.argmax()
        tok = int(r.argmax())
    assert max(errs) <= FLIP_TOL, (shape, ftype, errs)
    c.close(); m.close(); oc.close(); om.close()
    os.remove(path)


def test_device_argmax_follows_a_changing_number_of_flagged_rows(be, pkg, tmp_models):
    """Multi-slot serving: the number of flagged rows of a step changes (4 generating slots, then 2, then 1).  The single-launch device arg-max
    keeps one ticket word per row; those words must sit at an address that does not move with the row count (round-3 advisor finding: behind the
    part values they did, and a launch with fewer rows found a stale part value where its zero counter should be)."""
    path = make(pkg, tmp_models, "tiny-gqa4", "q4_k_m")
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=256, type_k=KV["q8_0"], type_v=KV["q8_0"], n_seq_max=4)
    rng = np.random.default_rng(21)
    lens = [9, 13, 6, 11]
    for s, n in enumerate(lens):
        c.decode(rng.integers(0, m.n_vocab, n), np.arange(n), seq=s)
    for step, live in enumerate([[0, 1, 2, 3], [0, 1, 2, 3], [1, 3], [3], [0, 2, 3], [2], [0, 1, 2, 3], [1]]):
        toks = rng.integers(0, m.n_vocab, len(live))
        pos = [lens[s] for s in live]
        assert c.decode(toks, pos, seq=live, logits=[1] * len(live)) == 0
        for i in range(len(live)):
            assert c.argmax(i) == int(c.logits(i).argmax()), (step, live, i)
        for s in live:
            lens[s] += 1
    c.close(); m.close()


@pytest.mark.parametrize("cfg,ftype,kv", [("tiny-8b-2l", "q4_k_m", "q8_0"), ("tiny-g8", "q4_k_m", "f16"), ("tiny-d128-mha", "q5_k_m", "q8_0"),
                                          ("tiny-d128", "q6_k", "q8_0"), ("tiny-d128", "q4_k_m", "f16")])
def test_attn_out_one_launch_agrees_with_the_two_launches(be, pkg, tmp_models, cfg, ftype, kv):
    """Single-token steps with the decode attention and the attn_output mat-vec in ONE launch (attn_out.hip, the default) against the same steps as two
    launches: the attention items run on 512 threads there (16 cell groups in the P.V pass instead of 8), so the comparison is f32 re-association, not
    bits - until a 1-ulp difference flips an int8 code, which moves a logit by up to ~1e-2 (FLIP_TOL) and, once it sits in a K / V row of the next layer,
    stays for the rest of the run.  So: every step within FLIP_TOL, and the FIRST step after each of five prompts (no earlier flip to inherit) agrees to
    round-off for most of them."""
    path = make(pkg, tmp_models, cfg, ftype)
    rows = {}
    for fused in (1, 0):
        be.set_option("attn_out_fused", fused)
        try:
            m = pkg.Model(path)
            c = pkg.Context(m, n_ctx=512, type_k=KV[kv], type_v=KV[kv])
            out = []
            for seed in range(5):
                rng = np.random.default_rng(100 + seed)
                n_p = 90 + 37 * seed
                c.kv_clear()
                c.decode(rng.integers(0, m.n_vocab, n_p), np.arange(n_p))
                for s, t in enumerate(rng.integers(0, m.n_vocab, 12)):
                    c.decode([int(t)], [n_p + s])
                    out.append(c.logits())
            rows[fused] = np.stack(out)
            c.close(); m.close()
        finally:
            be.set_option("attn_out_fused", -1)
    errs = np.array([rel_err(a, b) for a, b in zip(rows[1], rows[0])]).reshape(5, 12)
    assert float(errs.max()) <= FLIP_TOL, errs
    assert float(np.median(errs[:, 0])) <= 1e-5, errs[:, 0]


@pytest.mark.parametrize("cfg,ftype,kv", [("tiny-8b-2l", "q4_k_m", "q8_0"), ("tiny-8b-2l", "q5_k_m", "f16"), ("tiny-g8", "q4_k_m", "f16"), ("tiny-d128-mha", "q5_k_m", "q8_0"),
                                          ("tiny-d128", "q6_k", "q8_0"), ("tiny-e2048", "q4_k_m", "q8_0"), ("tiny-moe-e2048", "q4_k_m", "q8_0"),
                                          ("tiny-8b-2l", "q6_k", "q8_0")])
def test_qkv_inside_the_attention_launch_is_bitwise_the_separate_launch(be, pkg, tmp_models, cfg, ftype, kv):
    """Round 6: a single-token step's Q | K | V mat-vecs (RMSNorm -> Q8_K prologue included) run INSIDE the attention + attn_output launch (attn_out.hip QF: ten
    waves per workgroup, two of them DMA loaders; q / k / v travel between workgroups as tagged granules) - one launch per layer's attention block instead of two.
    The arithmetic is the weight stream's and the attention kernel's own, so the logits must equal the two-launch form's BIT FOR BIT: after prompts of five
    lengths (1 .. 32 attention items per kv head; the last run crosses 2048 cells, where the items become 128-cell chunks), twelve steps each.  tiny-moe-e2048: an
    8-expert file keeps attn_k / attn_v in Q8_0 - those workgroups quantise the layer input to Q8_0 blocks, the attn_q ones to Q8_K.  tiny-8b-2l q6_k: 80 KB of
    Q | K | V rows per workgroup, more slots than a loader wave may have in flight."""
    path = make(pkg, tmp_models, cfg, ftype)
    rows = {}
    for fused in (1, 0):
        be.set_option("qkv_attn_fused", fused)
        try:
            m = pkg.Model(path)
            c = pkg.Context(m, n_ctx=2304, type_k=KV[kv], type_v=KV[kv])
            out = []
            for seed, n_p in enumerate((3, 64, 127, 300, 2040)):
                rng = np.random.default_rng(300 + seed)
                c.kv_clear()
                c.decode(rng.integers(0, m.n_vocab, n_p), np.arange(n_p))
                for s, t in enumerate(rng.integers(0, m.n_vocab, 12)):
                    assert c.decode([int(t)], [n_p + s]) == 0
                    out.append(c.logits().copy())
            rows[fused] = np.stack(out)
            assert (c.qkv_attn_launches() > 0) == bool(fused), c.qkv_attn_launches()     # the form under test really ran (and really did not)
            c.close(); m.close()
        finally:
            be.set_option("qkv_attn_fused", -1)
    assert np.isfinite(rows[1]).all()
    assert np.array_equal(rows[1], rows[0]), float(np.abs(rows[1] - rows[0]).max())


@pytest.mark.parametrize("cfg,ftype,kv,n_prompt", [("tiny-8b-2l", "q4_k_m", "q8_0", 3968), ("tiny-8b-attn-2l", "q4_k_m", "q8_0", 3968), ("tiny-d128", "q4_k_m", "f16", 3968),
                                                   ("tiny-d128", "q4_k_m", "q4_0", 3000)])
def test_context_filled_to_4096_matches_oracle(be, pkg, tmp_models, cfg, ftype, kv, n_prompt):
    """BASELINE config 3's ctx_len = 4096 (src/llama_engine.cc:612) with the context filled: a 3968-token prompt in two micro-batches of 2048 (the second
    one's queries attend to 2048 earlier cells through the key splits of the prompt attention), then 8 single-token steps at positions 3968 .. 3975 whose
    attention scans ~4000 cells (128-cell items in the one-launch form; 63 chunks merged in two request rounds in the two-launch form) - logits of the
    prompt's last token and of every step against the CPU restatement.  Two layers of Llama-3-8B's geometry keep the CPU side affordable (round 5: the CPU
    restatement on every core the box has and built -O3 -mavx2 - same bits, tests/test_golden*.py - takes this case from 184 s to about 120); tiny-8b-attn-2l is
    the same attention geometry (4096 wide, 32 query heads over 8 kv heads of 128) with a 2048-wide feed-forward: a second weight set at a fifth of the CPU time."""
    path = make(pkg, tmp_models, cfg, ftype)
    oq.set_fa_v_acc_f32(1 if kv == "f16" else 0)
    try:
        m = pkg.Model(path)
        c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=KV[kv], type_v=KV[kv])
        om = oq.OracleModel(path)
        oc = oq.OracleContext(om, 4096, KV[kv], KV[kv], True, oq.threads())
        prompt = np.random.default_rng(41).integers(0, m.n_vocab, n_prompt)
        assert c.decode(prompt, np.arange(n_prompt)) == 0
        ref = oc.decode(prompt, np.arange(n_prompt))[0]
        errs = [rel_err(c.logits(), ref)]
        tok = int(ref.argmax())
        for s in range(8):
            assert c.decode([tok], [n_prompt + s]) == 0
            r = oc.decode([tok], [n_prompt + s])[0]
            g = c.logits()
            errs.append(rel_err(g, r))
            assert int(g.argmax()) == c.argmax()
            top2 = np.sort(r)[-2:]
            if top2[1] - top2[0] > 2 * FLIP_TOL * max(1.0, float(np.abs(r).max())):
                assert c.argmax() == int(r.argmax()), s
            tok = int(r.argmax())
        assert max(errs) <= FLIP_TOL, errs
        c.close(); m.close(); oc.close(); om.close()
    finally:
        oq.set_fa_v_acc_f32(0)


def test_moe_forced_routing_hook(be, pkg, tmp_models):
    """mi355_debug_force_moe_ids: the next decode call takes the experts it is handed instead of its router's selection.  With the CPU restatement's own
    selections handed over, prompt and steps agree with it within FLIP_TOL on every row (no token can be sent to another expert by a rounding flip on a near
    tie of the router); with a deliberately different selection the logits move - the hook is live; and it arms exactly one call."""
    path = make(pkg, tmp_models, "tiny-moe", "q4_k_m")
    m, c, om, oc = open_pair(pkg, path, 128, "q8_0")
    rng = np.random.default_rng(17)
    prompt = rng.integers(0, m.n_vocab, 21)
    oq.moe_record_start()
    ref = [oc.decode(prompt, np.arange(21))[0]]
    routes = [oq.moe_record_get().reshape(m.n_layer, 21, -1)]
    toks = []
    for s in range(6):
        toks.append(int(ref[-1].argmax()))
        oq.moe_record_start()
        ref.append(oc.decode([toks[-1]], [21 + s])[0])
        routes.append(oq.moe_record_get().reshape(m.n_layer, 1, -1))
    oq.moe_record_start(0)
    c.force_moe_ids(routes[0])
    assert c.decode(prompt, np.arange(21)) == 0
    errs = [rel_err(c.logits(), ref[0])]
    for s in range(6):
        c.force_moe_ids(routes[s + 1])
        assert c.decode([toks[s]], [21 + s]) == 0
        errs.append(rel_err(c.logits(), ref[s + 1]))
    assert max(errs) <= FLIP_TOL, errs
    # the hook is live: other experts, other logits; and the call after it routes freely again
    free = c.logits().copy()
    c.kv_seq_rm(0, 26, -1)
    wrong = (routes[6] + 3) % 8
    c.force_moe_ids(wrong)
    assert c.decode([toks[5]], [26]) == 0
    assert rel_err(c.logits(), free) > 1e-3
    c.kv_seq_rm(0, 26, -1)
    assert c.decode([toks[5]], [26]) == 0
    assert rel_err(c.logits(), ref[6]) <= FLIP_TOL
    c.close(); m.close(); oc.close(); om.close()
