"""Committed golden vectors of the ops and the whole decode path (tests/golden/ops_v1.npz, e2e_v1.npz, api_shapes_v1.json; made by
tests/golden/make_golden_ops.py — provenance in its docstring).  CPU tests: the generator reproduces the files, the C oracle reproduces the vectors (ops: at
the stated tolerance against independent float64 / f32-recurrence restatements; e2e: bit for bit, it wrote them), the engine façade's JSON matches the
shapes transcribed from the reference source.  GPU tests (-m gpu): the HIP path against the same files."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as oq

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GD = os.path.join(HERE, "golden")
OPS = np.load(os.path.join(GD, "ops_v1.npz"))
E2E = {**np.load(os.path.join(GD, "e2e_v1.npz")), **np.load(os.path.join(GD, "e2e_v2.npz"))}      # (v2: the cases round 4 added; v1 stays frozen)
KVT = {"q8_0": oq.Q8_0, "f16": oq.F16, "q4_0": oq.Q4_0}
API = json.load(open(os.path.join(GD, "api_shapes_v1.json"), encoding="utf-8"))
sys.path.insert(0, GD)
import make_golden_ops as gen  # noqa: E402  (E2E_CASES, N_PROMPT, N_STEPS)

ALL_E2E = gen.E2E_CASES + gen.E2E_CASES_V2

FLIP_TOL = 3e-2          # of the logit scale: one int8 rounding flip of an activation / cache code (tests/test_oracle_sensitivity.py)
ID_GAP = 6e-2            # a greedy id is compared where the recorded top-2 gap exceeds this fraction of the logit scale


def test_generator_reproduces_the_ops_and_api_files(tmp_path):
    d = tmp_path / "golden"
    d.mkdir()
    import shutil
    shutil.copy(os.path.join(GD, "make_golden_ops.py"), d / "make_golden_ops.py")
    shutil.copy(os.path.join(HERE, "np_twin.py"), tmp_path / "np_twin.py")
    subprocess.check_call([sys.executable, str(d / "make_golden_ops.py"), "ops", "api"], stdout=subprocess.DEVNULL)
    H = np.load(d / "ops_v1.npz")
    assert sorted(H.files) == sorted(OPS.files)
    for k in OPS.files:
        assert np.array_equal(OPS[k], H[k]), k
    assert json.load(open(d / "api_shapes_v1.json", encoding="utf-8")) == API


# ------------------------------------------------------------------------------------------------ ops: the oracle against the fixtures
def test_oracle_rope_matches_golden():
    x = OPS["rope_x"]
    H, D = x.shape
    for bi, base in enumerate(OPS["rope_base"]):
        for pi, pos in enumerate(OPS["rope_pos"]):
            y = oq.rope(x, H, D, int(pos), float(base))
            assert np.abs(y - OPS["rope_y"][bi, pi]).max() <= 1e-6, (base, pos)      # same f32 recurrence, same libm: a few ulp at most
    assert np.array_equal(OPS["rope_y"][0, 0], x)                                    # position 0 is the identity


def test_oracle_softmax_swiglu_route_match_golden():
    for r in range(OPS["softmax_x"].shape[0]):
        y = oq.soft_max(OPS["softmax_x"][r], OPS["softmax_mask"][r], float(OPS["softmax_scale"][0]))
        assert np.abs(y - OPS["softmax_y"][r]).max() <= 1e-7
    y = oq.silu(OPS["swiglu_g"]) * OPS["swiglu_u"]
    assert np.abs(y - OPS["swiglu_y"]).max() <= 1e-6 * max(1.0, float(np.abs(OPS["swiglu_y"]).max()))
    for t in range(OPS["route_logits"].shape[0]):
        ids, w = oq.moe_route(OPS["route_logits"][t], 2)
        assert ids.tolist() == OPS["route_ids"][t].tolist(), t
        assert np.abs(w - OPS["route_w"][t]).max() <= 1e-6


@pytest.mark.parametrize("name,tol", [("q8_0", 2e-5), ("f16", 3e-3)])
def test_oracle_attention_matches_golden(name, tol):
    """f16: the CPU path accumulates V in fp16 - it is the looser side of this comparison by design; with that accumulation in f32
    (oq.set_fa_v_acc_f32) it meets the tight bound too."""
    H, G, D = [int(v) for v in OPS["attn_shape_H_G_D"]]
    t = oq.Q8_0 if name == "q8_0" else oq.F16
    kc, vc = OPS[f"attn_k_{name}"], OPS[f"attn_v_{name}"]
    cell_pos = OPS["attn_cell_pos"]
    for mode, bound in ((0, tol), (1, 2e-5)):
        oq.set_fa_v_acc_f32(mode)
        try:
            for i, qp in enumerate(OPS["attn_q_pos"]):
                cells = np.nonzero((cell_pos >= 0) & (cell_pos <= qp))[0].astype(np.int32)
                y = oq.flash_attn(OPS["attn_q"][i], H, G, D, t, kc, t, vc, cells, 1.0 / np.sqrt(D))
                ref = OPS[f"attn_y_{name}"][i]
                assert np.abs(y.reshape(H, D) - ref).max() <= bound * max(1.0, float(np.abs(ref).max())), (mode, i)
        finally:
            oq.set_fa_v_acc_f32(0)


# ------------------------------------------------------------------------------------------------ e2e: the oracle wrote it, it must still write it
@pytest.mark.parametrize("cfg,ftype,kv,seed", ALL_E2E)
def test_oracle_reproduces_e2e_golden_bit_for_bit(pkg, tmp_path, cfg, ftype, kv, seed):
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    key = f"{cfg}.{ftype}.{kv}"
    om = oq.OracleModel(path)
    tkv = KVT[kv]
    oc = oq.OracleContext(om, 64, tkv, tkv, True, 2)
    prompt = E2E[f"{key}.prompt"]
    assert np.array_equal(prompt, np.random.default_rng(seed).integers(0, om.n_vocab if hasattr(om, "n_vocab") else pkg.gguf_synth.CONFIGS[cfg].n_vocab, gen.N_PROMPT))
    row = oc.decode(prompt, np.arange(gen.N_PROMPT))[0]
    for s in range(gen.N_STEPS):
        assert np.array_equal(row.astype(np.float32), E2E[f"{key}.logits"][s]), s
        tok = int(row.argmax())
        assert tok == int(E2E[f"{key}.ids"][s]), s
        row = oc.decode([tok], [gen.N_PROMPT + s])[0]
    oc.close(); om.close()


def test_oracle_reproduces_encoder_golden_bit_for_bit(pkg, tmp_path):
    """The encoder graph of the reference's embedding smoke model (nomic-bert, /root/reference Makefile:6): the last layer's hidden states of one sequence."""
    cfg, ftype, seed, n = gen.ENC_CASE
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    om = oq.OracleModel(path)
    oq.set_fa_v_acc_f32(1)
    try:
        oc = oq.OracleContext(om, 64, oq.F16, oq.F16, True, 2)
        toks = E2E[f"{cfg}.{ftype}.enc.tokens"]
        oc.decode(toks, np.arange(n), [0] * n, np.ones(n, np.int8))
        got = oc.layer_out(pkg.gguf_synth.CONFIGS[cfg].n_layer - 1, n).reshape(n, -1)
        assert np.array_equal(got.astype(np.float32), E2E[f"{cfg}.{ftype}.enc.hidden"])
        oc.close()
    finally:
        oq.set_fa_v_acc_f32(0)
    om.close()


# ------------------------------------------------------------------------------------------------ API shapes
def _match(pattern, got, path=""):
    if isinstance(pattern, dict):
        assert isinstance(got, dict), path
        assert sorted(pattern) == sorted(got), (path, sorted(pattern), sorted(got))
        for k in pattern:
            _match(pattern[k], got[k], path + "/" + k)
    elif isinstance(pattern, list):
        assert isinstance(got, list), path
        if not pattern:
            assert got == [], path
        for i, g in enumerate(got):
            _match(pattern[0], g, f"{path}[{i}]")
    elif pattern == "str":
        assert isinstance(got, str), (path, got)
    elif pattern == "int":
        assert isinstance(got, int) and not isinstance(got, bool), (path, got)
    else:
        assert got == pattern, (path, got, pattern)


def test_engine_json_matches_the_reference_shapes(tmp_path):
    from test_host_logic import SRCS
    exe = str(tmp_path / "host_tests")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-pthread", *SRCS, "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "--api-shapes"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    for name in ("load_model_ok", "load_model_no_id", "load_model_again", "unload_model_ok", "model_not_loaded", "get_model_status_ok"):
        _match(API[name]["status"], got[name]["status"], name + "/status")
        _match(API[name]["body"], got[name]["body"], name + "/body")
    # (a failed load carries the backend's reason next to the reference's message: an addition, the reference logs it instead)
    _match(API["load_model_failed"]["status"], got["load_model_failed"]["status"], "load_model_failed/status")
    assert got["load_model_failed"]["body"]["message"] == API["load_model_failed"]["body"]["message"]
    _match(API["get_models_body"], got["get_models"]["body"], "get_models")
    _match(API["chat_completion_status"], got["chat_completion"]["status"], "chat_completion/status")
    _match(API["chat_completion_body"], got["chat_completion"]["body"], "chat_completion/body")
    fr = API["stream_frame"]
    for usage, key in ((False, "stream"), (True, "stream_with_usage")):
        frames = got[key]
        assert len(frames) >= 2
        for f in frames[:-1]:
            _match(API["stream_status"]["running"], f["status"], key + "/status")
            data = f["body"]["data"]
            assert data.startswith(fr["data_prefix"]) and data.endswith(fr["data_suffix"])
            chunk = json.loads(data[len(fr["data_prefix"]):-len(fr["data_suffix"])])
            pat = dict(API["chat_chunk_body"])
            if usage:
                pat["usage"] = None                                   # "All other chunks will also include a usage field, but with a null value"
            _match(pat, chunk, key + "/chunk")
        last = frames[-1]
        _match(API["stream_status"]["final"], last["status"], key + "/final status")
        data = last["body"]["data"]
        assert data.endswith(fr["done"])
        chunk = json.loads(data[len(fr["data_prefix"]):-len(fr["done"]) - len(fr["data_suffix"])])
        _match(API["chat_chunk_usage_body"] if usage else API["chat_chunk_last_body"], chunk, key + "/last chunk")


# ------------------------------------------------------------------------------------------------ the HIP path against the same files
@pytest.mark.gpu
def test_hip_ops_match_golden(pkg):
    be = pkg.Backend()
    x = OPS["rope_x"]
    H, D = x.shape
    for bi, base in enumerate(OPS["rope_base"]):
        pos = OPS["rope_pos"]
        y = be.rope(np.broadcast_to(x, (pos.size, H, D)).copy(), H, D, pos, float(base))
        for pi in range(pos.size):
            assert np.abs(y[pi] - OPS["rope_y"][bi, pi]).max() <= 4e-6, (base, pos[pi])      # device cosf / sinf vs libm: a few ulp
    y = be.soft_max(OPS["softmax_x"], OPS["softmax_mask"], float(OPS["softmax_scale"][0]))
    assert np.abs(y - OPS["softmax_y"]).max() <= 1e-7
    y = be.swiglu(OPS["swiglu_g"], OPS["swiglu_u"])
    assert np.abs(y - OPS["swiglu_y"]).max() <= 1e-6 * max(1.0, float(np.abs(OPS["swiglu_y"]).max()))
    ids, w = be.moe_route(OPS["route_logits"], 2)
    assert np.array_equal(ids, OPS["route_ids"])
    assert np.abs(w - OPS["route_w"]).max() <= 1e-6
    Hh, G, Dd = [int(v) for v in OPS["attn_shape_H_G_D"]]
    for name, t in (("q8_0", oq.Q8_0), ("f16", oq.F16)):
        out = be.flash_attn(OPS["attn_q"], Hh, G, Dd, t, OPS[f"attn_k_{name}"], t, OPS[f"attn_v_{name}"], OPS["attn_cell_pos"], OPS["attn_q_pos"], 1.0 / np.sqrt(Dd))
        ref = OPS[f"attn_y_{name}"]
        # both cache types: the HIP kernels accumulate in f32, so they meet the float64 answer tightly (the CPU's fp16 V accumulation does not)
        assert np.abs(out.reshape(ref.shape) - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), name


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,ftype,kv,seed", ALL_E2E)
def test_hip_e2e_matches_golden(pkg, tmp_path, cfg, ftype, kv, seed):
    """12-token prompt + 32 teacher-forced greedy steps against the committed oracle run: every logits row within FLIP_TOL of the logit scale, the best rows
    within 2e-5 (f32 re-association only: before the first int8 rounding flip HIP == CPU), and the greedy id equal wherever the recorded gap between the two
    largest logits exceeds ID_GAP of the scale (below that a flip may legitimately change the winner)."""
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    key = f"{cfg}.{ftype}.{kv}"
    pkg.Backend()
    m = pkg.Model(path)
    tkv = {"q8_0": pkg.binding.Q8_0, "f16": pkg.binding.F16, "q4_0": 2}[kv]
    c = pkg.Context(m, n_ctx=64, type_k=tkv, type_v=tkv)
    prompt = E2E[f"{key}.prompt"]
    assert c.decode(prompt, np.arange(gen.N_PROMPT)) == 0
    errs, checked = [], 0
    for s in range(gen.N_STEPS):
        got, ref = c.logits(), E2E[f"{key}.logits"][s]
        scale = float(np.abs(ref).max())
        err = float(np.abs(got - ref).max()) / scale
        errs.append(err)
        assert err <= (5e-2 if kv == "f16" else FLIP_TOL), (s, err)
        if float(E2E[f"{key}.top2_gap"][s]) > ID_GAP * scale:
            assert int(got.argmax()) == int(E2E[f"{key}.ids"][s]), s
            checked += 1
        assert c.decode([int(E2E[f"{key}.ids"][s])], [gen.N_PROMPT + s]) == 0          # teacher-forced: the recorded id
    assert checked >= gen.N_STEPS // 2
    if kv != "f16":                                            # (f16 cache: the CPU's fp16 V accumulation is in the recorded rows)
        assert min(errs) <= 2e-5, min(errs)
    c.close(); m.close()


@pytest.mark.gpu
def test_hip_encoder_matches_golden(pkg, tmp_path):
    """nomic-bert encoder (the reference's embedding smoke model): the embeddings rows of one 24-token sequence against the committed hidden states of the
    CPU restatement's last layer (f16 weights: the CPU rounds the activations to f16 for vec_dot_f16, the device contracts f32 activations - 1e-2 on unit-scale
    rows, the typical element far closer)."""
    cfg, ftype, seed, n = gen.ENC_CASE
    path = str(tmp_path / "m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, cfg, ftype, seed=seed)
    pkg.Backend()
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=64, n_seq_max=1, type_k=pkg.binding.F16, type_v=pkg.binding.F16)
    toks = E2E[f"{cfg}.{ftype}.enc.tokens"]
    assert c.decode(toks, np.arange(n), [0] * n, np.ones(n, np.int8)) == 0
    emb = np.stack([c.embeddings(i).copy() for i in range(n)])
    ref = E2E[f"{cfg}.{ftype}.enc.hidden"]
    assert float(np.abs(emb - ref).max()) / max(1.0, float(np.abs(ref).max())) <= 1e-2
    assert float(np.median(np.abs(emb - ref))) <= 1e-3
    c.close(); m.close()
