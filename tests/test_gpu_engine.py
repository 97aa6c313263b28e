"""GPU end-to-end tests through the engine surface (mi355_engine_*): what the reference's e2e suite drives over HTTP
(.github/scripts/e2e-test-server-*.sh: loadmodel -> chat/completions stream + non-stream -> unloadmodel), here through
the C-ABI with a synthetic tiny GGUF.  The text a greedy request returns is checked against an independent greedy
loop over mi355_decode + mi355_get_argmax_ith on the same model."""
import json
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_vocab_model(pkg, tmp_path_factory):
    path = str(tmp_path_factory.mktemp("eng") / "tiny-d128.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "tiny-d128", "q4_k_m", with_vocab=True)
    return path


@pytest.fixture(scope="module")
def engine(pkg, tiny_vocab_model):
    e = pkg.Engine()
    st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=512, n_parallel=2, ngl=100, user_prompt="u:",
                            ai_prompt="a:", system_prompt="s:")
    assert st["status_code"] == 200 and not st["has_error"], (st, body)
    yield e
    e.close()


def _norm(s: str) -> str:
    """Runs of U+FFFD collapse to one: how many replacement characters an ill-formed byte run becomes is a JSON-writer detail."""
    import re
    return re.sub("\ufffd+", "\ufffd", s)


GREEDY = dict(temperature=0.0, repeat_penalty=1.0, frequency_penalty=0.0, presence_penalty=0.0)


def _greedy_reference(pkg, path, prompt: str, n_predict: int) -> str:
    """Independent greedy loop; mirrors the reference's budget rule (n_predict + 1 sampled tokens, see
    tests/host/host_tests.cc expected_text) and stops at EOS."""
    m = pkg.Model(path)
    c = pkg.Context(m, n_ctx=512, n_seq_max=1)
    toks = m.tokenize(prompt, add_special=True, parse_special=True)
    assert c.decode(toks, list(range(len(toks)))) == 0
    out, pos = b"", len(toks)
    eos = m.lib.mi355_token_eos(m.h)
    for _ in range(n_predict + 1):
        t = int(np.argmax(c.logits(-1)))
        if t == eos:
            break
        out += m.token_to_piece(t)
        assert c.decode([t], [pos]) == 0
        pos += 1
    c.close()
    m.close()
    return out.decode("utf-8", errors="replace")


def test_model_listing_and_status(engine):
    st, body = engine.get_models()
    assert st["status_code"] == 200
    assert [d["id"] for d in body["data"]] == ["tiny-d128"]
    d = body["data"][0]
    assert d["object"] == "model" and d["vram"] > 0 and d["model_size"] > 0 and d["engine"] == "cortex.llamacpp"
    st, body = engine.get_model_status(model="tiny-d128")
    assert st["status_code"] == 200
    st, body = engine.get_model_status(model="nope")
    assert st["status_code"] == 409
    assert engine.is_supported("HandleChatCompletion") and engine.is_supported("LoadModel")


def test_chat_completion_matches_greedy_decode(pkg, engine, tiny_vocab_model):
    msgs = [{"role": "system", "content": "be brief"}, {"role": "user", "content": "hello world"}]
    res = engine.chat_completion(model="tiny-d128", messages=msgs, max_tokens=12, **GREEDY)
    st, body = res[-1]
    assert st["status_code"] == 200 and not st["has_error"] and st["is_done"] and not st["is_stream"]
    assert body["object"] == "chat.completion" and body["model"] == "_"   # llama_engine.cc: create_full_return_json(.., "_", ..)
    content = body["choices"][0]["message"]["content"]
    want = _greedy_reference(pkg, tiny_vocab_model, "s:be briefu:hello worlda:", 12)
    if "u:" not in want and "<|im_end|>" not in want:
        assert _norm(content) in (_norm(want.lstrip(" ")), _norm(want)), (content, want)
    u = body["usage"]
    assert u["completion_tokens"] <= 12 and u["total_tokens"] == u["prompt_tokens"] + u["completion_tokens"]

    # streaming returns the same text, framed as SSE chunks terminated by [DONE]
    res = engine.chat_completion(model="tiny-d128", messages=msgs, max_tokens=12, stream=True, **GREEDY)
    text = ""
    for st, body in res:
        assert st["is_stream"]
        for ev in body["data"].split("\n\n"):
            if not ev:
                continue
            assert ev.startswith("data: ")
            if ev == "data: [DONE]":
                continue
            ch = json.loads(ev[6:])
            assert ch["object"] == "chat.completion.chunk"
            text += ch["choices"][0]["delta"].get("content") or ""
    assert res[-1][0]["is_done"] and "data: [DONE]" in res[-1][1]["data"]
    assert _norm(text) == _norm(content)


def test_parallel_requests_share_the_batch(engine):
    """n_parallel = 2: two concurrent greedy requests both finish, and each equals its own serial run."""
    prompts = ["abc def", "xyz uvw abc"]
    serial = [engine.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": p}], max_tokens=24, **GREEDY)[-1][1]
              ["choices"][0]["message"]["content"] for p in prompts]
    out = [None, None]

    def run(i):
        out[i] = engine.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": prompts[i]}], max_tokens=24,
                                        **GREEDY)[-1][1]["choices"][0]["message"]["content"]

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert out[0] is not None and out[1] is not None
    # batching two sequences changes f32 summation order nowhere (rows are independent), so the text is identical
    assert out == serial


def test_seeded_sampling_is_reproducible(engine):
    kw = dict(model="tiny-d128", messages=[{"role": "user", "content": "tell me"}], max_tokens=16, temperature=0.9, top_k=20,
              top_p=0.9, seed=1234)
    a = engine.chat_completion(**kw)[-1][1]["choices"][0]["message"]["content"]
    b = engine.chat_completion(**kw)[-1][1]["choices"][0]["message"]["content"]
    assert a == b


def test_errors_and_unload(pkg, tiny_vocab_model):
    e = pkg.Engine()
    st, body = e.load_model(llama_model_path="/nonexistent/model.gguf")
    assert st["status_code"] == 500 and st["has_error"]
    # ngl = 0 (BASELINE config 1: the reference's CPU configuration) and partial counts load: every layer is placed on the device regardless
    st, body = e.load_model(llama_model_path=tiny_vocab_model, ngl=0, ctx_len=256, model="cpu-config")
    assert st["status_code"] == 200 and not st["has_error"], (st, body)
    ref = e.chat_completion(model="cpu-config", messages=[{"role": "user", "content": "hi"}], max_tokens=8, **GREEDY)[-1][1]["choices"][0]["message"]["content"]
    assert e.unload_model(model="cpu-config")[0]["status_code"] == 200
    st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=256, cache_type="q8_0", ngl=1)
    assert st["status_code"] == 200
    st, body = e.load_model(llama_model_path=tiny_vocab_model)
    assert st["status_code"] == 409
    res = e.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": "hi"}], max_tokens=4, **GREEDY)
    assert res[-1][0]["status_code"] == 200
    st, body = e.embedding(model="tiny-d128", input=["hi", "hello world", [5, 300, 301]])
    assert st["status_code"] == 200 and body["object"] == "list" and len(body["data"]) == 3
    for i, d in enumerate(body["data"]):
        v = np.asarray(d["embedding"], np.float64)
        assert d["index"] == i and v.shape == (1024,) and abs(float((v * v).sum()) - 1.0) < 1e-4
    assert body["usage"]["prompt_tokens"] == body["usage"]["total_tokens"] > 6
    st, b64 = e.embedding(model="tiny-d128", input="hi", encoding_format="base64")
    import base64
    raw = np.frombuffer(base64.b64decode(b64["data"][0]["embedding"]), dtype="<f4")
    assert np.allclose(raw, np.asarray(body["data"][0]["embedding"], np.float32), atol=1e-6)
    st, body = e.unload_model(model="tiny-d128")
    assert st["status_code"] == 200
    res = e.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": "hi"}], max_tokens=4)
    assert res[-1][0]["status_code"] == 409
    e.close()


def test_context_shift_and_prompt_cache_on_device(pkg, tiny_vocab_model):
    """A slot context of 96 cells and 200 generated tokens: the loop must shift the context (kv_seq_rm + kv_seq_add with
    K re-rotation on the device) several times and keep producing tokens; then a second request sharing the prompt
    prefix reuses the cache (cache_prompt) and, being greedy, starts with the same text."""
    e = pkg.Engine()
    st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=96, n_parallel=1, ngl=100, user_prompt="u:", ai_prompt="a:",
                            caching_enabled=True)
    assert st["status_code"] == 200
    msgs = [{"role": "user", "content": "abc def ghi"}]
    r1 = e.chat_completion(model="tiny-d128", messages=msgs, max_tokens=200, ignore_eos=True, **GREEDY)[-1]
    assert r1[0]["status_code"] == 200
    u = r1[1]["usage"]
    assert u["completion_tokens"] == 200                       # went well past the 96-cell context
    r2 = e.chat_completion(model="tiny-d128", messages=msgs, max_tokens=12, ignore_eos=True, **GREEDY)[-1]
    assert r2[0]["status_code"] == 200
    a, b = _norm(r1[1]["choices"][0]["message"]["content"]), _norm(r2[1]["choices"][0]["message"]["content"])
    assert len(b) > 0 and a[:max(1, len(b) // 2)] == b[:max(1, len(b) // 2)]
    e.close()


def test_device_sampling_front_end_gives_the_host_chain_text(pkg, tiny_vocab_model):
    """The device-side head of the sampler chain (logit_bias -> penalties -> top_k on the device, k candidates to the host: mi355_get_topk_ith) against the
    whole chain over the whole row on the host (load option device_sampling=false): same seeded text, request by request, for chains the front end
    takes (top_k within its limit, penalties, logit_bias) and for chains it leaves to the host (top_k off, mirostat)."""
    reqs = [
        dict(max_tokens=24, temperature=0.9, top_k=20, top_p=0.9, seed=7),
        dict(max_tokens=24, temperature=1.3, top_k=40, top_p=0.95, min_p=0.02, seed=8, repeat_penalty=1.2, frequency_penalty=0.1, presence_penalty=0.2),
        dict(max_tokens=24, temperature=0.7, top_k=100, seed=9, logit_bias=[[5, 4.0], [9, -100.0], [17, False]], repeat_penalty=1.1),
        dict(max_tokens=16, temperature=0.0, repeat_penalty=1.3, top_k=40),          # greedy with penalties: not the plain arg-max path
        dict(max_tokens=16, temperature=0.8, top_k=0, seed=10),                      # top_k off: the whole row
        dict(max_tokens=16, temperature=0.8, mirostat=2, seed=11),
        dict(max_tokens=16, temperature=0.8, top_k=30, seed=12, n_probs=3),
    ]
    texts = {}
    for dev in (True, False):
        e = pkg.Engine()
        st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=512, n_parallel=1, ngl=100, user_prompt="u:", ai_prompt="a:",
                                system_prompt="s:", device_sampling=dev)
        assert st["status_code"] == 200, (st, body)
        out = []
        for r in reqs:
            res = e.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": "tell me a story"}], **r)[-1][1]
            out.append(res["choices"][0]["message"]["content"])
        texts[dev] = out
        e.close()
    assert texts[True] == texts[False]
    assert len(set(texts[True])) > 1


def test_grammar_and_response_format_constrain_the_text(pkg, tiny_vocab_model, tmp_path):
    """`grammar` (GBNF), `response_format` {json_object | json_schema} and the load option `grammar_file` (src/llama_engine.cc:573-585, 793-814): a
    random-weight model writes what the grammar leaves it - with the device sampling front end (k candidates, a refused draw re-drawn from the masked
    row) and with the whole chain on the host, seed for seed the same text."""
    texts = {}
    for dev in (True, False):
        e = pkg.Engine()
        st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=512, n_parallel=1, ngl=100, user_prompt="u:", ai_prompt="a:", system_prompt="s:",
                                device_sampling=dev)
        assert st["status_code"] == 200, (st, body)
        out = []

        def ask(**kw):
            res = e.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": "go"}], **kw)
            return res[-1][0], res[-1][1]
        st, b = ask(max_tokens=64, temperature=0.8, seed=3, grammar='root ::= ("yes" | "no") " " [a-z]{3,5} "."\n')
        text = b["choices"][0]["message"]["content"].replace("</s>", "")
        import re
        assert st["status_code"] == 200 and re.fullmatch(r"(yes|no) [a-z]{3,5}\.", text), text
        assert b["choices"][0]["finish_reason"] == "stop"
        out.append(text)
        st, b = ask(max_tokens=0 + 200, temperature=0.0, response_format={"type": "json_schema", "json_schema": {"schema": {
            "type": "object", "properties": {"n": {"type": "integer"}, "tag": {"enum": ["a", "b"]}, "ok": {"type": "boolean"}}, "required": ["n", "tag"]}}})
        text = b["choices"][0]["message"]["content"].replace("</s>", "")
        doc = json.loads(text)
        assert st["status_code"] == 200 and isinstance(doc["n"], int) and doc["tag"] in ("a", "b") and set(doc) <= {"n", "tag", "ok"}, text
        out.append(text)
        st, b = ask(max_tokens=48, temperature=0.9, seed=4, response_format={"type": "json_object"})
        text = b["choices"][0]["message"]["content"].replace("</s>", "")
        assert st["status_code"] == 200 and text.lstrip().startswith("{"), text
        out.append(text)
        st, b = ask(max_tokens=8, grammar="root ::= nothing\n")
        assert st["status_code"] == 400 and "undefined rule nothing" in b["message"]
        texts[dev] = out
        e.close()
    assert texts[True] == texts[False]
    # grammar_file: the model's grammar for every completion
    gf = tmp_path / "only.gbnf"
    gf.write_text('root ::= "<" [0-9]+ ">"\n')
    e = pkg.Engine()
    st, body = e.load_model(llama_model_path=tiny_vocab_model, ctx_len=256, grammar_file=str(gf))
    assert st["status_code"] == 200
    b = e.chat_completion(model="tiny-d128", messages=[{"role": "user", "content": "go"}], max_tokens=40, temperature=0.7, seed=9)[-1][1]
    import re
    assert re.match(r"<[0-9]+>?", b["choices"][0]["message"]["content"]), b
    st, body = e.load_model(llama_model_path=tiny_vocab_model, model="other", grammar_file=str(tmp_path / "missing.gbnf"))
    assert st["status_code"] == 500
    e.close()


def test_pooled_embeddings_follow_the_model_metadata(pkg, tmp_path):
    """{arch}.pooling_type (llama_get_embeddings_seq, src/llama_server_context.cc:1041-1044): a model that asks for MEAN pooling answers an embedding request
    with the L2-normalised mean of its tokens' final hidden states, CLS with the first token's.  Checked against the same weights WITHOUT the key (pooling
    none = last token): the hidden state of token i of a causal model is the last-token state of the prompt's first i + 1 tokens."""
    gs = pkg.gguf_synth
    base = gs.CONFIGS["tiny-d128"]
    import dataclasses
    paths = {}
    for name, extra in (("none", {}), ("mean", {"pooling_type": 1}), ("cls", {"pooling_type": 2})):
        cfg = dataclasses.replace(base, extra=extra)
        paths[name] = str(tmp_path / f"pool-{name}.gguf")
        gs.write_synthetic_llama(paths[name], cfg, "q4_k_m", with_vocab=True)
    toks = [1, 300, 301, 302, 303, 304, 305]

    def embed(path, tokens):
        e = pkg.Engine()
        st, body = e.load_model(llama_model_path=path, ctx_len=256, model="m")
        assert st["status_code"] == 200, (st, body)
        out = []
        for t in tokens:
            st, b = e.embedding(model="m", input=[t])
            assert st["status_code"] == 200, (st, b)
            out.append(np.asarray(b["data"][0]["embedding"], np.float64))
        e.close()
        return out

    # per-token hidden states (normalised) from the pooling-free file: the engine normalises, so un-normalised states are not available - compare directions:
    # CLS must equal the first prefix's embedding; MEAN is checked through the raw C-ABI embeddings below
    prefixes = embed(paths["none"], [toks[:i + 1] for i in range(len(toks))])
    cls = embed(paths["cls"], [toks])[0]
    assert np.allclose(cls, prefixes[0], atol=1e-6)
    last = embed(paths["none"], [toks])[0]
    assert np.allclose(last, prefixes[-1], atol=2e-3)            # (a batch of 7 and a batch of 1 take different kernels: the rounding-flip level)
    # MEAN against the un-normalised hidden states of every token, read through the C-ABI (all rows flagged, embeddings on)
    m = pkg.Model(paths["none"])
    c = pkg.Context(m, n_ctx=256)
    c.set_embeddings(True)
    assert c.decode(toks, list(range(len(toks))), logits=np.ones(len(toks), np.int8)) == 0
    rows = np.stack([c.embeddings(i) for i in range(len(toks))]).astype(np.float32)
    c.close(); m.close()
    want = rows.sum(axis=0, dtype=np.float32) * np.float32(1.0 / len(toks))
    want = want.astype(np.float64) / np.linalg.norm(want.astype(np.float64))
    mean = embed(paths["mean"], [toks])[0]
    assert np.allclose(mean, want, atol=1e-6), float(np.abs(mean - want).max())


def test_reference_smoke_flow_embedding_model_is_an_encoder(pkg, tmp_path):
    """The second half of the reference's smoke script (.github/scripts/e2e-test-server-linux-and-mac.sh:93-120): load an embedding model - upstream a nomic-bert
    file (Makefile:6) - with {"ctx_len": 50, "ngl": 32, "embedding": true, "model_type": "embedding"}, list the models, POST /v1/embeddings {"input": "Hello"}.
    Here with the encoder's real layer geometry (768 wide, 12 heads of 64, 3072, 30522 WordPiece entries, two layers, f16): the reply must be the L2-normalised MEAN of
    the hidden states of [CLS] hello [SEP] as the CPU restatement computes them, a chat completion on the encoder must be refused, and a batch of inputs works."""
    import oracle_py as oq
    path = str(tmp_path / "test-embedding.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "nomic-embed-2l", "f16", seed=3, with_vocab=True)
    e = pkg.Engine()
    st, body = e.load_model(llama_model_path=path, ctx_len=50, ngl=32, embedding=True, model_type="embedding", model="test-embedding")
    assert st["status_code"] == 200, (st, body)
    st, body = e.get_models()
    assert st["status_code"] == 200 and any(m["id"] == "test-embedding" for m in body["data"]), body
    st, b = e.embedding(model="test-embedding", input="Hello", encoding_format="float")
    assert st["status_code"] == 200, (st, b)
    got = np.asarray(b["data"][0]["embedding"], np.float64)
    assert got.shape == (768,) and abs(np.linalg.norm(got) - 1.0) < 1e-5
    # the tokens the request must have produced: [CLS] + greedy longest-match WordPiece of "hello" + [SEP]
    be = pkg.Backend()
    m = pkg.Model(path)
    toks = m.tokenize("Hello", add_special=True)
    assert toks[0] == 2 and toks[-1] == 3 and len(toks) >= 3, toks
    assert m.tokenize("HELLO", add_special=True) == toks                     # lower-cased
    m.close()
    om = oq.OracleModel(path); oc = oq.OracleContext(om, 64, oq.F16, oq.F16, True, oq.threads())
    oq.set_fa_v_acc_f32(1)
    try:
        oc.decode(toks, np.arange(len(toks)), [0] * len(toks), np.ones(len(toks), np.int8))
    finally:
        oq.set_fa_v_acc_f32(0)
    rows = oc.layer_out(1, len(toks)).reshape(len(toks), -1).astype(np.float64)
    want = rows.mean(axis=0); want /= np.linalg.norm(want)
    oc.close(); om.close()
    assert np.abs(got - want).max() <= 2e-3, float(np.abs(got - want).max())
    assert float(got @ want) >= 0.9999
    # several inputs in one request; a different text gives a different vector
    st, b = e.embedding(model="test-embedding", input=["Hello", "hello, world!", "Hello"])
    assert st["status_code"] == 200 and len(b["data"]) == 3, (st, b)
    v = [np.asarray(d["embedding"]) for d in b["data"]]
    assert np.allclose(v[0], got, atol=1e-6) and np.allclose(v[2], got, atol=1e-6) and np.abs(v[1] - got).max() > 1e-3
    # an encoder has no next-token head
    out = e.chat_completion(model="test-embedding", messages=[{"role": "user", "content": "hi"}], max_tokens=4)
    assert out[-1][0]["has_error"] or out[-1][0]["status_code"] != 200, out
    e.close()
