"""The HTTP host (cortex.llamacpp_amd/server/mi355_server.cc) over a real socket: the routes of the reference's example server
(examples/server/server.cc:262-273) driven the way its e2e script drives them (.github/scripts/e2e-test-server-linux-and-mac.sh:
POST /loadmodel -> POST /v1/chat/completions non-stream + stream -> POST /v1/embeddings -> POST /unloadmodel), plus transport behaviour
(keep-alive, Origin echo, 404 / 405 / malformed requests, DELETE /destroy).  The routing and error paths need no GPU; the model flow does."""
import http.client
import json
import os
import re
import socket
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERVER = os.path.join(ROOT, "cortex.llamacpp_amd", "bin", "mi355_server")


class Host:
    def __init__(self, lib=None, env=None):
        assert os.path.exists(SERVER), "build it: python cortex.llamacpp_amd/build.py"
        self.p = subprocess.Popen([SERVER, "127.0.0.1", "0"] + (["--lib", lib] if lib else []), stderr=subprocess.PIPE, text=True,
                                  env=dict(os.environ, **(env or {})))
        line = ""
        t0 = time.time()
        while "listening" not in line:
            line = self.p.stderr.readline()
            assert line or self.p.poll() is None, "server exited"
            assert time.time() - t0 < 120
        self.port = int(re.search(r":(\d+)\s*$", line).group(1))

    def conn(self):
        return http.client.HTTPConnection("127.0.0.1", self.port, timeout=120)

    def request(self, method, path, body=None, headers=None, conn=None):
        c = conn or self.conn()
        data = None if body is None else (body if isinstance(body, (bytes, str)) else json.dumps(body))
        c.request(method, path, body=data, headers=dict({"Content-Type": "application/json"}, **(headers or {})))
        r = c.getresponse()
        raw = r.read()
        if conn is None:
            c.close()
        return r, raw

    def close(self):
        if self.p.poll() is None:
            try:
                self.request("DELETE", "/destroy")
            except OSError:
                pass
            try:
                self.p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                self.p.kill()
        return self.p.returncode


@pytest.fixture()
def host(tmp_path):
    """The server in front of a stub engine library (tests/stubs/stub_engine.c): engine creation needs a GPU with the real one, and the
    transport does not."""
    lib = str(tmp_path / "libstub_engine.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-pthread", os.path.join(ROOT, "tests", "stubs", "stub_engine.c"), "-o", lib], check=True)
    h = Host(lib, {"STUB_ENGINE_STOP_FILE": str(tmp_path / "stopped")})
    h.stop_file = str(tmp_path / "stopped")
    yield h
    h.close()


def test_server_refuses_to_start_without_its_engine(tmp_path):
    p = subprocess.run([SERVER, "127.0.0.1", "0", "--lib", str(tmp_path / "missing.so")], capture_output=True, text=True, timeout=30)
    assert p.returncode == 2 and "cannot load the engine library" in p.stderr


def test_routes_and_error_paths_without_a_model(host):
    r, raw = host.request("GET", "/models", headers={"Origin": "http://example.test"})
    assert r.status == 200 and r.getheader("Access-Control-Allow-Origin") == "http://example.test"
    body = json.loads(raw)
    assert body["object"] == "list" and body["data"] == []
    assert r.getheader("Content-Type").startswith("application/json")
    # an unknown model: the engine's status code and message come through as they are (src/llama_engine.cc:435-446, 483-491)
    r, raw = host.request("POST", "/modelstatus", {"model": "nope"})
    assert r.status == 409 and "message" in json.loads(raw)
    r, raw = host.request("POST", "/v1/chat/completions", {"model": "nope", "messages": [{"role": "user", "content": "hi"}]})
    assert r.status == 409 and "message" in json.loads(raw)
    r, raw = host.request("POST", "/v1/chat/completions", {"model": "nope", "stream": True, "messages": [{"role": "user", "content": "hi"}]})
    assert r.status == 409 and "message" in json.loads(raw)                       # refused before the first token: JSON, not an event stream
    r, raw = host.request("POST", "/v1/embeddings", {"model": "nope", "input": "hi"})
    assert r.status == 409
    r, raw = host.request("POST", "/unloadmodel", {"model": "nope"})
    assert r.status in (400, 409) and "message" in json.loads(raw)
    # a model file that does not exist: 500 with a message (the load failed), and the server keeps serving
    r, raw = host.request("POST", "/loadmodel", {"llama_model_path": "/nonexistent/model.gguf", "model": "m"})
    assert r.status == 500 and "message" in json.loads(raw)
    # transport
    r, raw = host.request("GET", "/no/such/route")
    assert r.status == 404
    r, raw = host.request("GET", "/loadmodel")
    assert r.status == 405
    r, raw = host.request("GET", "/healthz")
    assert r.status == 200
    r, raw = host.request("POST", "/modelstatus", "this is not json")
    assert r.status in (400, 409) and "message" in json.loads(raw)                 # the engine answers for a body it cannot read
    # several requests on one connection
    c = host.conn()
    for _ in range(3):
        r, raw = host.request("GET", "/models", conn=c)
        assert r.status == 200 and json.loads(raw)["object"] == "list"
    c.close()
    # a request that is not HTTP
    s = socket.create_connection(("127.0.0.1", host.port), timeout=10)
    s.sendall(b"garbage\r\n\r\n")
    assert s.recv(64).startswith(b"HTTP/1.1 400")
    s.close()
    # a completion, whole and streamed: chunked text/event-stream carrying every callback's "data" string, in order, from the engine's thread
    r, raw = host.request("POST", "/v1/chat/completions", {"model": "m", "messages": []})
    assert r.status == 200 and json.loads(raw)["choices"][0]["message"]["content"] == "stub"
    c = host.conn()
    c.request("POST", "/v1/chat/completions", body=json.dumps({"model": "m", "stream": True, "messages": []}), headers={"Content-Type": "application/json"})
    r = c.getresponse()
    assert r.status == 200 and r.getheader("Content-Type") == "text/event-stream" and r.getheader("Transfer-Encoding") == "chunked"
    events = [e for e in r.read().decode().split("\n\n") if e]
    assert events == ['data: {"object":"chat.completion.chunk","i":%d}' % i for i in range(5)] + ["data: [DONE]"]
    r, raw = host.request("GET", "/models", conn=c)                                    # the connection survives a stream
    assert r.status == 200
    c.close()
    # a client that leaves mid-stream: the engine is told to stop generating for that model (ForceStopInferencing, server.cc:27-33)
    s = socket.create_connection(("127.0.0.1", host.port), timeout=10)
    body = json.dumps({"model": "long", "stream": True, "messages": []}).encode()
    s.sendall(b"POST /v1/chat/completions HTTP/1.1\r\nHost: x\r\nContent-Length: " + str(len(body)).encode() + b"\r\n\r\n" + body)
    assert s.recv(32).startswith(b"HTTP/1.1 200")
    s.setsockopt(socket.SOL_SOCKET, socket.SO_LINGER, b"\x01\x00\x00\x00\x00\x00\x00\x00")   # reset on close: the server's next write fails
    s.close()
    t0 = time.time()
    while not os.path.exists(host.stop_file) and time.time() - t0 < 20:
        time.sleep(0.05)
    assert open(host.stop_file).read() == "long"
    # a pre-flight
    r, raw = host.request("OPTIONS", "/v1/chat/completions", headers={"Origin": "http://example.test", "Access-Control-Request-Headers": "content-type"})
    assert r.status == 204 and r.getheader("Access-Control-Allow-Origin") == "http://example.test"
    assert host.close() == 0


@pytest.mark.gpu
def test_e2e_flow_over_http(pkg, tmp_path):
    path = str(tmp_path / "tiny-d128.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "tiny-d128", "q4_k_m", with_vocab=True)
    h = Host()
    try:
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": path, "model": "tiny", "ctx_len": 512, "n_parallel": 2, "ngl": 100, "user_prompt": "u:",
                                                 "ai_prompt": "a:", "system_prompt": "s:"})
        assert r.status == 200, raw
        r, raw = h.request("POST", "/modelstatus", {"model": "tiny"})
        assert r.status == 200 and "model_data" in json.loads(raw)
        r, raw = h.request("GET", "/models")
        assert [m["id"] for m in json.loads(raw)["data"]] == ["tiny"]
        req = {"model": "tiny", "messages": [{"role": "user", "content": "tell me"}], "max_tokens": 24, "temperature": 0.0, "repeat_penalty": 1.0,
               "frequency_penalty": 0.0, "presence_penalty": 0.0}
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200, raw
        whole = json.loads(raw)
        text = whole["choices"][0]["message"]["content"]
        assert whole["object"] == "chat.completion" and whole["usage"]["completion_tokens"] > 0
        # the same request as a stream: chunked text/event-stream, one "data: {json}" event per piece, "data: [DONE]" last; the pieces spell the same text
        c = h.conn()
        c.request("POST", "/v1/chat/completions", body=json.dumps(dict(req, stream=True)), headers={"Content-Type": "application/json"})
        r = c.getresponse()
        assert r.status == 200 and r.getheader("Content-Type") == "text/event-stream" and r.getheader("Transfer-Encoding") == "chunked"
        events = [e for e in r.read().decode("utf-8").split("\n\n") if e.strip()]
        c.close()
        assert events[-1].strip() == "data: [DONE]"
        pieces = []
        for e in events[:-1]:
            assert e.startswith("data: ")
            d = json.loads(e[6:])
            assert d["object"] == "chat.completion.chunk"
            pieces.append(d["choices"][0]["delta"].get("content") or "")
        assert "".join(pieces) == text
        # two clients at once (n_parallel = 2)
        import threading
        outs = [None, None]

        def one(i):
            outs[i] = h.request("POST", "/v1/chat/completions", req)
        th = [threading.Thread(target=one, args=(i,)) for i in range(2)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert all(o[0].status == 200 and json.loads(o[1])["choices"][0]["message"]["content"] == text for o in outs)
        # embeddings
        r, raw = h.request("POST", "/v1/embeddings", {"model": "tiny", "input": ["hello world", "tell me"]})
        assert r.status == 200, raw
        emb = json.loads(raw)
        assert emb["object"] == "list" and len(emb["data"]) == 2 and np.isfinite(np.asarray(emb["data"][0]["embedding"])).all()
        # a client that leaves mid-stream stops its generation; the server stays usable
        s = socket.create_connection(("127.0.0.1", h.port), timeout=30)
        body = json.dumps(dict(req, stream=True, max_tokens=400)).encode()
        s.sendall(b"POST /v1/chat/completions HTTP/1.1\r\nHost: x\r\nContent-Type: application/json\r\nContent-Length: " + str(len(body)).encode() + b"\r\n\r\n" + body)
        assert s.recv(32).startswith(b"HTTP/1.1 200")
        s.close()
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200 and json.loads(raw)["choices"][0]["message"]["content"] == text
        r, raw = h.request("POST", "/unloadmodel", {"model": "tiny"})
        assert r.status == 200
        r, raw = h.request("GET", "/models")
        assert json.loads(raw)["data"] == []
    finally:
        assert h.close() == 0


@pytest.mark.gpu
def test_row_split_model_over_http(pkg, tmp_path):
    """`"split_mode": "row"` in a /loadmodel body: the HTTP host needs nothing new - its engine library forms the group itself (two ranks
    sharing this box's GPU: worker process beside the server, shared-memory exchange), completions and a stream come back, /unloadmodel ends the worker and the
    server goes on to serve an ordinary model."""
    import psutil
    path = str(tmp_path / "tiny-e2048.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "tiny-e2048", "q4_k_m", with_vocab=True)
    h = Host()
    try:
        def workers():
            return [c for c in psutil.Process(h.p.pid).children(recursive=False) if "mi355_tp_worker" in (c.name() or "")]
        req = {"model": "tiny", "messages": [{"role": "user", "content": "tell me"}], "max_tokens": 16, "temperature": 0.0, "repeat_penalty": 1.0,
               "frequency_penalty": 0.0, "presence_penalty": 0.0}
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": path, "model": "tiny", "ctx_len": 512, "n_parallel": 2, "split_mode": "row", "split_ranks": 2})
        assert r.status == 200, raw
        assert len(workers()) == 1
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200, raw
        text = json.loads(raw)["choices"][0]["message"]["content"]
        assert json.loads(raw)["usage"]["completion_tokens"] == 16
        c = h.conn()
        c.request("POST", "/v1/chat/completions", body=json.dumps(dict(req, stream=True)), headers={"Content-Type": "application/json"})
        r = c.getresponse()
        events = [e for e in r.read().decode("utf-8").split("\n\n") if e.strip()]
        c.close()
        assert events[-1].strip() == "data: [DONE]"
        # (a stream holds an incomplete UTF-8 sequence back where the whole answer spells it as U+FFFD: these random models end mid-character now and then)
        assert "".join(json.loads(e[6:])["choices"][0]["delta"].get("content") or "" for e in events[:-1]).rstrip("\ufffd") == text.rstrip("\ufffd")
        # an uneven tensor_split is refused in the reference's load-error shape; the loaded model is not disturbed
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": path, "model": "uneven", "split_mode": "row", "tensor_split": [3, 1]})
        assert r.status == 500 and "even" in json.loads(raw).get("error", ""), raw
        r, raw = h.request("POST", "/unloadmodel", {"model": "tiny"})
        assert r.status == 200
        t0 = time.time()
        while workers() and time.time() - t0 < 15:
            time.sleep(0.1)
        assert not workers()
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": path, "model": "tiny", "ctx_len": 512})
        assert r.status == 200, raw
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200 and json.loads(raw)["usage"]["completion_tokens"] == 16
    finally:
        assert h.close() == 0


@pytest.mark.gpu
def test_reference_smoke_script_over_http(pkg, tmp_path):
    """.github/scripts/e2e-test-server-linux-and-mac.sh request by request against this host: load the LLM (upstream a TinyLlama Q2_K file, Makefile:5 - here its two-layer
    geometry in the same type mix) with ctx_len 50 / ngl 32, a streamed chat completion (max_tokens 50, temperature 0.1), unload, load the embedding model (upstream
    nomic-embed f16, an encoder - here two layers of its geometry) with "embedding": true / "model_type": "embedding", GET /models, POST /v1/embeddings.  The
    script's pass criterion is HTTP 200 on every step; the replies' shapes are checked too."""
    llm = str(tmp_path / "testllm.gguf"); emb = str(tmp_path / "test-embedding.gguf")
    pkg.gguf_synth.write_synthetic_llama(llm, "tiny-tl-2l", "q2_k", with_vocab=True)
    pkg.gguf_synth.write_synthetic_llama(emb, "nomic-embed-2l", "f16", with_vocab=True)
    h = Host()
    try:
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": llm, "model_alias": "testllm", "ctx_len": 50, "ngl": 32, "embedding": False})          # :50-58
        assert r.status == 200, raw
        c = h.conn()                                                                                                                                        # :66-84
        body = {"messages": [{"content": "Hello there", "role": "assistant"}, {"content": "Write a long and sad story for me", "role": "user"}],
                "stream": True, "model": "testllm", "max_tokens": 50, "stop": ["hello"], "frequency_penalty": 0, "presence_penalty": 0, "temperature": 0.1}
        c.request("POST", "/v1/chat/completions", body=json.dumps(body), headers={"Content-Type": "application/json", "Accept": "text/event-stream"})
        r = c.getresponse()
        assert r.status == 200
        events = [e for e in r.read().decode("utf-8", "replace").split("\n\n") if e.strip()]
        c.close()
        assert events[-1].strip() == "data: [DONE]" and len(events) >= 2
        r, raw = h.request("POST", "/unloadmodel", {"llama_model_path": llm, "model": "testllm"})                                                             # :87-91
        assert r.status == 200, raw
        r, raw = h.request("POST", "/loadmodel", {"llama_model_path": emb, "ctx_len": 50, "ngl": 32, "embedding": True, "model_type": "embedding"})            # :94-102
        assert r.status == 200, raw
        r, raw = h.request("GET", "/models")                                                                                                                 # :105-107
        assert r.status == 200 and [m["id"] for m in json.loads(raw)["data"]] == ["test-embedding"], raw
        r, raw = h.request("POST", "/v1/embeddings", {"input": "Hello", "model": "test-embedding", "encoding_format": "float"})                              # :110-120
        assert r.status == 200, raw
        d = json.loads(raw)
        v = np.asarray(d["data"][0]["embedding"], np.float64)
        assert d["object"] == "list" and v.shape == (768,) and abs(np.linalg.norm(v) - 1.0) < 1e-5
    finally:
        assert h.close() == 0


@pytest.mark.gpu
def test_image_request_over_http(pkg, tmp_path):
    """POST /loadmodel with `mmproj`, then a chat request whose user message carries an `image_url` piece (a data: URL): the answer over the socket is the one
    the engine surface gives for the same body (tests/test_gpu_llava.py checks that one against an independent decode loop and the oracle)."""
    import base64
    import io
    PIL = pytest.importorskip("PIL.Image")
    lm, mm = str(tmp_path / "tiny-d128.gguf"), str(tmp_path / "mmproj.gguf")
    pkg.gguf_synth.write_synthetic_llama(lm, "tiny-d128", "q4_k_m", with_vocab=True)
    pkg.gguf_synth.write_synthetic_clip(mm, "tiny-clip-1024")
    b = io.BytesIO()
    PIL.fromarray((np.random.default_rng(5).integers(0, 256, (50, 70, 3))).astype(np.uint8)).save(b, "PNG")
    url = "data:image/png;base64," + base64.b64encode(b.getvalue()).decode()
    req = {"model": "tiny", "messages": [{"role": "user", "content": [{"type": "text", "text": "what is in "}, {"type": "image_url", "image_url": {"url": url}}, {"type": "text", "text": " ?"}]}],
           "max_tokens": 8, "temperature": 0.0, "repeat_penalty": 1.0, "frequency_penalty": 0.0, "presence_penalty": 0.0}
    load = {"llama_model_path": lm, "mmproj": mm, "model": "tiny", "ctx_len": 512, "ngl": 100, "user_prompt": "u:", "ai_prompt": "a:", "system_prompt": "s:"}
    h = Host()
    try:
        r, raw = h.request("POST", "/loadmodel", load)
        assert r.status == 200, raw
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200, raw
        over_http = json.loads(raw)
        assert over_http["usage"]["prompt_tokens"] > 16                      # the image's 16 rows count as prompt positions
        r, raw = h.request("POST", "/v1/chat/completions", dict(req, messages=[{"role": "user", "content": [{"type": "image_url", "image_url": {"url": "data:image/png;base64,AAAA"}},
                                                                                                           {"type": "text", "text": "x"}]}]))
        assert r.status != 200 or "error" in raw.decode().lower(), raw       # bytes that are no image fail the request, not the server
        r, raw = h.request("POST", "/v1/chat/completions", req)
        assert r.status == 200 and json.loads(raw)["choices"][0]["message"]["content"] == over_http["choices"][0]["message"]["content"]
    finally:
        h.close()
    e = pkg.Engine()
    st, body = e.load_model(**load)
    assert st["status_code"] == 200, (st, body)
    st, body = e.chat_completion(**req)[-1]
    assert body["choices"][0]["message"]["content"] == over_http["choices"][0]["message"]["content"]
    e.close()
