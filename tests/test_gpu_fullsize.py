"""BASELINE's full size (Llama-3-8B-Instruct Q4_K_M shapes, 32 layers, 128 K vocabulary, q8_0 cache, n_ctx 4096; synthetic
weights): direct parity with the CPU oracle on a short prompt (the oracle needs ~0.5 s per token at this size), and the
size-independent properties of the decode path — determinism, graph replay == eager launches bit for bit, causality of a
prompt batch, device arg-max == arg-max of the host-visible logits — plus the row split of the full model over two ranks
sharing the GPU."""
import os

import numpy as np
import pytest

import oracle_py as oq
import test_gpu_tp as tpt

pytestmark = pytest.mark.gpu

FLIP_TOL = 3e-2


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.fixture(scope="module")
def big(pkg, tmp_path_factory):
    d = tmp_path_factory.mktemp("full")
    path = str(d / "llama-3-8b-q4_k_m.gguf")
    pkg.gguf_synth.write_synthetic_llama(path, "llama-3-8b", "q4_k_m", seed=0xC0FFEE, with_vocab=False)
    yield path
    try:
        os.remove(path)
    except OSError:
        pass


def test_fullsize_logits_match_oracle(pkg, big):
    """32 layers of 4096 / 14336-wide quantised activations: some int8 rounding flips on every token, so at this size the
    logits of two equally valid f32 summation orders differ at the 1e-2 level on every step.  That level is measured here on
    the CPU restatement itself (its sums re-associated, tests/test_oracle_sensitivity.py) and the HIP path must stay within
    the same band: FLIP_TOL, and no more than a small multiple of the CPU-vs-CPU figure."""
    pkg.Backend()
    m = pkg.Model(big)
    c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
    om = oq.OracleModel(big)
    nth = min(32, os.cpu_count() or 8)
    oc = oq.OracleContext(om, 64, 8, 8, True, nth)
    rng = np.random.default_rng(77)
    prompt = rng.integers(0, m.n_vocab, 10)
    c.decode(prompt, np.arange(10))
    ref = oc.decode(prompt, np.arange(10))[0]
    oq.set_assoc_variant(1)
    try:
        oc2 = oq.OracleContext(om, 64, 8, 8, True, nth)
        cpu_cpu = rel_err(oc2.decode(prompt, np.arange(10))[0], ref)
        oc2.close()
    finally:
        oq.set_assoc_variant(0)
    errs = [rel_err(c.logits(), ref)]
    tok = int(ref.argmax())
    for s in range(3):                                   # single-token steps: persistent mat-vec + single-launch attention, graphs
        c.decode([tok], [10 + s])
        r = oc.decode([tok], [10 + s])[0]
        g = c.logits()
        errs.append(rel_err(g, r))
        assert int(g.argmax()) == c.argmax()
        top2 = np.sort(r)[-2:]
        if top2[1] - top2[0] > 2 * FLIP_TOL * max(1.0, np.abs(r).max()):
            assert c.argmax() == int(r.argmax())
        tok = int(r.argmax())
    assert max(errs) <= FLIP_TOL, (errs, cpu_cpu)
    assert errs[0] <= max(4.0 * cpu_cpu, 5e-3), (errs, cpu_cpu)    # same order as the CPU's own re-association noise
    c.close(); m.close(); oc.close(); om.close()


def test_fullsize_determinism_graph_equals_eager_and_causality(pkg, big):
    pkg.Backend()
    m = pkg.Model(big)
    rng = np.random.default_rng(5)
    prompt = rng.integers(0, m.n_vocab, 512)

    def run(use_graphs):
        c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8, use_graphs=use_graphs)
        c.decode(prompt, np.arange(512))
        rows = [c.logits()]
        tok = c.argmax()
        toks = [tok]
        for s in range(24):
            c.decode([tok], [512 + s])
            rows.append(c.logits())
            assert int(rows[-1].argmax()) == c.argmax()
            tok = c.argmax()
            toks.append(tok)
        c.close()
        return np.stack(rows), toks

    a, ta = run(True)
    b, tb = run(True)
    e, te = run(False)
    assert np.array_equal(a, b) and ta == tb             # same calls, same bits
    assert np.array_equal(a, e) and ta == te             # hipGraph replay == eager launches
    # causality: row i of a batch depends on tokens 0..i only (another batch size takes other tiles and f32 orders, so the
    # agreement is at the rounding-flip level of this size, not bitwise)
    c = pkg.Context(m, n_ctx=4096, n_batch=2048, n_ubatch=2048, type_k=8, type_v=8)
    flags = np.zeros(96, np.int8); flags[63] = 1; flags[95] = 1
    c.decode(prompt[:96], np.arange(96), logits=flags)
    row63 = c.logits(63)
    c.kv_clear()
    c.decode(prompt[:64], np.arange(64))
    assert rel_err(c.logits(), row63) <= FLIP_TOL
    c.close(); m.close()


def test_fullsize_row_split_two_ranks(pkg, big, tmp_models):
    """Llama-3-8B cut two ways (16 heads + 4 KV heads, FF 7168 per rank), two processes sharing the GPU, host exchange."""
    pkg.Backend()
    om = oq.OracleModel(big)
    oc = oq.OracleContext(om, 64, 8, 8, True, min(32, os.cpu_count() or 8))
    rng = np.random.default_rng(9)
    prompt = rng.integers(0, om.n_vocab, 8).astype(np.int32)
    ref = [oc.decode(prompt, np.arange(8))[0]]
    steps = []
    for s in range(3):
        steps.append(int(ref[-1].argmax()))
        ref.append(oc.decode([steps[-1]], [8 + s])[0])
    oc.close(); om.close()
    plan = str(tmp_models / "tp-plan-full.npz")
    np.savez(plan, path=big, kv=8, transport="host", n_ctx=256, n_ubatch=64, prompt=prompt, steps=np.asarray(steps, np.int32),
             tail=np.zeros(0, np.int32))
    got = tpt.run_ranks(2, plan, str(tmp_models / "tp-out-full.npz"), timeout=900)
    lg = got["logits"]
    errs = [rel_err(a, b) for a, b in zip(lg, np.stack(ref))]
    assert int(got["n_head"]) == 16 and int(got["n_head_kv"]) == 4
    assert max(errs) <= FLIP_TOL, errs
