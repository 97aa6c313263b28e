"""CPU tests of the host-side mirror (tokenizer, sampler, slot loop, engine façade): the C++ unit-test program
tests/host/host_tests.cc is built with g++ against a deterministic fake arithmetic backend and must exit 0.
Covers what the reference's e2e suite checks at the HTTP surface: load/unload status codes, chat
completion body shape, stream framing and [DONE], usage counts, stop words, n_parallel batching, prompt cache,
context shift, KV-full error."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cortex.llamacpp_amd", "host")
SRCS = [os.path.join(ROOT, "tests", "host", "host_tests.cc")] + [
    os.path.join(HOST, f) for f in ("vocab.cc", "sampling.cc", "grammar.cc", "json_schema.cc", "server_context.cc", "engine.cc", "gguf.cc", "log.cc")]


@pytest.fixture(scope="module")
def host_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "host_tests")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-pthread", *SRCS, "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def test_host_logic_program(host_exe):
    exe = host_exe
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "all host-logic checks passed" in r.stdout


# The same program under the sanitizers (CPU builds only: no GPU sanitizers on this pool).  SURVEY.md §5: the reference documents races around its slot loop
# (src/llama_engine.cc:1269-1276); the host mirror's queues, slot states and the engine's model map are exercised by the multi-threaded checks of the program
# (parallel requests, stop / unload during generation), so a data race or a lifetime bug fails HERE instead of on a serving box.
@pytest.mark.parametrize("name,flags", [("tsan", ["-fsanitize=thread"]), ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])])
def test_host_logic_program_under_sanitizers(tmp_path, name, flags):
    exe = str(tmp_path / f"host_tests_{name}")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-Wall", "-pthread", *flags, *SRCS, "-o", exe],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-6000:]
    assert "all host-logic checks passed" in r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-6000:]


# ---------------------------------------------------------------------------------------------------------------------
# Tokenizer parity against HF `tokenizers` (an independent implementation of the same published algorithms): a small
# byte-level BPE is trained here with the Llama-3 pre-tokenizer pattern, written into a GGUF (tokenizer.ggml.model =
# "gpt2", tokens + merges as convert_hf_to_gguf lays them out), and the C++ tokenizer must produce the same ids.
LLAMA3_SPLIT = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|"
                r"\s+(?!\S)|\s+")
CORPUS = [
    "The quick brown fox jumps over the lazy dog. " * 4,
    "Hello, world! I'm here, you're there; they've gone and we'll stay. It's 2024, isn't it? He'd say so.",
    "def main():\n    print('hello')\n    return 12345 + 678\n\n\nclass Foo:\n\tpass\n",
    "Numbers: 1 22 333 4444 55555 1234567890, pi=3.14159; e=2.71828!!! ... ???",
    "naïve café résumé über straße — “quotes” … ¿qué? ¡sí! 日本語のテキスト 中文 한국어 emoji 🙂🙃",
    "   leading spaces and   multiple   spaces   \n\n  and trailing   ",
    "tabs\tand\r\nwindows\r\nnewlines \n \n x",
]
CASES = CORPUS[1:] + [
    "", " ", "  ", "a", " a", "a ", "Hello", " Hello", "Hello world", "HELLO'S 'tis DON'T", "x\n\ny", "x \n y", "1234", "12345678",
    "foo_bar-baz@example.com", "...!!!\n\n", "end with space ", "ünïcödé", "a b", "<|begin|>", "tail\n",
]


def test_bpe_tokenizer_matches_hf_tokenizers(tmp_path, pkg, host_exe):
    import json

    from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(LLAMA3_SPLIT), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=700, special_tokens=["<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    model = json.loads(tok.to_str())["model"]
    vocab = sorted(model["vocab"].items(), key=lambda kv: kv[1])
    tokens = [t for t, _ in vocab]
    merges = [m if isinstance(m, str) else " ".join(m) for m in model["merges"]]
    assert tokens[:3] == ["<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>"]

    w = pkg.gguf_synth.GGUFWriter()
    w.add("general.architecture", "str", "llama")
    w.add("tokenizer.ggml.model", "str", "gpt2")
    w.add("tokenizer.ggml.pre", "str", "llama-bpe")
    w.add_array("tokenizer.ggml.tokens", "str", tokens)
    w.add_array("tokenizer.ggml.token_type", "i32", [3, 3, 3] + [1] * (len(tokens) - 3))
    w.add_array("tokenizer.ggml.merges", "str", merges)
    w.add("tokenizer.ggml.bos_token_id", "u32", 0)
    w.add("tokenizer.ggml.eos_token_id", "u32", 1)
    w.add("tokenizer.ggml.eot_token_id", "u32", 2)
    gguf = str(tmp_path / "bpe.gguf")
    w.write(gguf)

    cases = list(CASES) + [{"text": "<|begin_of_text|>Hello<|eot_id|> there"}]
    cj = str(tmp_path / "cases.json")
    with open(cj, "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False)
    exe = host_exe
    r = subprocess.run([exe, "--tokenize", gguf, cj], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert got["bos"] == 0 and got["eos"] == 1 and got["n"] == len(tokens)
    for c, ids, rt in zip(cases, got["ids"], got["roundtrip"]):
        text = c["text"] if isinstance(c, dict) else c
        want = tok.encode(text, add_special_tokens=False).ids
        assert ids == want, (text, ids, want)
        assert rt == text, (text, rt)


def test_spm_tokenizer_synthetic_vocab(tmp_path, pkg, host_exe):
    """SentencePiece-style tokenizer on the synthetic vocab gguf_synth writes: greedy highest-score merges, U+2581 space
    prefix, byte fallback, and detokenize(tokenize(x)) == ' ' + x."""
    import json
    gguf = str(tmp_path / "tiny.gguf")
    pkg.gguf_synth.write_synthetic_llama(gguf, "tiny", "q8_0", with_vocab=True)
    cases = ["a", "ab cd", "hello world", "zz  top", "naïve ✓", "x\ny"]
    cj = str(tmp_path / "cases.json")
    with open(cj, "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False)
    exe = host_exe
    r = subprocess.run([exe, "--tokenize", gguf, cj], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    assert got["bos"] == 1 and got["eos"] == 2 and got["n"] == 512
    for text, ids, rt in zip(cases, got["ids"], got["roundtrip"]):
        assert rt == " " + text, (text, rt)
        assert all(0 <= i < 512 for i in ids)
    # 'ï' (C3 AF) has no piece: two byte-fallback tokens <0xC3> <0xAF> = ids 3 + 0xC3, 3 + 0xAF
    ids = got["ids"][4]
    assert 3 + 0xC3 in ids and 3 + 0xAF in ids
