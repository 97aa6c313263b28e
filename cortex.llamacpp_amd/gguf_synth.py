"""Synthetic GGUF v3 writer (tests / bench tooling, not product code).

There is no network and no real checkpoint on either box, so models are
synthesised: exact hparams / tensor names / shapes / `*_K_M` type mix of the named config
(SURVEY.md §A.4, §A.5), weights = seeded random *valid* quant blocks.  The files are ordinary
GGUF v3 and are loaded through the same loader as any unmodified GGUF.

Reference behaviour this feeds: `/loadmodel` -> LoadModelImpl (src/llama_engine.cc:547-732) ->
common_init_from_params (src/llama_server_context.cc:207).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

# ggml type ids
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K = 0, 1, 2, 8, 12, 13, 14
Q2_K, Q3_K = 10, 11
Q5_0, IQ4_NL = 6, 20
TYPE_NAME = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q2_K: "q2_K", Q3_K: "q3_K", Q5_0: "q5_0", IQ4_NL: "iq4_nl"}
BLOCK_ELEMS = {F32: 1, F16: 1, Q4_0: 32, Q8_0: 32, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q2_K: 256, Q3_K: 256, Q5_0: 32, IQ4_NL: 32}
BLOCK_BYTES = {F32: 4, F16: 2, Q4_0: 18, Q8_0: 34, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q2_K: 84, Q3_K: 110, Q5_0: 22, IQ4_NL: 18}

DT_Q4_0 = np.dtype([("d", "<f2"), ("qs", "u1", 16)])
DT_Q8_0 = np.dtype([("d", "<f2"), ("qs", "i1", 32)])
DT_Q4_K = np.dtype([("d", "<f2"), ("dmin", "<f2"), ("scales", "u1", 12), ("qs", "u1", 128)])
DT_Q5_K = np.dtype([("d", "<f2"), ("dmin", "<f2"), ("scales", "u1", 12), ("qh", "u1", 32), ("qs", "u1", 128)])
DT_Q6_K = np.dtype([("ql", "u1", 128), ("qh", "u1", 64), ("scales", "i1", 16), ("d", "<f2")])
DT_Q2_K = np.dtype([("scales", "u1", 16), ("qs", "u1", 64), ("d", "<f2"), ("dmin", "<f2")])
DT_Q3_K = np.dtype([("hmask", "u1", 32), ("qs", "u1", 64), ("scales", "u1", 12), ("d", "<f2")])
DT_Q5_0 = np.dtype([("d", "<f2"), ("qh", "u1", 4), ("qs", "u1", 16)])
BLOCK_DTYPE = {Q4_0: DT_Q4_0, Q8_0: DT_Q8_0, Q4_K: DT_Q4_K, Q5_K: DT_Q5_K, Q6_K: DT_Q6_K, Q2_K: DT_Q2_K, Q3_K: DT_Q3_K, Q5_0: DT_Q5_0, IQ4_NL: DT_Q4_0}
for _t, _dt in BLOCK_DTYPE.items():
    assert _dt.itemsize == BLOCK_BYTES[_t], (_t, _dt.itemsize)

# std of (dequantised weight / d) for uniformly random block payloads (derived in DESIGN.md)
# (Q2_K: w / d = sc * q - r * m with sc, m uniform 0..15, q uniform 0..3, r = 1.5: variance 144.7 + 2.25 * 21.25; Q3_K: (sc - 32) * q, sc 0..63, q -4..3)
_UNIT_STD = {Q4_0: 4.6, Q8_0: 73.9, Q4_K: 258.0, Q5_K: 527.0, Q6_K: 1367.0, Q2_K: 13.9, Q3_K: 43.3, Q5_0: 9.23, IQ4_NL: 67.2}   # (Q5_0: codes 0..31 minus 16; IQ4_NL: the sixteen code-book levels)
# dmin/d ratio that centres the weights of a random block on zero
_DMIN_RATIO = {Q4_K: 7.5, Q5_K: 15.5, Q2_K: 1.5}


def row_bytes(t: int, n: int) -> int:
    assert n % BLOCK_ELEMS[t] == 0, (t, n)
    return n // BLOCK_ELEMS[t] * BLOCK_BYTES[t]


def random_blocks(rng: np.random.Generator, t: int, n_elems: int, std: float) -> np.ndarray:
    """n_elems weights of ggml type t as raw bytes; dequantised std ~= `std`, mean ~= 0."""
    if t == F32:
        return (rng.standard_normal(n_elems, dtype=np.float32) * np.float32(std)).astype("<f4").view(np.uint8)
    if t == F16:
        return (rng.standard_normal(n_elems, dtype=np.float32) * np.float32(std)).astype("<f2").view(np.uint8)
    nb = n_elems // BLOCK_ELEMS[t]
    raw = rng.integers(0, 256, size=nb * BLOCK_BYTES[t], dtype=np.uint8)
    blk = raw.view(BLOCK_DTYPE[t])
    d = (rng.uniform(0.5, 1.5, size=nb) * (std / _UNIT_STD[t])).astype(np.float32)
    blk["d"] = d.astype("<f2")
    if t in _DMIN_RATIO:
        blk["dmin"] = (d * _DMIN_RATIO[t]).astype("<f2")
    return raw


# ---------------------------------------------------------------- container
_GT = {"u8": 0, "i8": 1, "u16": 2, "i16": 3, "u32": 4, "i32": 5, "f32": 6, "bool": 7, "str": 8, "arr": 9,
       "u64": 10, "i64": 11, "f64": 12}
_FMT = {0: "<B", 1: "<b", 2: "<H", 3: "<h", 4: "<I", 5: "<i", 6: "<f", 7: "<?", 10: "<Q", 11: "<q", 12: "<d"}


def _s(b: str | bytes) -> bytes:
    if isinstance(b, str):
        b = b.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


class GGUFWriter:
    """Streaming GGUF v3 writer: metadata first, tensor payloads generated one at a time."""

    def __init__(self, alignment: int = 32):
        self.alignment = alignment
        self.kv: list[bytes] = []
        self.tensors: list[tuple[str, tuple[int, ...], int, object]] = []

    def add(self, key: str, kind: str, value) -> None:
        t = _GT[kind]
        if kind == "str":
            payload = _s(value)
        else:
            payload = struct.pack(_FMT[t], value)
        self.kv.append(_s(key) + struct.pack("<I", t) + payload)

    def add_array(self, key: str, kind: str, values) -> None:
        et = _GT[kind]
        if kind == "str":
            body = b"".join(_s(v) for v in values)
        else:
            body = np.asarray(values).astype(np.dtype(_FMT[et][1:]).newbyteorder("<")).tobytes()
        self.kv.append(_s(key) + struct.pack("<I", 9) + struct.pack("<IQ", et, len(values)) + body)

    def add_tensor(self, name: str, ne: tuple[int, ...], t: int, data) -> None:
        """data: bytes-like / uint8 ndarray, or a zero-arg callable returning one (lazy)."""
        self.tensors.append((name, tuple(int(x) for x in ne), t, data))

    def write(self, path: str) -> int:
        al = self.alignment
        sizes, offs, off = [], [], 0
        for _, ne, t, _ in self.tensors:
            n = int(np.prod(ne))
            sz = row_bytes(t, ne[0]) * (n // ne[0])
            sizes.append(sz)
            offs.append(off)
            off += (sz + al - 1) // al * al
        with open(path, "wb") as f:
            f.write(struct.pack("<IIQQ", 0x46554747, 3, len(self.tensors), len(self.kv)))
            for kv in self.kv:
                f.write(kv)
            for (name, ne, t, _), o in zip(self.tensors, offs):
                f.write(_s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", x) for x in ne))
                f.write(struct.pack("<IQ", t, o))
            pos = f.tell()
            f.write(b"\0" * ((-pos) % al))
            # lazy tensors are generated by a few threads running ahead of the writer (numpy's generators release the GIL on bulk draws; every tensor has its
            # own seeded generator, so the bytes do not depend on who draws them or when), a bounded window of them in memory at a time
            from concurrent.futures import ThreadPoolExecutor
            n_workers = max(1, min(8, (os.cpu_count() or 2) - 1))
            window = 2 * n_workers
            with ThreadPoolExecutor(max_workers=n_workers) as ex:
                pending = {}

                def submit(i):
                    d = self.tensors[i][3]
                    pending[i] = ex.submit(d) if callable(d) else None

                nt = len(self.tensors)
                for i in range(min(window, nt)):
                    submit(i)
                for i, ((name, ne, t, data), sz) in enumerate(zip(self.tensors, sizes)):
                    fut = pending.pop(i)
                    buf = fut.result() if fut is not None else data
                    if i + window < nt:
                        submit(i + window)
                    buf = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.view(np.uint8).reshape(-1)
                    assert buf.size == sz, (name, buf.size, sz)
                    f.write(memoryview(buf))
                    f.write(b"\0" * ((-sz) % al))
                    del buf
            return f.tell()


# ---------------------------------------------------------------- model configs
@dataclass
class LlamaConfig:
    name: str
    n_embd: int
    n_layer: int
    n_head: int
    n_head_kv: int
    n_ff: int
    n_vocab: int
    rope_base: float = 10000.0
    eps: float = 1e-5
    n_ctx_train: int = 4096
    n_expert: int = 0
    n_expert_used: int = 0
    big_model: bool = False          # 70B-class rule: attn_v Q4_K -> Q5_K on the non-Q6_K layers
    arch: str = "llama"              # general.architecture: "qwen2" is the same graph with NEOX rope pairing and (qkv_bias) attention biases
    qkv_bias: bool = False
    extra: dict = field(default_factory=dict)

    @property
    def head_dim(self) -> int:
        return self.n_embd // self.n_head


CONFIGS = {
    # BASELINE.json configs (SURVEY.md §8a sizes)
    "tinyllama-1.1b": LlamaConfig("TinyLlama-1.1B-Chat", 2048, 22, 32, 4, 5632, 32000, 10000.0, 1e-5, 2048),
    "llama-2-7b": LlamaConfig("Llama-2-7B-Chat", 4096, 32, 32, 32, 11008, 32000, 10000.0, 1e-5, 4096),
    "llama-3-8b": LlamaConfig("Llama-3-8B-Instruct", 4096, 32, 32, 8, 14336, 128256, 500000.0, 1e-5, 8192),
    "mixtral-8x7b": LlamaConfig("Mixtral-8x7B-Instruct", 4096, 32, 32, 8, 14336, 32000, 1e6, 1e-5, 32768, 8, 2),
    "llama-3-70b": LlamaConfig("Llama-3-70B-Instruct", 8192, 80, 64, 8, 28672, 128256, 500000.0, 1e-5, 8192,
                               big_model=True),
    # qwen2 architecture (NEOX rope, Q / K / V biases), 7 query heads per kv head, hidden size and feed-forward width that are no multiples of 1024
    "qwen2-7b": LlamaConfig("Qwen2-7B-Instruct", 3584, 28, 28, 4, 18944, 152064, 1e6, 1e-6, 32768, arch="qwen2", qkv_bias=True),
    # the reference's embedding smoke model (Makefile:6): nomic-embed-text-v1.5's geometry, a bidirectional encoder (general.architecture nomic-bert)
    "nomic-embed": LlamaConfig("nomic-embed-text-v1.5", 768, 12, 12, 12, 3072, 30522, 1000.0, 1e-12, 2048, arch="nomic-bert"),
    # test-sized
    "tiny": LlamaConfig("tiny-test", 256, 2, 4, 2, 512, 512, 10000.0, 1e-5, 512),
    "tiny-gqa4": LlamaConfig("tiny-gqa4", 512, 3, 8, 2, 1024, 768, 500000.0, 1e-5, 1024),
    "tiny-d128": LlamaConfig("tiny-d128", 1024, 2, 8, 2, 2048, 512, 500000.0, 1e-5, 1024),      # head_dim 128, GQA 4:1
    "tiny-d128-mha": LlamaConfig("tiny-d128-mha", 512, 2, 4, 4, 1024, 512, 10000.0, 1e-5, 1024),  # head_dim 128, MHA
    "tiny-moe": LlamaConfig("tiny-moe", 256, 2, 4, 2, 512, 512, 1e6, 1e-5, 512, 8, 2),
    # hidden sizes that are multiples of 2048: the shapes the persistent single-token mat-vec (fused RMSNorm / quantise
    # prologues) and the MFMA prefill planes are built for
    "tiny-e2048": LlamaConfig("tiny-e2048", 2048, 2, 16, 4, 4096, 512, 500000.0, 1e-5, 1024),
    # TinyLlama-1.1B's layer geometry (the reference's smoke model, Makefile:5): head_dim 64, 8 query heads per kv head, FF = 5632 = 5.5 x 1024
    "tiny-tl-2l": LlamaConfig("tiny-tl-2l", 2048, 2, 32, 4, 5632, 512, 10000.0, 1e-5, 1024),
    # 8 KV heads of 128 (one 1024-wide K / V row, GQA 2:1): the shape the single-launch decode attention (rope + cache
    # store + split attention + merge) and the chunk-list batched steps are built for — Llama-3-8B's own KV geometry
    # Llama-3-8B's layer geometry (4096 / 14336, 32 heads over 8 KV heads), two layers: the whole-step kernel's shape
    "tiny-8b-2l": LlamaConfig("tiny-8b-2l", 4096, 2, 32, 8, 14336, 512, 500000.0, 1e-5, 1024),
    # Llama-3-8B's ATTENTION geometry (4096, 32 heads over 8 kv heads) with a narrow feed-forward: the ctx-4096 parity case needs a 3968-token prompt on the
    # CPU side, and four fifths of a layer's CPU time are its feed-forward, which does not depend on the context length (tests/test_gpu_model.py)
    "tiny-8b-attn-2l": LlamaConfig("tiny-8b-attn-2l", 4096, 2, 32, 8, 2048, 512, 500000.0, 1e-5, 4096),
    # two layers of Llama-3-70B's geometry (BASELINE config 5): split over 8 ranks a rank holds 8 query heads on ONE kv head, attn_output columns of 1024 = 4
    # super-blocks, ffn_down columns of 3584 = 14 super-blocks (and the 28672-wide ffn_down's column halves meet the column slicing)
    "tiny-70b-2l": LlamaConfig("tiny-70b-2l", 8192, 2, 64, 8, 28672, 512, 500000.0, 1e-5, 1024),
    "tiny-g8": LlamaConfig("tiny-g8", 2048, 2, 16, 8, 4096, 512, 500000.0, 1e-5, 1024),
    # Llama-2-7B's ATTENTION geometry (4096, 32 heads, 32 kv heads: BASELINE config 2) with a narrow feed-forward, two layers: a workgroup's share of the 12288 Q | K | V
    # rows is 140 KB - more than the fused attention-block launch (csrc/attn_out.hip) can hold: the two launches (tools/r6_qf_ring.patch streamed it through a ring: slower)
    "tiny-7b-attn-2l": LlamaConfig("tiny-7b-attn-2l", 4096, 2, 32, 32, 2048, 512, 10000.0, 1e-5, 1024),
    # Llama-3-70B's feed-forward width on a narrow model: 28672 has no weight-stream form, its single-token ffn_down runs as two column halves of 14336
    "tiny-ff28k": LlamaConfig("tiny-ff28k", 1024, 2, 8, 2, 28672, 512, 500000.0, 1e-5, 1024),
    # three layers of Llama-3-8B's geometry: two launches of the layer engine (decode_engine.hip) that hand Q | K | V on + the last one
    # general.architecture "qwen2": NEOX rope pairing + Q / K / V biases (one of the reference's weekend test families, e2e-test-server-weekend.py:22-76).
    # tiny-qwen2: head_dim 64; tiny-qwen2-1.5b-2l / -7b-2l: two layers of Qwen2-1.5B's (1536, 12 / 2 heads of 128, 8960) and Qwen2-7B's (3584, 28 / 4, 18944) geometry
    "tiny-qwen2": LlamaConfig("tiny-qwen2", 512, 3, 8, 2, 1024, 768, 1e6, 1e-6, 1024, arch="qwen2", qkv_bias=True),
    "tiny-qwen2-1.5b-2l": LlamaConfig("tiny-qwen2-1.5b-2l", 1536, 2, 12, 2, 8960, 512, 1e6, 1e-6, 1024, arch="qwen2", qkv_bias=True),
    "tiny-qwen2-7b-2l": LlamaConfig("tiny-qwen2-7b-2l", 3584, 2, 28, 4, 18944, 512, 1e6, 1e-6, 1024, arch="qwen2", qkv_bias=True),
    # three and five query heads of 128 per kv head (Llama-3.2-3B's 24 / 8; 40 / 8): the attention kernels' odd head ratios on prompts and steps
    "tiny-r3": LlamaConfig("tiny-r3", 1536, 2, 12, 4, 4096, 512, 500000.0, 1e-5, 1024),
    "tiny-r5": LlamaConfig("tiny-r5", 1280, 2, 10, 2, 3584, 512, 500000.0, 1e-5, 1024),
    # YaRN rope scaling read from the file (factor 4 over an original context of 32 positions, attn_factor 0.9): golden fixture + parity case
    "tiny-yarn": LlamaConfig("tiny-yarn", 512, 3, 8, 2, 1024, 768, 500000.0, 1e-5, 1024,
                             extra={"rope.scaling.type": "yarn", "rope.scaling.factor": 4.0, "rope.scaling.original_context_length": 32, "rope.scaling.attn_factor": 0.9}),
    # general.architecture nomic-bert (the reference's embedding smoke model is nomic-embed-text-v1.5, Makefile:6): a bidirectional encoder, embeddings only.
    # tiny-nomic: head_dim 64 as the real one; nomic-embed-2l: two layers of its geometry (768, 12 heads of 64, 3072, 30522 WordPiece tokens, rope base 1000)
    "tiny-nomic": LlamaConfig("tiny-nomic", 256, 2, 4, 4, 512, 512, 1000.0, 1e-12, 2048, arch="nomic-bert"),
    "nomic-embed-2l": LlamaConfig("nomic-embed-2l", 768, 2, 12, 12, 3072, 30522, 1000.0, 1e-12, 2048, arch="nomic-bert"),
    # a mixture-of-experts file whose expert tensors are wide enough for the weight-stream mat-vec (hidden 2048, rows of >= 1 KiB): 8 experts, 2 used
    "tiny-moe-e2048": LlamaConfig("tiny-moe-e2048", 2048, 2, 16, 4, 4096, 512, 1e6, 1e-5, 1024, 8, 2),
    "tiny-8b-3l": LlamaConfig("tiny-8b-3l", 4096, 3, 32, 8, 14336, 512, 500000.0, 1e-5, 1024),
}

FTYPE_ID = {"f16": 1, "q4_0": 2, "q5_0": 8, "iq4_nl": 25, "q8_0": 7, "q4_k_m": 15, "q5_k_m": 17, "q2_k": 10, "q3_k_s": 11, "q3_k_m": 12, "q3_k_l": 13, "q4_k_s": 14, "q5_k_s": 16, "q6_k": 18}


def use_more_bits(i: int, n: int) -> bool:
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def tensor_type(cfg: LlamaConfig, ftype: str, kind: str, il: int) -> int:
    """ggml type of a 2-D weight per llama-quantize's `*_K_M` mix."""
    if ftype == "f16":
        return F16
    if ftype == "q8_0":
        return Q8_0
    if ftype in ("q2_k", "q3_k_m"):
        # llama-quantize's rules for LLAMA_FTYPE_MOSTLY_Q2_K / Q3_K_M on the llama architecture (the reference's e2e smoke model is a TinyLlama Q2_K
        # file, /root/reference/Makefile:5): output Q6_K; Q2_K: attn_v Q4_K with a query / kv head ratio >= 4 (else Q3_K), ffn_down and attn_output
        # Q3_K; Q3_K_M: attn_v Q5_K in the first two layers then Q4_K, ffn_down Q5_K in the first n_layer / 16 layers then Q4_K, attn_output Q4_K
        q2 = ftype == "q2_k"
        if kind == "output":
            return Q6_K
        if kind == "attn_v":
            if q2:
                return Q4_K if cfg.n_head // cfg.n_head_kv >= 4 else Q3_K
            return Q5_K if il < 2 else Q4_K
        if kind == "ffn_down":
            return Q3_K if q2 else (Q5_K if il < cfg.n_layer // 16 else Q4_K)
        if kind == "attn_output":
            return Q3_K if q2 else Q4_K
        return Q2_K if q2 else Q3_K
    if ftype in ("q4_0", "q5_0", "iq4_nl"):
        # the 32-element formats (llama-quantize's legacy types and the fallbacks it takes for rows that are not a multiple of 256): output Q6_K;
        # IQ4_NL promotes attn_v (grouped-query models) and the first eighth of ffn_down to Q5_K, as upstream does for IQ4_NL / IQ4_XS
        if kind == "output":
            return Q6_K
        if ftype == "iq4_nl" and ((kind == "attn_v" and cfg.n_head // cfg.n_head_kv >= 4) or (kind == "ffn_down" and il < max(1, cfg.n_layer // 8))):
            return Q5_K
        return {"q4_0": Q4_0, "q5_0": Q5_0, "iq4_nl": IQ4_NL}[ftype]
    if ftype in ("q3_k_s", "q3_k_l", "q4_k_s", "q5_k_s", "q6_k"):
        # the other mixes the reference publishes (.github/workflows/convert-model-all-quant.yml:106-152), by llama-quantize's rules for the llama
        # architecture: output Q6_K everywhere; Q3_K_S: everything else Q3_K; Q3_K_L: attn_v, ffn_down, attn_output Q5_K, the rest Q3_K; Q4_K_S: Q4_K
        # with attn_v Q5_K in the first four layers and ffn_down Q5_K in the first n_layer / 8; Q5_K_S: Q5_K; Q6_K: Q6_K
        if kind == "output" or ftype == "q6_k":
            return Q6_K
        if ftype == "q3_k_s":
            return Q3_K
        if ftype == "q3_k_l":
            return Q5_K if kind in ("attn_v", "ffn_down", "attn_output") else Q3_K
        if ftype == "q4_k_s":
            if (kind == "attn_v" and il < 4) or (kind == "ffn_down" and il < max(1, cfg.n_layer // 8)):
                return Q5_K
            return Q4_K
        return Q5_K
    base = {"q4_k_m": Q4_K, "q5_k_m": Q5_K}[ftype]
    if kind == "output":
        return Q6_K
    if kind in ("attn_v", "ffn_down") and use_more_bits(il, cfg.n_layer):
        return Q6_K
    if cfg.n_expert == 8 and kind in ("attn_k", "attn_v"):
        return Q8_0
    if cfg.big_model and kind == "attn_v" and base == Q4_K:
        return Q5_K
    return base


def model_tensors(cfg: LlamaConfig, ftype: str):
    """[(name, ne, type, fan_in)] in file order."""
    E, F, V = cfg.n_embd, cfg.n_ff, cfg.n_vocab
    kv = cfg.n_head_kv * cfg.head_dim
    out = [("token_embd.weight", (E, V), tensor_type(cfg, ftype, "token_embd", 0), None)]
    if cfg.arch == "nomic-bert":
        # encoder: token-type table, LayerNorms with biases, fused Q | K | V; no output head (the reference's embedding smoke model, Makefile:6)
        out.append(("token_types.weight", (E, 2), F32, E))
        out.append(("token_embd_norm.weight", (E,), F32, None))
        out.append(("token_embd_norm.bias", (E,), F32, None))
        for il in range(cfg.n_layer):
            p = f"blk.{il}."
            out.append((p + "attn_qkv.weight", (E, E + 2 * kv), tensor_type(cfg, ftype, "attn_qkv", il), E))
            out.append((p + "attn_output.weight", (E, E), tensor_type(cfg, ftype, "attn_output", il), E))
            out.append((p + "attn_output_norm.weight", (E,), F32, None))
            out.append((p + "attn_output_norm.bias", (E,), F32, None))
            out.append((p + "ffn_up.weight", (E, F), tensor_type(cfg, ftype, "ffn_up", il), E))
            out.append((p + "ffn_gate.weight", (E, F), tensor_type(cfg, ftype, "ffn_gate", il), E))
            out.append((p + "ffn_down.weight", (F, E), tensor_type(cfg, ftype, "ffn_down", il), F))
            out.append((p + "layer_output_norm.weight", (E,), F32, None))
            out.append((p + "layer_output_norm.bias", (E,), F32, None))
        return out
    for il in range(cfg.n_layer):
        p = f"blk.{il}."
        out.append((p + "attn_norm.weight", (E,), F32, None))
        out.append((p + "attn_q.weight", (E, E), tensor_type(cfg, ftype, "attn_q", il), E))
        out.append((p + "attn_k.weight", (E, kv), tensor_type(cfg, ftype, "attn_k", il), E))
        out.append((p + "attn_v.weight", (E, kv), tensor_type(cfg, ftype, "attn_v", il), E))
        if cfg.qkv_bias:
            out.append((p + "attn_q.bias", (E,), F32, None))
            out.append((p + "attn_k.bias", (kv,), F32, None))
            out.append((p + "attn_v.bias", (kv,), F32, None))
        out.append((p + "attn_output.weight", (E, E), tensor_type(cfg, ftype, "attn_output", il), E))
        out.append((p + "ffn_norm.weight", (E,), F32, None))
        if cfg.n_expert:
            X = cfg.n_expert
            out.append((p + "ffn_gate_inp.weight", (E, X), F32, E))
            out.append((p + "ffn_gate_exps.weight", (E, F, X), tensor_type(cfg, ftype, "ffn_gate", il), E))
            out.append((p + "ffn_down_exps.weight", (F, E, X), tensor_type(cfg, ftype, "ffn_down", il), F))
            out.append((p + "ffn_up_exps.weight", (E, F, X), tensor_type(cfg, ftype, "ffn_up", il), E))
        else:
            out.append((p + "ffn_gate.weight", (E, F), tensor_type(cfg, ftype, "ffn_gate", il), E))
            out.append((p + "ffn_down.weight", (F, E), tensor_type(cfg, ftype, "ffn_down", il), F))
            out.append((p + "ffn_up.weight", (E, F), tensor_type(cfg, ftype, "ffn_up", il), E))
    out.append(("output_norm.weight", (E,), F32, None))
    out.append(("output.weight", (E, V), tensor_type(cfg, ftype, "output", 0), E))
    return out


def weight_bytes_per_token(cfg: LlamaConfig, ftype: str) -> int:
    """Algorithmic weight bytes one decoded token reads: every tensor once, one
    token_embd row, and n_expert_used of n_expert experts."""
    total = 0
    for name, ne, t, _ in model_tensors(cfg, ftype):
        n = int(np.prod(ne))
        b = row_bytes(t, ne[0]) * (n // ne[0])
        if name == "token_embd.weight":
            b = row_bytes(t, ne[0])
        elif "_exps." in name:
            b = b // cfg.n_expert * cfg.n_expert_used
        total += b
    return total


def write_synthetic_llama(path: str, cfg: LlamaConfig | str, ftype: str = "q4_k_m", seed: int = 0xC0FFEE,
                          with_vocab: bool = True, gain: float = 1.0) -> int:
    """Write a synthetic llama-arch GGUF; returns the file size in bytes."""
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    w = GGUFWriter()
    a = cfg.arch
    w.add("general.architecture", "str", a)
    w.add("general.name", "str", cfg.name + " (synthetic)")
    w.add("general.file_type", "u32", FTYPE_ID[ftype])
    w.add(f"{a}.context_length", "u32", cfg.n_ctx_train)
    w.add(f"{a}.embedding_length", "u32", cfg.n_embd)
    w.add(f"{a}.block_count", "u32", cfg.n_layer)
    w.add(f"{a}.feed_forward_length", "u32", cfg.n_ff)
    w.add(f"{a}.attention.head_count", "u32", cfg.n_head)
    w.add(f"{a}.attention.head_count_kv", "u32", cfg.n_head_kv)
    if cfg.arch == "nomic-bert":
        w.add(f"{a}.attention.layer_norm_epsilon", "f32", cfg.eps)
        w.add(f"{a}.attention.causal", "bool", False)
        w.add(f"{a}.pooling_type", "u32", 1)
    else:
        w.add(f"{a}.attention.layer_norm_rms_epsilon", "f32", cfg.eps)
    w.add(f"{a}.rope.dimension_count", "u32", cfg.head_dim)
    w.add(f"{a}.rope.freq_base", "f32", cfg.rope_base)
    w.add(f"{a}.vocab_size", "u32", cfg.n_vocab)
    if cfg.n_expert:
        w.add(f"{a}.expert_count", "u32", cfg.n_expert)
        w.add(f"{a}.expert_used_count", "u32", cfg.n_expert_used)
    for k, v in cfg.extra.items():                 # e.g. {"pooling_type": 1, "rope.scaling.type": "yarn", "rope.scaling.factor": 4.0}: extra keys under the architecture prefix
        w.add(f"{a}.{k}", "str" if isinstance(v, str) else "f32" if isinstance(v, float) else "u32", v)
    if with_vocab and cfg.arch == "nomic-bert":
        # synthetic WordPiece vocabulary in the GGUF convention (word-initial pieces carry the U+2581 prefix, continuations are bare): the five BERT specials,
        # letters / digits / punctuation in both roles, then generated multi-letter pieces
        toks, types = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"], [3, 3, 3, 3, 3]
        alphabet = "abcdefghijklmnopqrstuvwxyz0123456789"
        for ch in alphabet + ".,!?'-":
            toks += ["\u2581" + ch, ch]
            types += [1, 1]
        i = 0
        while len(toks) < cfg.n_vocab:
            s, k = "", i + 36
            while True:
                s = alphabet[k % 26] + s
                k //= 26
                if k == 0:
                    break
            piece = ("\u2581" + s) if i % 2 == 0 else s
            if piece not in toks[:100] and len(s) > 1:
                toks.append(piece)
                types.append(1)
            i += 1
        w.add("tokenizer.ggml.model", "str", "bert")
        w.add_array("tokenizer.ggml.tokens", "str", toks[: cfg.n_vocab])
        w.add_array("tokenizer.ggml.token_type", "i32", types[: cfg.n_vocab])
        w.add("tokenizer.ggml.token_type_count", "u32", 2)
        w.add("tokenizer.ggml.unknown_token_id", "u32", 1)
        w.add("tokenizer.ggml.seperator_token_id", "u32", 3)
        w.add("tokenizer.ggml.padding_token_id", "u32", 0)
        w.add("tokenizer.ggml.cls_token_id", "u32", 2)
        w.add("tokenizer.ggml.mask_token_id", "u32", 4)
        w.add("tokenizer.ggml.bos_token_id", "u32", 2)
        w.add("tokenizer.ggml.eos_token_id", "u32", 3)
    elif with_vocab:
        # synthetic SentencePiece-style vocab: <unk>,<s>,</s>, 256 byte tokens, then printable pieces
        toks, scores, types = ["<unk>", "<s>", "</s>"], [0.0, 0.0, 0.0], [2, 3, 3]
        for b in range(256):
            toks.append(f"<0x{b:02X}>")
            scores.append(0.0)
            types.append(6)
        alphabet = "abcdefghijklmnopqrstuvwxyz"
        for piece in ["▁"] + list(alphabet):
            toks.append(piece)
            scores.append(-1000.0)
            types.append(1)
        i = 26
        while len(toks) < cfg.n_vocab:
            s, k = "", i
            while True:
                s = alphabet[k % 26] + s
                k //= 26
                if k == 0:
                    break
            toks.append(("▁" + s) if i % 2 == 0 else s)
            scores.append(-float(len(toks)))
            types.append(1)
            i += 1
        w.add("tokenizer.ggml.model", "str", "llama")
        w.add_array("tokenizer.ggml.tokens", "str", toks[: cfg.n_vocab])
        w.add_array("tokenizer.ggml.scores", "f32", scores[: cfg.n_vocab])
        w.add_array("tokenizer.ggml.token_type", "i32", types[: cfg.n_vocab])
        w.add("tokenizer.ggml.bos_token_id", "u32", 1)
        w.add("tokenizer.ggml.eos_token_id", "u32", 2)
        w.add("tokenizer.ggml.unknown_token_id", "u32", 0)
        w.add("tokenizer.ggml.add_bos_token", "bool", True)
        w.add("tokenizer.ggml.add_eos_token", "bool", False)

    for idx, (name, ne, t, fan_in) in enumerate(model_tensors(cfg, ftype)):
        n = int(np.prod(ne))

        def gen(idx=idx, name=name, ne=ne, t=t, fan_in=fan_in, n=n):
            rng = np.random.default_rng([seed, idx])
            if name.endswith(".bias"):
                return (rng.standard_normal(n) * 0.25).astype("<f4").view(np.uint8)
            if len(ne) == 1:  # norm weights
                return rng.uniform(0.9, 1.1, size=n).astype("<f4").view(np.uint8)
            if name == "token_embd.weight":
                return random_blocks(rng, t, n, 1.0)
            return random_blocks(rng, t, n, gain / np.sqrt(fan_in))

        w.add_tensor(name, ne, t, gen)
    return w.write(path)


# ---------------------------------------------------------------- LLaVA projector files ("mmproj": general.architecture clip)
@dataclass
class ClipConfig:
    """The vision tower + projector of a LLaVA-1.5 style mmproj file as llama.cpp's convert_image_encoder_to_gguf.py writes it: CLIP ViT (pre-LN, learned position
    embeddings, class token, quick-GELU MLP) cut after its second-to-last layer (block_count = layers - 1), and the two-layer MLP projector mm.0 / mm.2.
    clip.cpp then runs block_count - 1 of the file's blocks for a LLaVA projector (get_deepest_feature_layer): the file's last block is dead weight."""
    name: str
    image_size: int
    patch_size: int
    n_embd: int          # vision hidden size
    n_head: int
    n_ff: int
    n_layer: int         # blocks in the file
    proj_dim: int        # the language model's n_embd
    eps: float = 1e-5
    use_gelu: bool = False
    mean: tuple = (0.48145466, 0.4578275, 0.40821073)
    std: tuple = (0.26862954, 0.26130258, 0.27577711)
    # LLaVA-1.6: the (width, height) canvases of the image grid, flattened, and the merge type that turns the grid on
    pinpoints: tuple = ()
    merge_type: str = ""

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2


CLIP_CONFIGS = {
    # ViT-L/14-336 as LLaVA-1.5-7B uses it (576 patches -> 576 rows of 4096)
    "clip-vit-l-336": ClipConfig("clip-vit-large-patch14-336", 336, 14, 1024, 16, 4096, 23, 4096),
    # the same graph at test size: 56 x 56 pixels = 16 patches of 14 x 14, head_dim 64 like the real tower (d128: 36 patches, 4 heads, 3 blocks)
    "tiny-clip": ClipConfig("tiny-clip", 56, 14, 128, 2, 256, 3, 256),
    "tiny-clip-d128": ClipConfig("tiny-clip-d128", 84, 14, 256, 4, 512, 4, 512),
    # head size 128 at the real tower's 577 rows (the tiled attention's largest LDS footprint: 157 KB)
    "clip-d128-336": ClipConfig("clip-d128-336", 336, 14, 256, 2, 512, 2, 256),
    "tiny-clip-gelu": ClipConfig("tiny-clip-gelu", 56, 14, 128, 2, 256, 3, 4096, use_gelu=True),
    # a projector as wide as the tiny language models of the engine tests (n_embd 1024): 16 rows per image
    "tiny-clip-1024": ClipConfig("tiny-clip-1024", 56, 14, 128, 2, 256, 3, 1024),
    # LLaVA-1.6: ViT-L/14-336 with the image grid of llava-v1.6 (an overview + up to four tiles: 2880 rows a picture), and the same at test size
    "clip-vit-l-336-grid": ClipConfig("llava-v1.6-clip", 336, 14, 1024, 16, 4096, 23, 4096, pinpoints=(336, 672, 672, 336, 672, 672, 1008, 336, 336, 1008),
                                      merge_type="spatial_unpad"),
    "tiny-clip-grid": ClipConfig("tiny-clip-grid", 56, 14, 128, 2, 256, 3, 256, pinpoints=(56, 112, 112, 56, 112, 112, 168, 56, 56, 168), merge_type="spatial_unpad"),
    "tiny-clip-grid-1024": ClipConfig("tiny-clip-grid-1024", 56, 14, 128, 2, 256, 3, 1024, pinpoints=(56, 112, 112, 56, 112, 112, 168, 56, 56, 168),
                                      merge_type="spatial_unpad"),
    # a grid in the file but the merge type "flat": only the overview is encoded
    "tiny-clip-grid-flat": ClipConfig("tiny-clip-grid-flat", 56, 14, 128, 2, 256, 3, 256, pinpoints=(56, 112, 112, 56, 112, 112), merge_type="flat"),
}


def clip_tensors(cfg: ClipConfig):
    """(name, ne, type, fan_in); ne in ggml order (fastest dimension first).  The feed-forward names are the converter's: `ffn_down` is the FIRST projection
    (n_embd -> n_ff) and `ffn_up` the second - clip.cpp reads them that way round."""
    E, FF, P = cfg.n_embd, cfg.n_ff, cfg.patch_size
    out = [("v.class_embd", (E,), F32, 1),
           ("v.patch_embd.weight", (P, P, 3, E), F16, 3 * P * P),
           ("v.position_embd.weight", (E, cfg.n_patches + 1), F16, 1),
           ("v.pre_ln.weight", (E,), F32, 1), ("v.pre_ln.bias", (E,), F32, 1)]
    for il in range(cfg.n_layer):
        p = f"v.blk.{il}."
        for nm in ("attn_q", "attn_k", "attn_v", "attn_out"):
            out += [(p + nm + ".weight", (E, E), F16, E), (p + nm + ".bias", (E,), F32, 1)]
        out += [(p + "ln1.weight", (E,), F32, 1), (p + "ln1.bias", (E,), F32, 1)]
        out += [(p + "ffn_down.weight", (E, FF), F16, E), (p + "ffn_down.bias", (FF,), F32, 1)]
        out += [(p + "ffn_up.weight", (FF, E), F16, FF), (p + "ffn_up.bias", (E,), F32, 1)]
        out += [(p + "ln2.weight", (E,), F32, 1), (p + "ln2.bias", (E,), F32, 1)]
    out += [("mm.0.weight", (E, cfg.proj_dim), F16, E), ("mm.0.bias", (cfg.proj_dim,), F32, 1),
            ("mm.2.weight", (cfg.proj_dim, cfg.proj_dim), F16, cfg.proj_dim), ("mm.2.bias", (cfg.proj_dim,), F32, 1)]
    if cfg.pinpoints:      # LLaVA-1.6 files carry the row separator of the original model; llama.cpp's layout ("without newline tokens") never reads it
        out += [("model.image_newline", (cfg.proj_dim,), F32, 1)]
    return out


def write_synthetic_clip(path: str, cfg: ClipConfig | str, seed: int = 0xC11F) -> int:
    """A random-weight mmproj file of the given geometry (same container, keys and tensor names as a converted LLaVA-1.5 projector)."""
    if isinstance(cfg, str):
        cfg = CLIP_CONFIGS[cfg]
    w = GGUFWriter()
    w.add("general.architecture", "str", "clip")
    w.add("general.name", "str", cfg.name)
    w.add("general.file_type", "u32", 1)
    w.add("clip.has_text_encoder", "bool", False)
    w.add("clip.has_vision_encoder", "bool", True)
    w.add("clip.has_llava_projector", "bool", True)
    w.add("clip.projector_type", "str", "mlp")
    w.add("clip.use_gelu", "bool", cfg.use_gelu)
    w.add("clip.vision.image_size", "u32", cfg.image_size)
    w.add("clip.vision.patch_size", "u32", cfg.patch_size)
    w.add("clip.vision.embedding_length", "u32", cfg.n_embd)
    w.add("clip.vision.feed_forward_length", "u32", cfg.n_ff)
    w.add("clip.vision.projection_dim", "u32", cfg.proj_dim)
    w.add("clip.vision.attention.head_count", "u32", cfg.n_head)
    w.add("clip.vision.attention.layer_norm_epsilon", "f32", cfg.eps)
    w.add("clip.vision.block_count", "u32", cfg.n_layer)
    w.add_array("clip.vision.image_mean", "f32", list(cfg.mean))
    w.add_array("clip.vision.image_std", "f32", list(cfg.std))
    if cfg.pinpoints:
        w.add_array("clip.vision.image_grid_pinpoints", "i32", list(cfg.pinpoints))
    if cfg.merge_type:
        w.add("clip.vision.mm_patch_merge_type", "str", cfg.merge_type)
    for idx, (name, ne, t, fan_in) in enumerate(clip_tensors(cfg)):
        n = int(np.prod(ne))

        def gen(idx=idx, name=name, ne=ne, t=t, fan_in=fan_in, n=n):
            rng = np.random.default_rng([seed, idx])
            if name.endswith(".bias"):
                return (rng.standard_normal(n) * 0.1).astype("<f4").view(np.uint8)
            if name in ("v.class_embd",):
                return (rng.standard_normal(n) * 0.5).astype("<f4").view(np.uint8)
            if len(ne) == 1:      # LayerNorm weights
                return rng.uniform(0.9, 1.1, size=n).astype("<f4").view(np.uint8)
            if name == "v.position_embd.weight":
                return (rng.standard_normal(n) * 0.3).astype("<f2").view(np.uint8)
            return (rng.standard_normal(n) * (1.4 / np.sqrt(fan_in))).astype("<f2").view(np.uint8)

        w.add_tensor(name, ne, t, gen)
    return w.write(path)
