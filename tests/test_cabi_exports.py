"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/mi355_llama.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mi355_llama.h")).read()
    return sorted(set(re.findall(r"MI355_API[^;(]*?\b(mi355_\w+)\s*\(", src)))


def test_header_declares_what_binding_binds(pkg):
    decl = declared_symbols()
    assert len(decl) > 40
    assert sorted(pkg.binding.SYMBOLS) == decl


def test_library_builds_loads_and_exports_every_symbol(pkg):
    import __graft_entry__ as ge
    ge.build()
    lib = pkg.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_no_cpu_fallback(pkg):
    lib = pkg.load_library()
    if lib.mi355_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.MI355Error):
        pkg.Backend()
    assert lib.mi355_backend_init() == -100
    with pytest.raises(pkg.MI355Error):
        pkg.Model("/nonexistent.gguf")
    import numpy as np
    x = np.zeros(256, np.float32)
    out = np.zeros(292, np.uint8)
    rc = lib.mi355_op_quantize_act(15, x.ctypes.data, 256, 1, out.ctypes.data)
    assert rc == -100


def test_product_does_not_touch_oracle():
    """The product path (package + C-ABI sources) must never import / link / call the oracle."""
    pkg_dir = os.path.join(ROOT, "cortex.llamacpp_amd")
    for base, _, files in os.walk(pkg_dir):
        if os.sep + "build" in base or os.sep + "lib" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in txt.lower() or f == "build.py", os.path.join(base, f)


def test_fused_attention_block_plan_by_geometry(pkg):
    """Round 6 host logic (no GPU): which layers run their Q | K | V inside the attention + attn_output launch (csrc/attn_out.hip QF).  The plan is a
    question of LDS: the workgroup's rows of attn_q | attn_k | attn_v and of attn_output, the activation planes and the attention's scratch in <= 160 KB.
    Llama-3-8B's geometry fits for Q4_K_M, Q5_K_M and an all-Q6_K file; an 8-expert file's Q8_0 attn_k / attn_v fit too; Llama-2-7B's 12288 Q | K | V rows (140 KB per
    workgroup), the 70B's 8192-wide rows, head size 64, a Q8_0 attn_q / attn_output and n_embd != n_head * head_dim do not and keep the two launches."""
    import ctypes as C
    lib = pkg.load_library()
    Q40, Q80, Q4K, Q5K, Q6K = 2, 8, 12, 13, 14

    def plan(tq, tk, tv, to, E, H, G, D=128, kv=8, n_kv=576):
        slots = C.c_int32(0)
        lds = int(lib.mi355_debug_qkv_attn_plan(tq, tk, tv, to, E, H, G, D, kv, n_kv, C.byref(slots)))
        return lds, int(slots.value)

    lds, slots = plan(Q4K, Q4K, Q6K, Q4K, 4096, 32, 8)                 # Llama-3-8B Q4_K_M ("more bits" layer: attn_v in Q6_K)
    assert 100 * 1024 < lds <= 160 * 1024 and 20 <= slots <= 30, (lds, slots)
    lds4k, _ = plan(Q4K, Q4K, Q4K, Q4K, 4096, 32, 8, n_kv=4096)        # the context filled: 128-cell items, one per workgroup
    assert 0 < lds4k <= 160 * 1024
    assert plan(Q5K, Q5K, Q6K, Q5K, 4096, 32, 8)[0] > lds              # Q5_K_M: more bytes per row, still one launch
    assert 0 < plan(Q6K, Q6K, Q6K, Q6K, 4096, 32, 8)[0] <= 160 * 1024
    assert plan(Q5K, Q80, Q80, Q5K, 4096, 32, 8)[1] > 30               # Mixtral's attention tensors: more slots than a loader wave may have in flight - capped, not refused
    assert plan(Q5K, Q5K, Q6K, Q5K, 4096, 32, 32, kv=1)[0] == 0        # Llama-2-7B: 140 KB of Q | K | V rows per workgroup
    assert plan(Q4K, Q4K, Q6K, Q4K, 8192, 64, 8)[0] == 0               # Llama-3-70B on one GPU
    assert plan(Q4K, Q4K, Q4K, Q4K, 2048, 32, 4, D=64)[0] == 0         # TinyLlama: head size 64
    assert plan(Q80, Q80, Q80, Q80, 4096, 32, 8)[0] == 0               # a Q8_0 file: attn_output has no form in this launch
    assert plan(Q40, Q4K, Q4K, Q4K, 4096, 32, 8)[0] == 0
    assert plan(Q4K, Q4K, Q4K, Q4K, 3584, 32, 8)[0] == 0               # n_embd != n_head * head_dim
    assert plan(Q4K, Q4K, Q4K, Q4K, 1024, 8, 2)[0] > 0                 # the tiny test models' geometry (tests/test_gpu_model.py)
