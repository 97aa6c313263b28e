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
