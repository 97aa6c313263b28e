"""The outer drop-in boundary: integration/mi355_engine_adapter.cc must implement EVERY pure virtual of the reference's EngineI
(base/cortex-common/enginei.h:31-73) and export get_engine (src/llama_engine.cc:1300-1304).  Compile-only: the reference's own enginei.h (read from
/root/reference, which exists in the build container only) + stub json / trantor headers under tests/stubs/.  Pins the SHAPE of the boundary, no numbers."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BASE = "/root/reference/base"
LIB_DIR = os.path.join(ROOT, "cortex.llamacpp_amd", "lib")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_BASE, "cortex-common", "enginei.h")), reason="reference interface header not present on this box")
def test_adapter_is_concrete_and_get_engine_links(tmp_path):
    so = tmp_path / "libengine.so"
    cmd = ["g++", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF_BASE, "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "integration", "mi355_engine_adapter.cc"), "-L", LIB_DIR, "-lmi355_llama", "-Wl,-rpath," + LIB_DIR, "-Wl,--no-undefined", "-o", str(so)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    # (`new Mi355Engine()` inside get_engine is ill-formed if one pure virtual is left unimplemented; -Wl,--no-undefined: every mi355_engine_* it calls exists)
    assert r.returncode == 0, r.stderr
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout
    assert " T get_engine" in syms
    und = subprocess.run(["nm", "-D", "--undefined-only", str(so)], capture_output=True, text=True).stdout
    for fn in ("mi355_engine_load", "mi355_engine_unload", "mi355_engine_set_file_logger", "mi355_engine_set_log_level", "mi355_engine_create"):
        assert fn in und
