"""CPU tests of the host side of the engine-formed row split (cortex.llamacpp_amd/host/shm_exchange.h): the shared-memory exchange between forked processes -
sums in rank order, gathers, messages longer than a piece, a rank that exits, a rank that never arrives (world sizes 2, 4 and 8)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shm") / "shm_exchange_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-pthread", os.path.join(ROOT, "tests", "host", "shm_exchange_test.cc"), "-o", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_shared_memory_exchange_between_processes(exe, ranks):
    r = subprocess.run([exe, str(ranks)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "all checks passed" in r.stdout
