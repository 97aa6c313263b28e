"""The exchange step of the row split on CPU, two ranks over gloo: the host transport's callback (binding.gloo_exchange,
what mi355_tp_set_host_exchange is given by tests/test_gpu_tp.py) must implement op 0 = in-place sum over ranks and
op 1 = all-gather with the caller's part already in place — the same contract the RCCL calls in host/tp_comm.cc fulfil.
Also: without a GPU the row-split entry points fail loudly instead of doing anything on the host."""
import ctypes as C
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fn = pkg.binding.gloo_exchange()
    n = 1000
    # op 0: in-place sum
    a = (np.arange(n, dtype=np.float32) + 1.0) * (rank + 1)
    assert fn(None, a.ctypes.data_as(C.POINTER(C.c_float)), n, 0) == 0
    want = (np.arange(n, dtype=np.float32) + 1.0) * sum(r + 1 for r in range(world))
    assert np.array_equal(a, want), (rank, a[:4], want[:4])
    # op 1: all-gather, own part in place, the rest garbage
    g = np.full(n * world, -1.0, np.float32)
    g[rank * n:(rank + 1) * n] = 100.0 * rank + np.arange(n, dtype=np.float32) / 1024
    assert fn(None, g.ctypes.data_as(C.POINTER(C.c_float)), n, 1) == 0
    for r in range(world):
        assert np.array_equal(g[r * n:(r + 1) * n], 100.0 * r + np.arange(n, dtype=np.float32) / 1024), (rank, r)
    dist.barrier()
    dist.destroy_process_group()
    print("ok", rank)
""") % ROOT


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_host_exchange_contract_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f"ok {r}" in outs[r], outs[r][-2000:]


def test_row_split_entry_points_need_a_device(pkg):
    import torch
    if torch.cuda.is_available():
        return
    lib = pkg.load_library()
    buf = (C.c_uint8 * 128)()
    assert lib.mi355_tp_unique_id(buf, 128) < 0
    assert lib.mi355_tp_init(0, 0, 1, buf, 128) != 0
    assert lib.mi355_tp_size() == 1 and lib.mi355_tp_rank() == 0
