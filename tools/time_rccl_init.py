#!/usr/bin/env python3
"""Where the time of forming a one-rank RCCL group goes (library load / communicator / first collective)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t0 = time.time()
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # noqa: F401  (its bundled librccl is found first by soname)
    print(f"import torch {time.time() - t0:.1f}s", flush=True)
import __graft_entry__ as ge
pkg = ge.load_pkg(); pkg.Backend()
lib = pkg.load_library()
t = time.time(); buf = (C.c_uint8 * 128)(); lib.mi355_tp_unique_id(buf, 128); print(f"unique_id (dlopen) {time.time() - t:.1f}s", flush=True)
t = time.time(); rc = lib.mi355_tp_init(0, 0, 1, buf, 128); print(f"tp_init rc={rc} {time.time() - t:.1f}s", flush=True)
import numpy as np
path = "/tmp/tp-time-tiny.gguf"
pkg.gguf_synth.write_synthetic_llama(path, "tiny-e2048", "q4_k_m", seed=3)
t = time.time(); m = pkg.Model(path, tp_rank=0, tp_size=1); c = pkg.Context(m, n_ctx=256, type_k=8, type_v=8)
c.decode(np.arange(8), np.arange(8)); c.synchronize(); print(f"first batch {time.time() - t:.1f}s", flush=True)
t = time.time(); c.decode([1], [8]); c.synchronize(); print(f"first graph step {time.time() - t:.1f}s", flush=True)
t = time.time(); c.decode([1], [9]); c.synchronize(); print(f"second graph step {time.time() - t:.3f}s", flush=True)
c.close(); m.close(); lib.mi355_tp_shutdown()
