import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_pkg(); pkg.Backend()
import numpy as np
path = "/tmp/tp-time-tiny0.gguf"
pkg.gguf_synth.write_synthetic_llama(path, "tiny-e2048", "q4_k_m", seed=3)
m = pkg.Model(path); c = pkg.Context(m, n_ctx=256, type_k=8, type_v=8); c.decode(np.arange(8), np.arange(8)); c.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "close":
    c.close(); m.close()
t = time.time()
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_rccl_init.py")], capture_output=True, text=True)
print(r.stdout[-600:]); print("child took", round(time.time() - t, 1), "s")
