import os, sys, tempfile, pathlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import test_gpu_tp as T
pkg = ge.load_pkg()
pkg.Backend()
tmp = pathlib.Path(tempfile.mkdtemp())
cfg, ftype, kv, npmt, world = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
transport = sys.argv[6] if len(sys.argv) > 6 else "host"
path, plan_path, ref = T.make_plan(pkg, tmp, cfg, ftype, kv, npmt, transport)
os.environ["MI355_TP_DUMP_ALL"] = "1"
got = T.run_ranks(world, plan_path, str(tmp / "out.npz"))
one, taps1, heads = T.unsplit_logits(pkg, plan_path)
lg = got["logits"]
print("heads", heads, int(got["n_head"]), int(got["n_head_kv"]), int(got["bytes_per_token"]))
print("vs ref", [round(T.rel_err(a, b), 6) for a, b in zip(lg, ref)])
print("vs one", [round(T.rel_err(a, b), 6) for a, b in zip(lg, one)])
print("one vs ref", [round(T.rel_err(a, b), 6) for a, b in zip(one, ref)])
print("lg[0][:8]", lg[0][:8], "\nref[0][:8]", ref[0][:8], "\nlg[0][-8:]", lg[0][-8:], "\nref[0][-8:]", ref[0][-8:])
print("nan", np.isnan(lg).sum(), "absmax", np.abs(lg).max(), np.abs(ref).max())

for r in range(1, world):
    o = np.load(str(tmp / "out.npz") + f".rank{r}.npz")["logits"]
    print("rank", r, "vs ref", [round(T.rel_err(a, b), 6) for a, b in zip(o, ref)])
    V = o.shape[1]; VL = V // world
    print("  own slice nonzero", [int((row[r*VL:(r+1)*VL] != 0).sum()) for row in o], "other", [int((row[:VL] != 0).sum()) for row in o])
    print("  r row0 [0:4]", o[0][:4], "[VL:VL+4]", o[0][VL:VL+4]); print("  ref    [0:4]", ref[0][:4], "[VL:VL+4]", ref[0][VL:VL+4])
    print("  r row1 [0:4]", o[1][:4], "[VL:VL+4]", o[1][VL:VL+4]); print("  ref    [0:4]", ref[1][:4], "[VL:VL+4]", ref[1][VL:VL+4])

tapsP = got["taps"]
for il in range(tapsP.shape[0]):
    a, b = tapsP[il], taps1[il]
    print("layer", il, "per-token err", np.round(np.abs(a - b).max(axis=1) / max(1.0, np.abs(b).max()), 6))
