#!/usr/bin/env python3
"""Matrix-pipe utilisation of the prompt kernels from two rocprofv3 --pmc databases (rocpd sqlite):
  pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
  pass 2: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 GRBM_GUI_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES = 32 per v_mfma_i32_32x32x32_i8, summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs:
available SIMD-cycles = GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs.
usage: tools/pmc_prefill.py <pass1.db> <pass2.db> <out.json> [note]"""
import json
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    agg = {}
    for k, c, v in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if not any(s in k for s in ("mmq_planes", "flash_attn_prefill", "mmq_kernel", "mmq_ksplit")):
            continue
        i = min(x for x in (k.find("mmq_"), k.find("flash_attn")) if x >= 0)
        j = k.find("(", i)
        a = agg.setdefault(k[i:j if j > 0 else None], {})
        a[c] = a.get(c, 0.0) + float(v)
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
out = {"source": "two rocprofv3 --pmc passes with --kernel-trace over python3 tools/decode_loop.py 1 512 (MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1)"
                 + (": " + sys.argv[4] if len(sys.argv) > 4 else ""),
       "units": __doc__.split("\n")[3:6], "kernels": {}}
for k in sorted(set(a) | set(b)):
    c = dict(a.get(k, {})); c.update(b.get(k, {}))
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        c["mfma_util"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 256 * 4), 4)
    if c.get("SQ_INSTS_MFMA", 0) > 0 and "SQ_INSTS_VALU" in c:
        c["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 2)
    if c.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_INST_ANY" in c:
        c["wait_fraction_of_wave_cycles"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    out["kernels"][k] = c
    print(k, {x: c[x] for x in ("mfma_util", "valu_per_mfma", "wait_fraction_of_wave_cycles") if x in c})
json.dump(out, open(sys.argv[3], "w"), indent=1)
