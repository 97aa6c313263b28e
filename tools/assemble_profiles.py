#!/usr/bin/env python3
"""Turns what tools/run_profiles.sh left under gpurun_out/r3prof/ into the committed summaries under profiles/ (round 3):
kernel stats with their header, the decode traffic figure tied to the hash of the mat-vec sources it was measured on, the prompt-kernel counters, the
power / clock trace.  usage: tools/assemble_profiles.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r3prof")
P = os.path.join(ROOT, "profiles")
sha = open(f"{O}/kernel_sources_sha256.txt").read().strip()
fs = json.load(open(f"{O}/r3_pmc_fetch_size_by_kernel.json"))


def find(sub):
    for k, v in fs.items():
        if sub in k:
            return v
    raise KeyError(sub)


k21, k72, k20 = find("mmvq_stream_kernel<2, 1"), find("mmvq_stream_kernel<7, 2"), find("mmvq_stream_kernel<2, 0")
tot_l = k21["launches"] + k72["launches"] + k20["launches"]
b = lambda v: int(round(v["fetch_size_sum"] * 1024 * 2))   # noqa: E731
per_tok = int(round((b(k21) + b(k72) + b(k20)) * 129 / tot_l))
alg = 4616331264
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/decode_loop.py 16 (MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1; tools/run_profiles.sh step 2), "
              "Llama-3-8B Q4_K_M synthetic, prompt 512; round 3",
    "correction": "FETCH_SIZE is reported in KiB and tallies 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM; same for global_load and global_load_lds): "
                  "bytes = FETCH_SIZE * 1024 * 2",
    "kernel_sources_sha256": sha,
    "kernel_sources": ["mmvq_stream.hip", "mmvq_stream_dev.h", "mmvq_fast.hip", "mmvq_fast_dev.h", "mmvq.hip", "quant_dev.h", "dev_common.h"],
    "decode_steps": 16,
    "kernels": [
        {"kernel": "mmvq_stream_kernel<2,1> (qkv, gate+up with fused RMSNorm prologue, lm-head)", "launches": k21["launches"], "hbm_read_bytes": b(k21)},
        {"kernel": "mmvq_stream_kernel<7,2> (down, quantise prologue)", "launches": k72["launches"], "hbm_read_bytes": b(k72)},
        {"kernel": "mmvq_stream_kernel<2,0> (attn_output)", "launches": k20["launches"], "hbm_read_bytes": b(k20)},
    ],
    "matvec_hbm_read_bytes_per_token": per_tok,
    "algorithmic_weight_bytes_per_token": alg,
    "ratio": round(per_tok / alg, 4),
    "note": f"{tot_l} traced launches of the three kernels = 16 steps x 129 launches + the prompt's lm-head; per token = total x 129 / {tot_l}",
}
json.dump(out, open(f"{P}/r3_pmc_decode_traffic.json", "w"), indent=1)
print("traffic per token", per_tok, "ratio", out["ratio"], "sha", sha[:12])
for n in ("r3_pmc_fetch_size_by_kernel.json", "r3_pmc_prefill_mfma.json"):
    shutil.copy(f"{O}/{n}", f"{P}/{n}")
shutil.copy(f"{O}/bench_under_rocprof.json", f"{P}/r3_bench_under_rocprof.json")
old = open(f"{P}/r3_rocprof_kernel_stats.txt").read()
hdr = "".join(line + "\n" for line in old.splitlines() if line.startswith("#"))
open(f"{P}/r3_rocprof_kernel_stats.txt", "w").write(hdr + open(f"{O}/r3_rocprof_kernel_stats.txt").read())
old = open(f"{P}/r3_rocm_smi_prefill_trace.txt").read()
hdr = "".join(line + "\n" for line in old.splitlines()[:3])
loop = open(f"{O}/exp_p2_loop.txt").read()
open(f"{P}/r3_rocm_smi_prefill_trace.txt", "w").write(hdr + "".join("#   " + line + "\n" for line in loop.strip().splitlines()) + open(f"{O}/r3_rocm_smi_prefill_trace.txt").read())
