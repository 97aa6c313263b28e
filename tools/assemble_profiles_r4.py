#!/usr/bin/env python3
"""Turns what tools/run_profiles_r4.sh left under gpurun_out/r4prof/ into the committed summaries under profiles/ (round 4): kernel stats with a header,
the decode traffic figure and the traced per-token time of the weight-stream launches, both tied to the hash of the kernel sources they were measured
on, and the prompt-kernel counters.  usage: tools/assemble_profiles_r4.py"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

O = os.path.join(ROOT, "gpurun_out", "r4prof")
P = os.path.join(ROOT, "profiles")
sha = open(f"{O}/kernel_sources_sha256.txt").read().strip()
assert sha == bench.kernel_sources_sha256(ROOT), "the kernel sources changed since the profiles were taken"
N_LAYER, STEPS = 32, 64
ALG_TOKEN = 4616331264 + 0          # weight bytes of one token (bench.py config.weight_bytes_per_token minus the embedding row)
WO_BYTES = 32 * 4096 * 4096 * 144 // 256     # attn_output (Q4_K) of every layer: streamed inside attn_out_kernel since round 4


def stats_rows(path):
    rows = []
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    return rows


def with_header(src, dst, header):
    open(dst, "w").write("".join("# " + h + "\n" for h in header) + open(src).read())


with_header(f"{O}/r4_rocprof_kernel_stats.txt", f"{P}/r4_rocprof_kernel_stats.txt", [
    "round 4: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 128 --warmup 16 --no-cpu-baseline (tools/run_profiles_r4.sh step 1)",
    "MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1: eager launches (rocprofv3 7.2 crashes while tracing hipGraph replays); Llama-3-8B Q4_K_M synthetic, cache q8_0,",
    "512-token prompt, decode at pos 512..., then the 3968-token fill and the steps at pos ~3976 (long_context), the device-greedy loop and the sweep.",
    "attn_out_kernel = decode attention + attn_output mat-vec in one launch (csrc/attn_out.hip); mmvq_stream_kernel<2,1> = Q|K|V, gate|up, lm-head; <7,2> = ffn_down",
    f"kernel_sources_sha256 {sha}",
])
with_header(f"{O}/r4_rocprof_prefill_kernel_stats.txt", f"{P}/r4_rocprof_prefill_kernel_stats.txt", [
    "round 4: rocprofv3 --kernel-trace --stats -- python3 tools/decode_loop.py 1 512 (tools/run_profiles_r4.sh step 3): model load (expand / repack kernels), ONE 512-token",
    "prompt (32 layers: calls / 32 = launches per layer) and one single-token step; eager launches",
    f"kernel_sources_sha256 {sha}",
])
with_header(f"{O}/r4_rocprof_decode_kernel_stats.txt", f"{P}/r4_rocprof_decode_kernel_stats.txt", [
    "round 4: rocprofv3 --kernel-trace --stats -- python3 tools/decode_loop.py 64 (tools/run_profiles_r4.sh step 2): a 512-token prompt, then 64 single-token steps at pos 512..575; eager launches",
    f"kernel_sources_sha256 {sha}",
])

# ---- the weight-stream launches of a token from the traced durations (what roofline.frac_rocprof of the bench line is)
rows = stats_rows(f"{O}/r4_rocprof_decode_kernel_stats.txt")
st = [(n, c, t, a) for n, c, t, a in rows if "mmvq_stream_kernel" in n]
ao = [(n, c, t, a) for n, c, t, a in rows if "attn_out_kernel" in n]
n_steps = sum(c for n, c, t, a in st if "<7, 2" in n) // N_LAYER
n_st_launch = sum(c for _, c, _, _ in st)
st_us = sum(t for _, _, t, _ in st)
per_tok_launches = 3 * N_LAYER + 1
# (the prompt's lm-head launch rides in the <2,1> row: one launch of the n_st_launch; scale to the launches of whole steps)
st_us_tok = st_us * (per_tok_launches * n_steps) / n_st_launch / n_steps
st_bytes_tok = ALG_TOKEN - WO_BYTES
ao_us_tok = sum(t for _, _, t, _ in ao) / max(1, n_steps)
roof = {
    "source": "profiles/r4_rocprof_decode_kernel_stats.txt (rocprofv3 --kernel-trace --stats, eager launches, 64 steps at pos 512..575)",
    "kernel_sources_sha256": sha,
    "decode_steps": n_steps,
    "stream_launches_per_token": per_tok_launches,
    "stream_us_per_token": round(st_us_tok, 2),
    "stream_avg_launch_us": round(st_us_tok / per_tok_launches, 3),
    "stream_weight_bytes_per_token": st_bytes_tok,
    "stream_GBps": round(st_bytes_tok / st_us_tok / 1e3, 1),
    "frac_rocprof": round(st_bytes_tok / st_us_tok / 1e3 / 8000.0, 4),
    "attn_out_us_per_token": round(ao_us_tok, 2),
    "attn_out_weight_bytes_per_token": WO_BYTES,
    "all_matvec_GBps": round(ALG_TOKEN / (st_us_tok + ao_us_tok) / 1e3, 1),
    "frac_rocprof_with_attention_launch": round(ALG_TOKEN / (st_us_tok + ao_us_tok) / 1e3 / 8000.0, 4),
}
json.dump(roof, open(f"{P}/r4_rocprof_decode_roofline.json", "w"), indent=1)
print("traced stream", roof["stream_us_per_token"], "us/token ->", roof["stream_GBps"], "GB/s, frac", roof["frac_rocprof"])

# ---- PMC traffic
fs = json.load(open(f"{O}/r4_pmc_fetch_size_by_kernel.json"))
b = lambda v: int(round(v["fetch_size_sum"] * 1024 * 2))   # noqa: E731
ks = {k: v for k, v in fs.items() if "mmvq_stream_kernel" in k}
ka = {k: v for k, v in fs.items() if "attn_out_kernel" in k}
tot_l = sum(v["launches"] for v in ks.values())
steps16 = sum(v["launches"] for k, v in ks.items() if "<7, 2" in k) // N_LAYER
per_tok = int(round(sum(b(v) for v in ks.values()) * per_tok_launches / tot_l))
ao_tok = int(round(sum(b(v) for v in ka.values()) / max(1, steps16)))
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/decode_loop.py 16 (MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1; tools/run_profiles_r4.sh step 4), "
              "Llama-3-8B Q4_K_M synthetic, prompt 512; round 4",
    "correction": "FETCH_SIZE is reported in KiB and tallies 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM; same for global_load and global_load_lds): "
                  "bytes = FETCH_SIZE * 1024 * 2",
    "kernel_sources_sha256": sha,
    "kernel_sources": list(bench.KERNEL_SOURCES),
    "decode_steps": steps16,
    "kernels": [{"kernel": k[:100], "launches": v["launches"], "hbm_read_bytes": b(v)} for k, v in list(ks.items()) + list(ka.items())],
    "matvec_hbm_read_bytes_per_token": per_tok,
    "algorithmic_weight_bytes_per_token": st_bytes_tok,
    "ratio": round(per_tok / st_bytes_tok, 4),
    "attn_out_hbm_read_bytes_per_token": ao_tok,
    "attn_out_algorithmic_bytes_per_token": WO_BYTES,
    "note": f"{tot_l} traced launches of the weight-stream kernel = {steps16} steps x {per_tok_launches} launches + the prompt's lm-head; per token = total x {per_tok_launches} / {tot_l}.  "
            "attn_output's 302 MB per token are read inside attn_out_kernel (with the KV cells of the step) and listed separately",
}
json.dump(out, open(f"{P}/r4_pmc_decode_traffic.json", "w"), indent=1)
print("traffic per token", per_tok, "ratio", out["ratio"], "attn_out", ao_tok, "sha", sha[:12])
for n in ("r4_pmc_fetch_size_by_kernel.json", "r4_pmc_prefill_mfma.json"):
    if os.path.exists(f"{O}/{n}"):
        shutil.copy(f"{O}/{n}", f"{P}/{n}")
shutil.copy(f"{O}/bench_under_rocprof.json", f"{P}/r4_bench_under_rocprof.json")
shutil.copy(f"{O}/r4_bench.json", f"{P}/r4_bench.json")
