#!/usr/bin/env python3
"""Turns what tools/run_profiles_r6.sh left under gpurun_out/r6prof/ into the committed summaries under profiles/ (round 6).  New this round: the weight-stream
kernel carries one NAME per role of a decode step (mmvq_stream_qkv / _gate_up / _ffn_down / _head, csrc/mmvq_stream.hip), so the traced table has one row per
role and the achieved bytes per second of every role are worked out here from the trace alone (r6_rocprof_decode_roofline.json: `roles`), next to the
attention + attn_output launch.  usage: tools/assemble_profiles_r6.py"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

O = os.path.join(ROOT, "gpurun_out", "r6prof")
P = os.path.join(ROOT, "profiles")
sha = open(f"{O}/kernel_sources_sha256.txt").read().strip()
assert sha == bench.kernel_sources_sha256(ROOT), "the kernel sources changed since the profiles were taken"
N_LAYER = 32
E, FF, V, GD = 4096, 14336, 128256, 1024
Q4K, Q6K = 144 / 256, 210 / 256                      # bytes per weight
# Llama-3-8B Q4_K_M (gguf_synth's type mix = llama.cpp's): attn_v and ffn_down are Q6_K in the "more bits" layers (use_more_bits: the first and last eighth and
# every third layer in between), Q4_K elsewhere; the head is Q6_K


def more_bits(i, n):
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


n_more = sum(more_bits(i, N_LAYER) for i in range(N_LAYER))
ROLE_BYTES = {                                        # algorithmic weight bytes per token, by role
    "qkv": N_LAYER * (E * E + E * GD) * Q4K + (N_LAYER - n_more) * E * GD * Q4K + n_more * E * GD * Q6K,
    "gate_up": N_LAYER * 2 * E * FF * Q4K,
    "ffn_down": (N_LAYER - n_more) * E * FF * Q4K + n_more * E * FF * Q6K,
    "head": V * E * Q6K,
}
WO_BYTES = N_LAYER * E * E * Q4K
ROLE_LAUNCHES = {"qkv": N_LAYER, "gate_up": N_LAYER, "ffn_down": N_LAYER, "head": 1}


def stats_rows(path):
    rows = []
    for line in open(path).read().splitlines()[1:]:
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)$", line)
        if m:
            rows.append((m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    return rows


def with_header(src, dst, header):
    open(dst, "w").write("".join("# " + h + "\n" for h in header) + open(src).read())


with_header(f"{O}/r6_rocprof_kernel_stats.txt", f"{P}/r6_rocprof_kernel_stats.txt", [
    "round 6: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 128 --warmup 16 --no-cpu-baseline (tools/run_profiles_r6.sh step 1)",
    "MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1: eager launches (rocprofv3 7.2 crashes while tracing hipGraph replays); Llama-3-8B Q4_K_M synthetic, cache q8_0,",
    "512-token prompt, decode at pos 512..., then the 3968-token fill and the steps at pos ~3976 (long_context), the device-greedy loop and the sweep.",
    "qkv_attn_out_kernel = Q|K|V + decode attention + attn_output mat-vec in one launch (csrc/attn_out.hip, round 6); mmvq_stream_gate_up / _ffn_down / _head = the weight-stream kernel by role",
    f"kernel_sources_sha256 {sha}",
])
with_header(f"{O}/r6_rocprof_prefill_kernel_stats.txt", f"{P}/r6_rocprof_prefill_kernel_stats.txt", [
    "round 6: rocprofv3 --kernel-trace --stats -- python3 tools/decode_loop.py 1 512 (tools/run_profiles_r6.sh step 3): model load (expand / repack kernels), ONE 512-token",
    "prompt (32 layers: calls / 32 = launches per layer) and one single-token step; eager launches",
    f"kernel_sources_sha256 {sha}",
])
with_header(f"{O}/r6_rocprof_decode_kernel_stats.txt", f"{P}/r6_rocprof_decode_kernel_stats.txt", [
    "round 6: rocprofv3 --kernel-trace --stats -- python3 tools/decode_loop.py 64 (tools/run_profiles_r6.sh step 2): a 512-token prompt, then 64 single-token steps at pos 512..575; eager launches",
    "one row per role of the weight-stream kernel: mmvq_stream_gate_up (SwiGLU pair), _ffn_down (quantising prologue + residual), _head (output head); Q|K|V run inside",
    "qkv_attn_out_kernel since round 6 (one launch per layer: RMSNorm -> Q8_K, Q|K|V, rope, KV store, attention, merge, Q8_K, attn_output + residual)",
    f"kernel_sources_sha256 {sha}",
])

rows = stats_rows(f"{O}/r6_rocprof_decode_kernel_stats.txt")


def role_rows(tag):
    # (gate | up and the head also run as mmvq_stream_<role>_fast: the form that starts its stream from preloaded kernel arguments)
    # (ffn_down also runs as mmvq_stream_ffn_down_early: its activation requested from preloaded arguments)
    return [(n, c, t, a) for n, c, t, a in rows if f"mmvq_stream_{tag}<" in n or f"mmvq_stream_{tag}_fast<" in n or f"mmvq_stream_{tag}_early<" in n]


down = role_rows("ffn_down")
n_steps = sum(c for _, c, _, _ in down) // N_LAYER
roles, st_us_tok, st_bytes_tok = {}, 0.0, 0.0
for tag in ("qkv", "gate_up", "ffn_down", "head"):
    rr = role_rows(tag)
    calls, tot = sum(c for _, c, _, _ in rr), sum(t for _, _, t, _ in rr)
    if calls == 0:
        continue
    avg = tot / calls                                   # (the head row also holds the prompt's one head launch: the average is per launch either way)
    us_tok = avg * ROLE_LAUNCHES[tag]
    b_tok = ROLE_BYTES[tag]
    roles[tag] = {"launches_per_token": ROLE_LAUNCHES[tag], "avg_launch_us": round(avg, 3), "us_per_token": round(us_tok, 2),
                  "weight_bytes_per_token": int(b_tok), "weight_bytes_per_launch": int(b_tok / ROLE_LAUNCHES[tag]),
                  "GBps": round(b_tok / us_tok / 1e3, 1), "frac_of_8TBps": round(b_tok / us_tok / 1e3 / 8000.0, 4)}
    st_us_tok += us_tok
    st_bytes_tok += b_tok
ao = [(n, c, t, a) for n, c, t, a in rows if "attn_out_kernel" in n]
# round 6: where the step runs Q | K | V inside its attention + attn_output launch (qkv_attn_out_kernel, csrc/attn_out.hip QF) the trace has no mmvq_stream_qkv row and
# the attention launch reads attn_q / attn_k / attn_v as well
QF = any("qkv_attn_out_kernel" in n for n, _, _, _ in ao)
AO_BYTES = WO_BYTES + (ROLE_BYTES["qkv"] if QF else 0)
ao_us_tok = sum(t for _, _, t, _ in ao) / max(1, n_steps)
per_tok_launches = sum(ROLE_LAUNCHES[t] for t in roles)
roof = {
    "source": "profiles/r6_rocprof_decode_kernel_stats.txt (rocprofv3 --kernel-trace --stats, eager launches, 64 steps at pos 512..575)",
    "kernel_sources_sha256": sha,
    "decode_steps": n_steps,
    "roles": roles,
    "stream_launches_per_token": per_tok_launches,
    "stream_us_per_token": round(st_us_tok, 2),
    "stream_avg_launch_us": round(st_us_tok / per_tok_launches, 3),
    "stream_weight_bytes_per_token": int(st_bytes_tok),
    "stream_GBps": round(st_bytes_tok / st_us_tok / 1e3, 1),
    "frac_rocprof": round(st_bytes_tok / st_us_tok / 1e3 / 8000.0, 4),
    "attn_out_us_per_token": round(ao_us_tok, 2),
    "attn_out_avg_launch_us": round(ao_us_tok / N_LAYER, 3),
    "attn_out_kernel": "qkv_attn_out_kernel (RMSNorm -> Q8_K, Q | K | V, rope, KV store, attention, merge, Q8_K, attn_output + residual: one launch per layer)" if QF
                       else "attn_out_kernel (rope, KV store, attention, merge, Q8_K, attn_output + residual)",
    "attn_out_weight_bytes_per_token": int(AO_BYTES),
    "attn_out_frac_of_8TBps": round(AO_BYTES / ao_us_tok / 1e3 / 8000.0, 4),
    "all_matvec_GBps": round((st_bytes_tok + AO_BYTES) / (st_us_tok + ao_us_tok) / 1e3, 1),
    "frac_rocprof_with_attention_launch": round((st_bytes_tok + AO_BYTES) / (st_us_tok + ao_us_tok) / 1e3 / 8000.0, 4),
}
json.dump(roof, open(f"{P}/r6_rocprof_decode_roofline.json", "w"), indent=1)
print("traced stream", roof["stream_us_per_token"], "us/token ->", roof["stream_GBps"], "GB/s, frac", roof["frac_rocprof"])
for k, v in roles.items():
    print(f"  {k:<9} {v['avg_launch_us']:>7.2f} us/launch  {v['weight_bytes_per_launch'] / 1e6:>7.1f} MB  {v['GBps']:>7.1f} GB/s  frac {v['frac_of_8TBps']}")
print(f"  attn_out  {roof['attn_out_avg_launch_us']:>7.2f} us/launch")

# ---- PMC traffic
fs = json.load(open(f"{O}/r6_pmc_fetch_size_by_kernel.json"))
b = lambda v: int(round(v["fetch_size_sum"] * 1024 * 2))   # noqa: E731
ks = {k: v for k, v in fs.items() if "mmvq_stream_" in k}
ka = {k: v for k, v in fs.items() if "attn_out_kernel" in k}
steps16 = sum(v["launches"] for k, v in ks.items() if "mmvq_stream_ffn_down<" in k or "mmvq_stream_ffn_down_early<" in k) // N_LAYER
per_role = {}
per_tok = 0
for tag in ("qkv", "gate_up", "ffn_down", "head"):
    kk = {k: v for k, v in ks.items() if f"mmvq_stream_{tag}<" in k or f"mmvq_stream_{tag}_fast<" in k or f"mmvq_stream_{tag}_early<" in k}
    n_l = sum(v["launches"] for v in kk.values())
    if n_l == 0:
        continue
    per_launch = sum(b(v) for v in kk.values()) / n_l
    per_role[tag] = {"hbm_read_bytes_per_launch": int(per_launch), "algorithmic_bytes_per_launch": int(ROLE_BYTES[tag] / ROLE_LAUNCHES[tag]),
                     "ratio": round(per_launch / (ROLE_BYTES[tag] / ROLE_LAUNCHES[tag]), 4)}
    per_tok += per_launch * ROLE_LAUNCHES[tag]
ao_tok = int(round(sum(b(v) for v in ka.values()) / max(1, steps16)))
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/decode_loop.py 16 (MI355_NO_GRAPHS=1 MI355_PROFILER_SAFE=1; tools/run_profiles_r6.sh step 4), "
              "Llama-3-8B Q4_K_M synthetic, prompt 512; round 6",
    "correction": "FETCH_SIZE is reported in KiB and tallies 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM; same for global_load and global_load_lds): "
                  "bytes = FETCH_SIZE * 1024 * 2",
    "kernel_sources_sha256": sha,
    "kernel_sources": list(bench.KERNEL_SOURCES),
    "decode_steps": steps16,
    "kernels": [{"kernel": k[:100], "launches": v["launches"], "hbm_read_bytes": b(v)} for k, v in list(ks.items()) + list(ka.items())],
    "roles": per_role,
    "matvec_hbm_read_bytes_per_token": int(per_tok),
    "algorithmic_weight_bytes_per_token": int(st_bytes_tok),
    "ratio": round(per_tok / st_bytes_tok, 4),
    "attn_out_hbm_read_bytes_per_token": ao_tok,
    "attn_out_algorithmic_bytes_per_token": int(AO_BYTES),
    "note": "per role: traced HBM read bytes per launch against the tensor's bytes; attn_output's 302 MB per token (and, round 6, the 470 MB of attn_q / attn_k / attn_v) are "
            "read inside the attention launch (with the KV cells of the step) and listed separately",
}
json.dump(out, open(f"{P}/r6_pmc_decode_traffic.json", "w"), indent=1)
print("traffic per token", int(per_tok), "ratio", out["ratio"], "attn_out", ao_tok, "sha", sha[:12])
for n in ("r6_pmc_fetch_size_by_kernel.json", "r6_pmc_prefill_mfma.json"):
    if os.path.exists(f"{O}/{n}"):
        shutil.copy(f"{O}/{n}", f"{P}/{n}")
shutil.copy(f"{O}/bench_under_rocprof.json", f"{P}/r6_bench_under_rocprof.json")
shutil.copy(f"{O}/r6_bench.json", f"{P}/r6_bench.json")
