#!/usr/bin/env python3
"""bench.py — decode (and prefill) throughput of the MI355X-native GGUF backend on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): Llama-3-8B-Instruct Q4_K_M, flash_attn,
KV cache q8_0, n_ctx 4096, synthetic GGUF (exact shapes / type mix, seeded random valid blocks), synthetic prompt
(seeded uniform token ids).  One "step" = one mi355_decode (llama_decode) call on a single new token with its logits
row made host-visible, i.e. one pass of the hot path; the prompt is prefilled (and timed separately) first.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank holds a full replica and decodes its own
sequence; no data-path collective (weak scaling, "replicas").  value = N*K tokens / max-over-ranks time.
The same launch then measures the ROW SPLIT of the same model over the N GPUs (one sequence, every rank
holds 1/N of each projection, two RCCL all-reduces per layer inside the decode graph) and reports it next to the
replicas number as `row_split` (single-stream tok/s: what one request sees; strong scaling).  That section is fenced
by a watchdog so that a communication problem cannot take the headline line with it.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_UBATCH = int(os.environ.get("MI355_BENCH_UBATCH", "2048"))   # the reference default: n_ubatch = n_batch = 2048 (llama_engine.cc:617-620)
MFMA_I8_PEAK_TOPS = 5000.0      # dense int8 / fp8 matrix-core peak of MI355X (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s measured achievable


KERNEL_SOURCES = ("mmvq_stream.hip", "mmvq_stream_dev.h", "mmvq_fast.hip", "mmvq_fast_dev.h", "mmvq.hip", "quant_dev.h", "dev_common.h",
                  "attn_out.hip", "attn_decode_dev.h")


def kernel_sources_sha256(root: str) -> str:
    """Identity of the decode mat-vec kernel sources (what a committed PMC traffic figure is valid for)."""
    import hashlib
    h = hashlib.sha256()
    for n in KERNEL_SOURCES:
        with open(os.path.join(root, "cortex.llamacpp_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def aggregate(n_gpus: int, steps: int, dt_max: float) -> dict:
    """Whole-job numbers from the slowest rank's wall time: every rank decoded `steps` tokens of its own sequence."""
    return {"value": round(n_gpus * steps / dt_max, 2), "ms_per_step": round(dt_max / steps * 1e3, 4)}


def row_split_section(args, pkg, path, KV, rank, local_rank, world, dist, torch, partial=None) -> dict:
    """Row split of the bench model over the ranks of this launch: the same prompt on every rank (they execute one
    sequence together), prefill, warm-up, K timed single-token steps bracketed by barriers, MAX over ranks.
    Timed twice: every exchange through RCCL first (the result is stashed in `partial`, so the watchdog can still report it), then with the one-shot
    peer-to-peer all-reduce kernel for the decode-sized exchanges unless --no-p2p; the better one is `decode_tok_s`."""
    partial = {} if partial is None else partial
    pkg.binding.tp_init(rank, world, device=local_rank, transport="rccl")
    model = pkg.Model(path, main_gpu=local_rank, tp_rank=rank, tp_size=world)
    prompt = np.random.default_rng(1234).integers(0, model.n_vocab, args.prompt)
    holder = {}

    def new_context():
        if "ctx" in holder:
            holder["ctx"].close()
        holder["ctx"] = pkg.Context(model, n_ctx=args.ctx, n_batch=2048, n_ubatch=N_UBATCH, type_k=KV, type_v=KV, flash_attn=True, use_graphs=True)

    def prefill():
        ctx = holder["ctx"]
        ctx.kv_clear()
        t = time.perf_counter()
        for i0 in range(0, args.prompt, 2048):
            chunk = prompt[i0:i0 + 2048]
            assert ctx.decode(chunk, np.arange(i0, i0 + chunk.size)) == 0
        tok = ctx.argmax()                                 # first sampled token: end of the reference's prompt time
        return time.perf_counter() - t, tok

    def sync_all():
        holder["ctx"].synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    def timed():
        """prefill, warm-up, K timed single-token steps; (seconds, prefill seconds, last token, first token) with the MAX over ranks"""
        ctx = holder["ctx"]
        t_pf, tok = prefill()
        tok_first = tok
        pos = args.prompt
        for _ in range(args.warmup):
            assert ctx.decode([tok], [pos]) == 0
            ctx.logits_ready(); tok = ctx.argmax(); pos += 1
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            assert ctx.decode([tok], [pos]) == 0
            ctx.logits_ready(); tok = ctx.argmax(); pos += 1
        ctx.synchronize()
        dt_ = time.perf_counter() - t0
        sync_all()
        tt = torch.tensor([dt_, t_pf], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt[0].item()), float(tt[1].item()), tok, tok_first

    def agree(tok):
        toks = torch.tensor([tok], dtype=torch.int64, device=f"cuda:{local_rank}")
        lo, hi = toks.clone(), toks.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(int(lo.item()) == int(hi.item()))

    # ---- every exchange through RCCL
    new_context()
    prefill()
    sync_all()
    dt_rccl, t_prefill, tok_rccl, tok_first_rccl = timed()
    res = {"parallelism": f"row split over {world} GPUs (attn_output / ffn_down partial sums all-reduced twice per layer through RCCL; logits gathered)",
           "scaling": "strong", "ranks": int(pkg.binding.load_library().mi355_tp_size()), "decode_tok_s": round(args.steps / dt_rccl, 2),
           "ms_per_step": round(dt_rccl / args.steps * 1e3, 4), "decode_tok_s_p2p": None, "p2p_exchanges": 0,
           "decode_tok_s_rccl": round(args.steps / dt_rccl, 2), "p2p_and_rccl_agree_on_last_token": None,
           "prefill_tok_s": round(args.prompt / t_prefill, 1), "weight_bytes_per_token_per_gpu": int(model.bytes_per_token),
           "ranks_agree_on_last_token": agree(tok_rccl)}
    partial.update(res)
    # ---- the decode-sized exchanges through the one-shot peer-to-peer kernel (IPC-mapped buffers).  Enabling is collective; every rank must succeed
    if world > 1 and not args.no_p2p:
        ok, why = 1, ""
        try:
            pkg.binding.tp_p2p_enable(rank, world, local_rank, 16384, prompt_floats=model.n_embd * min(args.prompt, N_UBATCH))
            pkg.Backend().set_option("tp_p2p_prompt", 0)       # (its own phase below: this one keeps RCCL for the prompt batches)
        except Exception as e:  # noqa: BLE001 - reported in the record
            ok, why = 0, f"{type(e).__name__}: {e}"[:200]
        flag = torch.tensor([ok], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            partial["p2p_error"] = "the peer-to-peer phase did not finish"      # what the watchdog reports if this phase hangs
            new_context()                                  # new graphs: the exchange is part of them
            prefill()
            sync_all()
            dt_p2p, _, tok_p2p, _ = timed()
            used = int(pkg.binding.tp_p2p_exchanges())
            res["decode_tok_s_p2p"] = round(args.steps / dt_p2p, 2)
            res["p2p_exchanges"] = used
            res["p2p_and_rccl_agree_on_last_token"] = bool(tok_p2p == tok_rccl)
            res["ranks_agree_on_last_token"] = res["ranks_agree_on_last_token"] and agree(tok_p2p)
            if used > 0 and dt_p2p < dt_rccl:
                res["decode_tok_s"], res["ms_per_step"] = round(args.steps / dt_p2p, 2), round(dt_p2p / args.steps * 1e3, 4)
                res["parallelism"] = (f"row split over {world} GPUs (attn_output / ffn_down partial sums all-reduced twice per layer: one-shot peer-to-peer kernel "
                                      f"over IPC-mapped buffers for the decode steps, RCCL for prompt batches; logits gathered)")
            partial.pop("p2p_error", None)
            partial.update(res)
            # ---- the prompt batches' exchanges as ONE reduce-scatter + all-gather kernel over all links (host/tp_comm.cc p2p_rsag_kernel); prefill only
            # (this kernel has never run over real xGMI links: a failure here - a bounded wait that gives up on every rank - is recorded and must not
            # take the two measurements above with it)
            partial["rsag_error"] = "the reduce-scatter + all-gather phase did not finish"
            res["prefill_tok_s_rccl"] = res["prefill_tok_s"]
            try:
                pkg.Backend().set_option("tp_p2p_prompt", 1)
                new_context()
                prefill()
                sync_all()
                t_rs, tok_rs = prefill()
                holder["ctx"].synchronize()
                tt = torch.tensor([t_rs], dtype=torch.float64, device=f"cuda:{local_rank}")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                res["prefill_tok_s_rsag"] = round(args.prompt / float(tt[0].item()), 1)
                res["rsag_exchanges"] = int(pkg.binding.tp_p2p_prompt_exchanges())
                res["rsag_ranks_agree_on_first_token"] = agree(tok_rs)
                # every rank receives the OWNER's sum in a reduce-scatter + all-gather, so the ranks agree with each other even when that sum is wrong (a stale
                # line at the owner): the figure is only adopted when the first token is also the one the RCCL prefill produced, on every rank
                same = torch.tensor([1 if tok_rs == tok_first_rccl else 0], dtype=torch.int64, device=f"cuda:{local_rank}")
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                res["rsag_matches_rccl"] = bool(int(same.item()) == 1)
                if (res["rsag_exchanges"] > 0 and res["rsag_ranks_agree_on_first_token"] and res["rsag_matches_rccl"]
                        and res["prefill_tok_s_rsag"] > res["prefill_tok_s"]):
                    res["prefill_tok_s"] = res["prefill_tok_s_rsag"]
                partial.pop("rsag_error", None)
            except Exception as e:  # noqa: BLE001 - reported in the record
                res["rsag_error"] = f"{type(e).__name__}: {e}"[:200]
                partial["rsag_error"] = res["rsag_error"]
                pkg.Backend().set_option("tp_p2p_prompt", 0)
        else:
            pkg.Backend().set_option("tp_p2p", 0)
            res["p2p_error"] = why or "another rank could not map its peers' buffers"
        partial.update(res)
    holder["ctx"].close(); model.close()
    pkg.binding.tp_shutdown()
    return res


def simulate(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    step_s = 0.002 * (1 + rank)                      # rank r is (r + 1) x slower: the MAX over ranks must win
    for _ in range(args.warmup):
        time.sleep(step_s)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(step_s)
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        out = {"metric": "decode tok/s (simulated step)", "unit": "tok/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "simulated",
               "config": {"workload": "sleep-based stand-in for the decode step", "parallelism": f"{world} replicas, no collective"},
               "roofline": None, "cpu_baseline": None, "ranks_seen": dist.get_world_size() if dist is not None else 1}
        out.update(aggregate(args.gpus, args.steps, dt))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--config", default="llama-3-8b")
    ap.add_argument("--ftype", default="q4_k_m")
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--cache-type", default="q8_0", choices=["f16", "q8_0", "q4_0"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12)
    ap.add_argument("--model-dir", default=os.environ.get("MI355_BENCH_DIR", "/tmp"))
    ap.add_argument("--keep-model", action="store_true")
    ap.add_argument("--no-long-context", action="store_true", help="skip the context-filled-to-3968 measurement")
    ap.add_argument("--no-row-split", action="store_true", help="N > 1: skip the row-split measurement after the replicas one")
    ap.add_argument("--no-p2p", action="store_true", help="row split: every exchange through RCCL (no peer-to-peer all-reduce kernel)")
    ap.add_argument("--row-split-timeout", type=float, default=240.0, help="seconds the row-split section may take before it is abandoned")
    ap.add_argument("--simulate", action="store_true",
                    help="no GPU: the same rank bookkeeping (rendezvous over gloo, barriers, MAX over ranks, rank-0 JSON) around a "
                         "sleep standing in for the decode step; used by the two-rank CPU test")
    args = ap.parse_args()
    # --gpus N must mean N ranks.  Started without a launcher (no WORLD_SIZE) and N > 1: start N FRESH child processes under torch.distributed.run - as the
    # very first thing, before any import that could initialise the GPU (never an exec of a process that has) - and relay their output (rank 0 prints the
    # line); started under a launcher whose world size is not N: refuse.  One rank printing "n_gpus": N can then not happen.
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        return subprocess.run(cmd).returncode
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks: refusing to report a line for {args.gpus} GPUs",
              file=sys.stderr, flush=True)
        return 2
    if args.simulate:
        return simulate(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    dist = None
    torch = None
    # MI355_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers, row-split section) with one rank
    if world > 1 or os.environ.get("MI355_BENCH_FORCE_DIST") == "1":
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    gs = pkg.gguf_synth
    pkg.Backend()

    cfg = gs.CONFIGS[args.config]
    path = os.path.join(args.model_dir, f"mi355-bench-{args.config}-{args.ftype}.gguf")
    t0 = time.time()
    if rank == 0 and not os.path.exists(path):
        gs.write_synthetic_llama(path + ".tmp", cfg, args.ftype, seed=0xC0FFEE, with_vocab=False)
        os.replace(path + ".tmp", path)
    if dist is not None:
        dist.barrier()
    t_gen = time.time() - t0

    KV = {"f16": 1, "q8_0": 8, "q4_0": 2}[args.cache_type]
    t0 = time.time()
    model = pkg.Model(path, main_gpu=local_rank if world > 1 else 0)
    t_load = time.time() - t0
    ctx = pkg.Context(model, n_ctx=args.ctx, n_batch=2048, n_ubatch=N_UBATCH, type_k=KV, type_v=KV, flash_attn=True, use_graphs=True)

    rng = np.random.default_rng(1234 + rank)
    prompt = rng.integers(0, model.n_vocab, args.prompt)

    # ---- prefill (timed separately; the reference's prompt_per_second, llama_client_slot.cc:62-76)
    def prefill():
        ctx.kv_clear()
        t = time.perf_counter()
        for i0 in range(0, args.prompt, 2048):           # n_batch chunks like UpdateSlots (ctx.cc:1628)
            chunk = prompt[i0:i0 + 2048]
            rc = ctx.decode(chunk, np.arange(i0, i0 + chunk.size))
            assert rc == 0, rc
        tok = ctx.argmax()
        return time.perf_counter() - t, tok

    prefill()                                            # warm (first-touch, graph capture happens on first decode step)
    t_prefill, tok = prefill()

    def step(tok, pos):
        rc = ctx.decode([tok], [pos])
        assert rc == 0, rc
        ctx.logits_ready()                               # logits row host-visible = llama_decode's contract
        return ctx.argmax()

    pos = args.prompt
    for _ in range(args.warmup):
        tok = step(tok, pos)
        pos += 1

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tok = step(tok, pos)
        pos += 1
    ctx.synchronize()
    dt = time.perf_counter() - t0
    sync_all()
    # the same K steps under the same contract (mi355_decode of one token, its logits row host-visible, the arg-max fed back) driven from the C side
    # (mi355_greedy_steps: the inner loop of the reference's C++ slot loop) instead of from this script - `value` stays the script-driven loop; the difference is
    # the per-step cost of the scripting caller (numpy marshalling + three ctypes calls), which the C++ host of north_star does not pay
    c_loop_tok_s = None
    pos_timed_end = pos                                    # (the headline's timed steps end here)
    if pos + args.steps < args.ctx - 1:
        t0c = time.perf_counter()
        toks_c = ctx.greedy_steps(tok, pos, args.steps)
        ctx.synchronize()
        c_loop_tok_s = args.steps / (time.perf_counter() - t0c)
        tok, pos = int(toks_c[-1]), pos + args.steps
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    def finish_rank(out):
        """Everything after the replicas measurement that involves all ranks: the row-split section (fenced), then the line."""
        if dist is not None and not args.no_row_split:
            import threading

            rs_partial = {}                                # what the section has measured so far (the RCCL timing comes first)

            def split_headline(o, rs):
                if args.config == "llama-3-70b" and "error" not in rs:
                    # BASELINE config 5 is defined as the row split of ONE sequence over the GPUs: that is the headline of this config
                    # (the replicas figure stays in the line as `replicas_value`)
                    o["replicas_value"], o["replicas_ms_per_step"] = o["value"], o["ms_per_step"]
                    o["value"], o["ms_per_step"], o["scaling"] = rs["decode_tok_s"], rs["ms_per_step"], "strong"
                    o["config"]["parallelism"] = rs["parallelism"]

            def give_up():                                 # a rank stuck in a collective cannot be recovered in-process
                if rank == 0 and out is not None:
                    if "decode_tok_s" in rs_partial:
                        out["row_split"] = dict(rs_partial, p2p_error=f"abandoned after {args.row_split_timeout:.0f} s")
                        split_headline(out, out["row_split"])
                    else:
                        out["row_split"] = {"error": f"abandoned after {args.row_split_timeout:.0f} s"}
                    print(json.dumps(out), flush=True)     # the replicas measurement is complete: keep the line
                else:
                    print(f"[bench rank {rank}] row-split section abandoned after {args.row_split_timeout:.0f} s", file=sys.stderr, flush=True)
                os._exit(3)                                # a process that touched the GPU and gave up must not report success
            wd = threading.Timer(args.row_split_timeout, give_up)
            wd.daemon = True
            wd.start()
            try:
                rs = row_split_section(args, pkg, path, KV, rank, local_rank, world, dist, torch, rs_partial)
            except Exception as e:  # noqa: BLE001 - reported in the line; the other ranks run into the watchdog
                rs = {"error": f"{type(e).__name__}: {e}"[:300]}
            if rank == 0 and out is not None:
                out["row_split"] = rs
                split_headline(out, rs)
                print(json.dumps(out), flush=True)
            if "error" in rs:                              # peers may be stuck: do not enter another collective
                if rank != 0:
                    print(f"[bench rank {rank}] row-split section failed: {rs['error']}", file=sys.stderr, flush=True)
                os._exit(3)
            wd.cancel()
        elif rank == 0 and out is not None:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()

    if rank != 0:
        ctx.close(); model.close()
        if dist is not None:
            dist.barrier()                                 # rank 0 finishes its single-GPU extras first
        finish_rank(None)
        return 0

    # ---- the same per-role figures measured LIVE in this run: eager single-token steps with one HIP event per role boundary on the
    # context's stream (mi355_profile_enable / mi355_profile_last_decode) AND, for the weight-stream and attention launches, the dispatch's own begin / end
    # timestamps (hipExtLaunchKernelGGL start / stop events, host/runtime.cc KTimer): the latter is the quantity rocprofv3's kernel trace reports and is what
    # `roles_live` / `frac_live` are computed from; the event intervals (which include the marker's cost and the launch boundary) are kept beside them.
    # `roles` / `frac_rocprof` (the committed trace) are the cross-check
    roles_live, frac_live = None, None
    if world == 1 and cfg.n_expert == 0:
        gs = pkg.gguf_synth
        rb = {"qkv": 0, "attn_out": 0, "ffn_gate_up": 0, "ffn_down": 0, "lm_head": 0}
        for name, ne, t, _ in gs.model_tensors(cfg, args.ftype):
            nbytes = gs.row_bytes(t, ne[0]) * (int(np.prod(ne)) // ne[0])
            if name == "output.weight":
                rb["lm_head"] += nbytes
                continue
            for key, role in ((".attn_q.", "qkv"), (".attn_k.", "qkv"), (".attn_v.", "qkv"), (".attn_output.", "attn_out"), (".ffn_gate.", "ffn_gate_up"), (".ffn_up.", "ffn_gate_up"),
                              (".ffn_down.", "ffn_down")):
                if key in name and name.endswith("weight"):
                    rb[role] += nbytes
        n_prof = 16
        acc = {}
        ctx.profile(True)
        try:
            pos_p, tok_p = pos, tok
            for _ in range(n_prof):
                if pos_p >= args.ctx - 1:
                    break
                ctx.decode([tok_p], [pos_p]); tok_p = ctx.argmax(); pos_p += 1
                for k, v in ctx.last_profile().items():
                    acc[k] = acc.get(k, 0.0) + v
            n_done = pos_p - pos
        finally:
            ctx.profile(False)
        # kernel begin / end timestamps of the stream and attention launches themselves ("k:<role>" = us, "n:<role>" = launches; hipExtLaunchKernelGGL's start / stop
        # events: what rocprofv3's kernel trace reports, no marker between two launches involved)
        kroles = {"qkv": "qkv", "gate_up": "ffn_gate_up", "ffn_down": "ffn_down", "head": "lm_head", "attn_out": "attn_out", "qkv_attn_out": "qkv_attn_out"}
        kernel_us = {kroles[k[2:]]: v / n_done for k, v in acc.items() if k.startswith("k:") and k[2:] in kroles} if n_done > 0 else {}
        kernel_n = {kroles[k[2:]]: v / n_done for k, v in acc.items() if k.startswith("n:") and k[2:] in kroles} if n_done > 0 else {}
        # round 6: where the step runs Q | K | V inside its attention + attn_output launch (csrc/attn_out.hip QF) there is ONE launch per layer for the whole attention
        # block - role "qkv_attn_out", weight bytes = attn_q + attn_k + attn_v + attn_output - and the weight-stream launches of a token are gate | up, ffn_down, head
        qf = kernel_us.get("qkv_attn_out", 0.0) > 0
        if qf:
            rb["qkv_attn_out"] = rb["qkv"] + rb["attn_out"]
        stream_roles = ("ffn_gate_up", "ffn_down", "lm_head") if qf else ("qkv", "ffn_gate_up", "ffn_down", "lm_head")
        if n_done > 0 and all(k in acc for k in ("qkv", "ffn_gate_up", "ffn_down", "lm_head")):
            L_n = cfg.n_layer
            roles_live = {}
            role_list = ((("qkv_attn_out", L_n),) if qf else (("qkv", L_n), ("attn_out", L_n))) + (("ffn_gate_up", L_n), ("ffn_down", L_n), ("lm_head", 1))
            for role, n_l in role_list:
                if role == "qkv_attn_out":
                    us_tok = (acc.get("qkv", 0.0) + acc.get("attn", 0.0) + acc.get("rope_kv", 0.0) + acc.get("attn_out", 0.0)) / n_done
                else:
                    us_tok = (acc.get(role, 0.0) + (acc.get("attn", 0.0) + acc.get("rope_kv", 0.0) if role == "attn_out" else 0.0)) / n_done
                if us_tok <= 0:
                    continue
                ev_us = us_tok                            # interval between the role's HIP events (includes the marker's own cost and the launch boundary)
                if role in kernel_us and kernel_us[role] > 0:
                    us_tok = kernel_us[role]              # the kernels' own durations
                roles_live[role] = {"launches_per_token": n_l, "avg_us": round(us_tok / n_l, 3), "us_per_token": round(us_tok, 2), "weight_bytes_per_token": int(rb[role]),
                                    "GBps": round(rb[role] / (us_tok * 1e-6) / 1e9, 1), "frac_of_8TBps": round(rb[role] / (us_tok * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                    "source": "kernel begin/end timestamps" if role in kernel_us else "HIP event interval", "event_interval_us_per_token": round(ev_us, 2),
                                    "timed_launches_per_token": round(kernel_n.get(role, 0.0), 2)}
            st_b = sum(rb[r] for r in stream_roles)
            st_us = sum(roles_live[r]["us_per_token"] for r in stream_roles)
            frac_live = round(st_b / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)
            live_launches = sum(roles_live[r]["launches_per_token"] for r in stream_roles)
            roles_live["_stream"] = {"launches_per_token": live_launches, "us_per_token": round(st_us, 2), "avg_launch_us": round(st_us / live_launches, 3),
                                     "weight_bytes_per_token": int(st_b), "roles": list(stream_roles)}
            # every launch of a token that reads weights, the attention block included: all weight bytes / the sum of those kernels' own durations
            all_roles = [r for r in roles_live if not r.startswith("_")]
            all_b = sum(rb[r] for r in all_roles)
            all_us = sum(roles_live[r]["us_per_token"] for r in all_roles)
            roles_live["_all_weight_launches"] = {"launches_per_token": sum(roles_live[r]["launches_per_token"] for r in all_roles), "us_per_token": round(all_us, 2),
                                                  "weight_bytes_per_token": int(all_b), "frac_of_8TBps": round(all_b / (all_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)}
            roles_live["_note"] = (f"{n_done} eager single-token steps at pos {pos}.., HIP events between the roles on the context's stream; " +
                                   ("qkv_attn_out = RMSNorm / Q8_K + Q | K | V + rope / KV store + attention + attn_output in ONE launch per layer (round 6), weight bytes of the four "
                                    "projections; " if qf else "attn_out = rope / KV store + attention + attn_output (one launch) with attn_output's weight bytes only; ") +
                                   "other per-token time: " +
                                   ", ".join(f"{k} {v / n_done:.1f} us" for k, v in sorted(acc.items()) if k not in ("qkv", "attn", "rope_kv", "attn_out", "ffn_gate_up", "ffn_down", "lm_head")))

    # ---- the long-context point of the same config under the SAME contract as the headline (logits row host-visible after every step): context
    # filled to 3968 of 4096, then greedy decode
    long_host = None
    if args.ctx >= 4096 and not args.no_long_context:
        fill = args.ctx - 128
        lp0 = np.random.default_rng(4321).integers(0, model.n_vocab, fill)
        ctx.kv_clear()
        for i0 in range(0, fill, 2048):
            chunk = lp0[i0:i0 + 2048]
            assert ctx.decode(chunk, np.arange(i0, i0 + chunk.size)) == 0
        tok_l, pos_l = ctx.argmax(), fill
        for _ in range(8):
            tok_l = step(tok_l, pos_l); pos_l += 1
        ctx.synchronize()
        n_l = min(64, args.ctx - pos_l)
        t0 = time.perf_counter()
        for _ in range(n_l):
            tok_l = step(tok_l, pos_l); pos_l += 1
        ctx.synchronize()
        long_host = n_l / (time.perf_counter() - t0)

    # ---- device-greedy variant: logits stay on the device, only the argmax crosses
    ctx.close()
    ctx = pkg.Context(model, n_ctx=args.ctx, n_batch=2048, n_ubatch=N_UBATCH, type_k=KV, type_v=KV, flash_attn=True, use_graphs=True,
                      logits_to_host=False)
    _, tok = prefill()
    pos = args.prompt
    for _ in range(args.warmup):
        ctx.decode([tok], [pos]); tok = ctx.argmax(); pos += 1
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decode([tok], [pos])
        tok = ctx.argmax()
        pos += 1
    ctx.synchronize()
    dt_greedy = time.perf_counter() - t0

    # ---- the long-context point of the same config: context filled to 3968 of 4096, then greedy decode
    long_ctx = None
    if args.ctx >= 4096 and not args.no_long_context:
        fill = args.ctx - 128
        lp = rng.integers(0, model.n_vocab, fill)
        ctx.kv_clear()
        t0 = time.perf_counter()
        for i0 in range(0, fill, 2048):
            chunk = lp[i0:i0 + 2048]
            assert ctx.decode(chunk, np.arange(i0, i0 + chunk.size)) == 0
        tok_l = ctx.argmax()
        t_fill = time.perf_counter() - t0
        pos_l = fill
        for _ in range(8):
            ctx.decode([tok_l], [pos_l]); tok_l = ctx.argmax(); pos_l += 1
        ctx.synchronize()
        n_l = min(64, args.ctx - pos_l)
        t0 = time.perf_counter()
        for _ in range(n_l):
            ctx.decode([tok_l], [pos_l]); tok_l = ctx.argmax(); pos_l += 1
        ctx.synchronize()
        dt_l = time.perf_counter() - t0
        b_per_l = {"f16": 2.0, "q8_0": 34.0 / 32.0, "q4_0": 18.0 / 32.0}[args.cache_type]
        kv_l = 2 * cfg.n_layer * cfg.n_head_kv * cfg.head_dim * (pos_l - n_l // 2) * b_per_l
        long_ctx = {"prompt": fill, "prefill_tok_s": round(fill / t_fill, 1), "decode_tok_s": round(long_host, 2) if long_host else None,
                    "decode_tok_s_device_greedy": round(n_l / dt_l, 2),
                    "decode_pos": [pos_l - n_l, pos_l], "kv_bytes_per_token": int(kv_l),
                    "decode_hbm_fraction_of_8TBps": round((model.bytes_per_token + kv_l) * (long_host or n_l / dt_l) / (HBM_PEAK_GBPS * 1e9), 4)}

    # ---- roofline of the dominant kernel (quantised mat-vec), HIP events on the kernel's own stream
    # the launches of the weight-stream kernel as the step issues them: qkv, gate+up, down per layer (all selected experts of a mixture-of-experts layer
    # share the two launches) + lm-head; attn_output is one of them only where it does not run inside the attention launch (attn_out.hip, round 4)
    sweep_us, sweep_bytes, n_launch = ctx.weight_sweep(iters=5)
    achieved = sweep_bytes / (sweep_us * 1e-6) / 1e9
    hbm_read = pkg.Backend().hbm_read_gbps(2 << 30, 5)

    # HBM bytes the same sweep moved, from the PMC pass committed under profiles/ (FETCH_SIZE, corrected x2 per the gfx950
    # note of the microarchitecture guide); null when the profile is absent or is for another workload
    traffic, traffic_note = None, "no PMC profile for this workload"
    root = os.path.dirname(os.path.abspath(__file__))
    sha_now = kernel_sources_sha256(root)

    def committed_profile(suffix):
        """the newest profiles/r<N>_<suffix> (by round number); the caller checks its kernel_sources_sha256 against this tree's"""
        import glob
        import re
        best = None
        for f in glob.glob(os.path.join(root, "profiles", "r*_" + suffix)):
            m = re.match(r"r(\d+)_", os.path.basename(f))
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), f)
        return best[1] if best else os.path.join(root, "profiles", "r0_" + suffix)
    tpath = committed_profile("pmc_decode_traffic.json")
    tname = "profiles/" + os.path.basename(tpath)
    if os.path.exists(tpath) and args.config == "llama-3-8b" and args.ftype == "q4_k_m":
        try:
            with open(tpath) as f:
                tj = json.load(f)
            # the profile names the kernel sources it was taken from (sha256 over the mat-vec kernel files): a kernel change since then makes the
            # figure stale, and it is withheld rather than repeated
            if tj.get("kernel_sources_sha256") == sha_now:
                traffic, traffic_note = int(tj["matvec_hbm_read_bytes_per_token"]), tname + " (rocprofv3 --pmc FETCH_SIZE, x2 per the gfx950 note)"
            else:
                traffic_note = tname + " was taken from other kernel sources than this tree's: withheld"
        except (OSError, ValueError, KeyError):
            traffic = None
    # the same fraction from the TRACED durations of the committed rocprofv3 summary (eager launches, per-kernel averages) instead of the graph sweep
    frac_rocprof, frac_rocprof_note, roles_rocprof = None, "no rocprofv3 summary for this workload", None
    rpath = committed_profile("rocprof_decode_roofline.json")
    rname = "profiles/" + os.path.basename(rpath)
    if os.path.exists(rpath) and args.config == "llama-3-8b" and args.ftype == "q4_k_m":
        try:
            with open(rpath) as f:
                rj = json.load(f)
            if rj.get("kernel_sources_sha256") == sha_now:
                frac_rocprof = float(rj["frac_rocprof"])
                roles_rocprof = rj.get("roles")
                frac_rocprof_note = (f"{rname}: {rj['stream_weight_bytes_per_token']} B / {rj['stream_us_per_token']} us of traced "
                                     f"mmvq_stream_kernel time per token ({rj['stream_avg_launch_us']} us per launch)")
            else:
                frac_rocprof_note = rname + " was taken from other kernel sources than this tree's: withheld"
        except (OSError, ValueError, KeyError):
            frac_rocprof = None

    # prompt processing against the matrix-core peak: 2 * (projection weights) * tokens, int8 MFMA dense peak
    E, FF, L_, GD = cfg.n_embd, cfg.n_ff, cfg.n_layer, cfg.n_head_kv * cfg.head_dim
    p_layer = E * E * 2 + 2 * E * GD + 3 * E * FF
    prefill_ops = 2.0 * p_layer * L_ * args.prompt + 4.0 * (args.prompt / 2.0) * E * L_ * args.prompt
    prefill_tops = prefill_ops / t_prefill / 1e12

    kv_pos_mid = args.prompt + args.warmup + args.steps // 2
    b_per = {"f16": 2.0, "q8_0": 34.0 / 32.0, "q4_0": 18.0 / 32.0}[args.cache_type]
    kv_bytes = 2 * cfg.n_layer * cfg.n_head_kv * cfg.head_dim * kv_pos_mid * b_per
    tok_s = aggregate(n_gpus, args.steps, dt)["value"]
    decode_frac = (model.bytes_per_token + kv_bytes) * (args.steps / dt) / (HBM_PEAK_GBPS * 1e9)

    out = {
        "metric": "decode tok/s, Llama-3-8B Q4_K_M GGUF (prefill tok/s in `prefill_tok_s`)" if (args.config, args.ftype) == ("llama-3-8b", "q4_k_m")
                  else f"decode tok/s, {cfg.name} {args.ftype.upper()} GGUF (prefill tok/s in `prefill_tok_s`)",
        "value": tok_s,
        "unit": "tok/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8 dot (Q4_K/Q6_K x Q8_K), f32 accumulate",
        "data": "synthetic",
        "config": {
            "workload": f"{cfg.name} {args.ftype.upper()} synthetic GGUF, flash_attn, cache_type={args.cache_type}, "
                        f"n_ctx={args.ctx}, prompt={args.prompt}, greedy decode at pos {args.prompt + args.warmup}..{pos_timed_end}",
            "parallelism": "single GPU" if n_gpus == 1 else f"{n_gpus} replicas (one sequence per GPU, no collective)",
            "weight_bytes_per_token": int(model.bytes_per_token),
            "kv_bytes_per_token_mid": int(kv_bytes),
        },
        "prefill_tok_s": round(args.prompt / t_prefill, 1),
        "prefill_ms": round(t_prefill * 1e3, 2),
        "decode_tok_s_device_greedy": round(args.steps / dt_greedy, 2),
        # K more steps of the headline's contract (logits row host-visible every step) with the loop on the C side (mi355_greedy_steps); not `value`
        "decode_tok_s_c_loop": round(c_loop_tok_s, 2) if c_loop_tok_s else None,
        "decode_hbm_fraction_of_8TBps": round(decode_frac, 4),
        "roofline": {
            "bound": "hbm",
            "kernel": "mmvq_stream_kernel (single-token quantised mat-vec as an LDS-DMA weight stream: gate|up, ffn_down of every layer and the output head; Q|K|V and attn_output run inside the "
                      "attention launch since round 6 / 4 - role qkv_attn_out of roles_live, and roles_live._all_weight_launches prices every weight byte of a token against every such launch)",
            # `achieved` / `frac`: measured LIVE in this run - algorithmic weight bytes of the stream launches of a token / the sum of those kernels' own durations
            # (begin / end timestamps of every dispatch, `roles_live`), averaged over the profiled steps; the committed rocprofv3 trace of the same kernel sources
            # (`frac_rocprof`, `roles`) is the cross-check and `frac_live_over_rocprof` their ratio; the hipGraph sweep (launch boundaries counted as kernel time) beside them
            "achieved": round(frac_live * HBM_PEAK_GBPS, 1) if frac_live else (round(frac_rocprof * HBM_PEAK_GBPS, 1) if frac_rocprof is not None else round(achieved, 1)),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": frac_live if frac_live else (frac_rocprof if frac_rocprof is not None else round(achieved / HBM_PEAK_GBPS, 4)),
            "frac_source": "live: kernel begin/end timestamps of this run (roles_live)" if frac_live else
                           ("rocprofv3 trace (profiles/)" if frac_rocprof is not None else "live hipGraph sweep (no committed trace of these kernel sources)"),
            "frac_graph_sweep": round(achieved / HBM_PEAK_GBPS, 4),
            "achieved_graph_sweep": round(achieved, 1),
            "frac_rocprof": frac_rocprof,
            "frac_rocprof_source": frac_rocprof_note,
            "roles": roles_rocprof,
            "roles_live": roles_live,
            "frac_live": frac_live,
            "frac_live_over_rocprof": round(frac_live / frac_rocprof, 4) if (frac_live and frac_rocprof) else None,
            "traffic": traffic,
            "traffic_source": traffic_note,
            "method": "the step's own launches of the weight-stream kernel (gate|up, ffn_down of every layer + the output head: 65 for this model; Q|K|V and "
                      "attn_output run inside the attention launch since rounds 6 / 4 and are neither launched nor counted here), replayed from a hipGraph, HIP "
                      "events on the context's stream; achieved = algorithmic weight bytes of those launches / sweep time, so launch boundaries count as "
                      "kernel time",
            "bytes_per_sweep": int(sweep_bytes),
            "launches_per_sweep": n_launch,
            "avg_launch_us": round(sweep_us / n_launch, 3),
            "measured_stream_read_GBps": round(hbm_read, 1),
        },
        "prefill_roofline": {
            "bound": "mfma",
            "kernel": "mmq_planes2_kernel / mmq_planes_kernel (int8 v_mfma_i32_32x32x32_i8, weights pre-expanded into two exact int8 planes)",
            "achieved": round(prefill_tops, 1),
            "peak": MFMA_I8_PEAK_TOPS,
            "unit": "TOP/s",
            "frac": round(prefill_tops / MFMA_I8_PEAK_TOPS, 4),
            "note": "algorithmic ops (2 x weights x tokens + attention); the kernel issues 2x the MACs (hi / lo planes)",
            "planes_bytes": int(model.planes_bytes),
        },
        "long_context": long_ctx,
        "ranks_seen": int(dist.get_world_size()) if dist is not None else 1,    # ranks of the process group this line was measured over (== n_gpus by construction)
        "load_s": round(t_load, 2),
        "synth_s": round(t_gen, 2),
    }

    # ---- CPU baseline: the oracle (a port, not the reference binary) on the host cores, bounded sample
    if not args.no_cpu_baseline and world == 1:
        import oracle_py as oq
        try:
            nth = len(os.sched_getaffinity(0))
        except AttributeError:
            nth = os.cpu_count() or 1
        nth = max(1, min(nth, 32))                           # the scalar port stops scaling (and oversubscribes) beyond this
        om = oq.OracleModel(path)
        oc = oq.OracleContext(om, 64, KV, KV, True, nth)
        oc.decode(prompt[:4], np.arange(4))              # untimed warm-up (page-in of the mmap'd weights)
        t0 = time.perf_counter()
        t_ = int(prompt[4])
        for s in range(args.cpu_steps):
            r = oc.decode([t_], [4 + s])[0]
            t_ = int(r.argmax())
        dt_cpu = time.perf_counter() - t0
        oc.close(); om.close()
        out["cpu_baseline"] = {
            "value": round(args.cpu_steps / dt_cpu, 3), "unit": "tok/s", "cores": nth, "kind": "port",
            "sample": f"{args.cpu_steps} greedy decode steps at pos 4.. of the same GGUF with the scalar CPU restatement "
                      f"(oracle/, OpenMP over weight rows, {nth} threads)",
        }
    else:
        out["cpu_baseline"] = None

    ctx.close(); model.close()
    if dist is not None:
        dist.barrier()                                     # releases the other ranks into the row-split section
    finish_rank(out)
    if not args.keep_model and world == 1 and os.environ.get("MI355_BENCH_KEEP") is None:
        try:
            os.remove(path)
        except OSError:
            pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
