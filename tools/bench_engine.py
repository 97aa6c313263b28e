#!/usr/bin/env python3
"""Load test through the engine surface (mi355_engine_*), the shape of the reference's own scripts/benchmark.py
(40 users x rounds of /v1/chat/completions against one model loaded with n_parallel 32, ctx_len 1000 per sequence):
aggregate completion tokens per second over the whole run — slot loop, tokenizer, host sampler and JSON included.
usage: tools/bench_engine.py [--users 40] [--rounds 3] [--max-tokens 200] [--n-parallel 32]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--max-tokens", type=int, default=200)
    ap.add_argument("--n-parallel", type=int, default=32)
    ap.add_argument("--ctx-per-seq", type=int, default=1000)
    ap.add_argument("--config", default="llama-3-8b")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("--device-sampling", type=int, default=1, help="0: the whole sampler chain on the host over the whole logits row (load option device_sampling)")
    args = ap.parse_args()
    pkg = ge.load_pkg()
    gs = pkg.gguf_synth
    path = f"/tmp/mi355-engine-{args.config}-q4_k_m-vocab.gguf"
    if not os.path.exists(path):
        gs.write_synthetic_llama(path, gs.CONFIGS[args.config], "q4_k_m", seed=0xC0FFEE, with_vocab=True)
    eng = pkg.Engine()
    t0 = time.time()
    st, body = eng.load_model(llama_model_path=path, ctx_len=args.ctx_per_seq * args.n_parallel, n_parallel=args.n_parallel, ngl=300,
                              model_alias="bench", user_prompt="user: ", ai_prompt="assistant: ", device_sampling=bool(args.device_sampling))
    assert st["status_code"] == 200, (st, body)
    print(f"load: {time.time() - t0:.1f} s", flush=True)
    extra = dict(temperature=0.0, repeat_penalty=1.0) if args.greedy else {}
    prompts = ["what is a gpu and how does it differ from a cpu in practice", "write a haiku about memory bandwidth",
               "summarise the plot of a story about a lighthouse keeper", "list five prime numbers and explain why they are prime"]
    eng.chat_completion(model="bench", messages=[{"role": "user", "content": "warm up"}], max_tokens=8, **extra)
    done_tokens, prompt_tokens, lock = [0], [0], threading.Lock()

    def user(u):
        for r in range(args.rounds):
            res = eng.chat_completion(model="bench", messages=[{"role": "user", "content": prompts[(u + r) % len(prompts)]}],
                                      max_tokens=args.max_tokens, ignore_eos=True, **extra)[-1]
            usage = res[1].get("usage", {})
            with lock:
                done_tokens[0] += int(usage.get("completion_tokens", 0))
                prompt_tokens[0] += int(usage.get("prompt_tokens", 0))

    t0 = time.time()
    th = [threading.Thread(target=user, args=(u,)) for u in range(args.users)]
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.time() - t0
    print(f"users={args.users} rounds={args.rounds} n_parallel={args.n_parallel} max_tokens={args.max_tokens} greedy={args.greedy} device_sampling={args.device_sampling}: "
          f"{done_tokens[0]} completion tokens (+{prompt_tokens[0]} prompt) in {dt:.2f} s = {done_tokens[0] / dt:.0f} tok/s aggregate")
    eng.close()


if __name__ == "__main__":
    main()
