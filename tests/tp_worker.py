"""One rank of a row-split (tensor-parallel) run, started by tests/test_gpu_tp.py — not a test module itself.

    python tests/tp_worker.py <plan.npz> <out.npz>      with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment

plan.npz: path (GGUF), kv (cache type id), transport ("host" | "rccl"), n_ctx, prompt (token ids), steps (teacher-forced
token ids, one single-token decode each), tail (token ids decoded as ONE batch with every row flagged).
out.npz (rank 0): logits rows in that order, the residual stream after every layer for the prompt, the model's local
head counts and per-rank bytes per token.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    plan = np.load(sys.argv[1], allow_pickle=False)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    transport = str(plan["transport"])
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg.Backend()
    p2p = int(plan["p2p_floats"]) if "p2p_floats" in plan.files else 0
    p2p_prompt = int(plan["p2p_prompt_floats"]) if "p2p_prompt_floats" in plan.files else 0
    pkg.binding.tp_init(rank, world, device=0, transport=transport, p2p_floats=p2p, p2p_prompt_floats=p2p_prompt)
    model = pkg.Model(str(plan["path"]), tp_rank=rank, tp_size=world)
    ctx = pkg.Context(model, n_ctx=int(plan["n_ctx"]), type_k=int(plan["kv"]), type_v=int(plan["kv"]), n_ubatch=int(plan["n_ubatch"]))
    rows = []
    prompt = plan["prompt"]
    if world > 1:
        # what a row split's driver provides in production: the ranks are stepped in lock-step (rank 0 hands every rank the batch), so they reach an exchange
        # together.  Eight python processes that each import torch, load a shard and build a context on one shared box are SECONDS apart at their first batch
        # (round 6 trace: 3.5 s) - more than an in-kernel wait for a peer should ever sit out
        dist.barrier()
    ctx.enable_taps(True)                          # residual stream after every layer, for the prompt
    assert ctx.decode(prompt, np.arange(prompt.size)) == 0
    rows.append(ctx.logits())
    # (the taps hold the last micro-batch of a prompt that was cut into several)
    taps = np.stack([ctx.layer_out(il, prompt.size).reshape(-1, model.n_embd) for il in range(model.n_layer)])
    ctx.enable_taps(False)
    pos = prompt.size
    for tok in plan["steps"]:
        assert ctx.decode([int(tok)], [pos]) == 0
        rows.append(ctx.logits())
        assert ctx.argmax() == int(rows[-1].argmax())
        pos += 1
    tail = plan["tail"]
    if tail.size:
        assert ctx.decode(tail, np.arange(pos, pos + tail.size), logits=np.ones(tail.size)) == 0
        for i in range(tail.size):
            rows.append(ctx.logits(i))
    if os.environ.get("MI355_TP_DUMP_ALL") and rank > 0:
        np.savez(sys.argv[2] + f".rank{rank}.npz", logits=np.stack(rows))
    if rank == 0:
        np.savez(sys.argv[2], logits=np.stack(rows), taps=taps, n_head=model.n_head, n_head_kv=model.n_head_kv,
                 bytes_per_token=model.bytes_per_token, p2p_exchanges=pkg.binding.tp_p2p_exchanges(),
                 p2p_prompt_exchanges=pkg.binding.tp_p2p_prompt_exchanges())
    ctx.close(); model.close()
    pkg.binding.tp_shutdown()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
