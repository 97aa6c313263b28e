"""The N > 1 path of bench.py on CPU: two ranks over gloo (torch.distributed.run, 127.0.0.1), barriers on both sides of the
timed region, MAX over ranks, one JSON line from rank 0, whole-job value = ranks x steps / slowest rank's time."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_bench_line_over_gloo():
    steps, warmup = 20, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
           "--warmup", str(warmup), "--simulate"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak"
    # rank 1 sleeps 4 ms per step: the slowest rank sets the time, both ranks' tokens count
    assert 4.0 <= out["ms_per_step"] <= 8.0, out
    assert abs(out["value"] - 2 * 1000.0 / out["ms_per_step"]) <= 0.02 * out["value"]


def test_single_rank_simulated_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "1", "--simulate"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and 2.0 <= out["ms_per_step"] <= 4.0


def test_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher must not print a 2-GPU line from ONE rank: it starts two fresh ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "1", "--simulate"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == out["ranks_seen"] == 2, out
    assert 4.0 <= out["ms_per_step"] <= 8.0, out            # the slower simulated rank (4 ms per step) sets the time: both ranks really ran


def test_world_size_that_contradicts_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--simulate"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout
