"""`python bench.py --gpus N` must work as typed (the parent starts the N ranks itself, before any GPU call) and under
torch.distributed.run as the driver launches it.  Checked here on CPU with the gloo backend and `--launch-check`, which runs
the multi-rank plumbing (rendezvous, clip sharding, fixed-shape all-gather, max-over-ranks timing, JSON relay) with synthetic
detection rows instead of the model."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
FLAGS = ["--gpus", "2", "--launch-check", "--backend", "gloo", "--clips", "3", "--steps", "3", "--warmup", "1"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.lstrip().startswith("{")]


def test_gpus_2_as_typed_self_launches_two_ranks():
    p = subprocess.run([sys.executable, BENCH] + FLAGS, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout
    r = lines[0]
    assert r["n_gpus"] == 2 and r["gather_ok"] is True and r["config"]["frames_per_step"] == 6 and r["steps"] == 3


def test_under_torch_distributed_run_as_the_driver_launches_it():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH] + FLAGS
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["gather_ok"] is True


def test_world_size_mismatch_is_an_error_not_a_hang():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH] + FLAGS, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
