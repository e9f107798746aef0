"""The N-rank flow of bench.py END TO END on the GPU box: two ranks under torch.distributed.run, both on device 0 (--share-gpu, gloo exchange -- RCCL refuses two
ranks on one device), through every pass of main(): timed region, no-gather pass, instrumented pass, the sampler pass, max-over-ranks timing, rank 0's result line,
the final barrier.  No 8-GPU node is available to the builder; this is what executes the rank-symmetric structure of the file for real (round 6 found a pass that
only rank 0 ran, with barriers and all-gathers inside: it would have hung every N > 1 run on hardware)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_run_the_whole_bench_flow_and_print_one_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--clips", "2", "--steps", "4", "--warmup", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.lstrip().startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and d["config"]["frames_per_step"] == 4 and d["scaling"] == "weak"
    c = d["collective"]
    assert c["backend"] == "gloo" and c["world_size"] == 2 and c["gathered_shape"][0] == 4 and c["last_gather_equals_local_block"] is True
    assert "roofline" in d and d["roofline_im2col"]["launches"] > 0            # the sampler pass ran (on BOTH ranks) and rank 0 reported it
    assert "extras" not in d and "cpu_baseline" not in d                       # N = 1 only


def test_bench_as_typed_with_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 ...` without a launcher: the parent (which touches no GPU API) starts the two ranks as a child process and relays rank 0's line."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--clips", "1", "--steps", "3", "--warmup", "2"]
    out = subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "clip-dp2" and d["collective"]["world_size"] == 2
