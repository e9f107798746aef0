"""Process plumbing of bench.py: `--gpus N` as typed (self-launch), the host-only launch check, barriers, the result line."""
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def barrier():
    """dist.barrier() that names this rank's GPU under the RCCL backend: the process group is created WITHOUT device_id (see main), so the first
    collective -- usually this barrier -- is what creates the communicator, and it must not have to guess the device."""
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()



# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
def self_launch(args, argv, script):
    """Parent of an N-rank run.  Touches no GPU API; starts torch.distributed.run as a child and relays its output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks finished without a result line\n")
        rc = 1
    return rc


def launch_check(args, rank, world):
    """The N>1 contract without the model: every rank packs deterministic rows for its clips (clip c of rank r = global clip
    r + c*world), steps are bracketed by barriers, the all-gather result is verified on every rank."""
    from stmask_amd import dist as sdist
    top_k = 8

    def rows(step):
        p = torch.zeros(args.clips, top_k, sdist.DET_COLS)
        for c in range(args.clips):
            g = rank + c * world
            p[c, :, 0] = g
            p[c, :, 1] = step
            p[c, : 1 + g % top_k, 7] = 1.0
        return p

    ok = True
    for t in range(args.warmup):
        sdist.all_gather_detections(rows(t))
    if dist.is_initialized():
        barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        full = sdist.all_gather_detections(rows(t))
        for r in range(world):
            blk = full[r * args.clips:(r + 1) * args.clips]
            want = torch.tensor([r + c * world for c in range(args.clips)], dtype=torch.float32)
            ok = ok and bool((blk[:, 0, 0] == want).all()) and bool((blk[:, 0, 1] == t).all())
    if dist.is_initialized():
        barrier()
    elapsed = time.perf_counter() - t0
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        flag = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if rank == 0:
        frames = world * args.clips * args.steps
        print(json.dumps({"metric": "launch-check (no model): gathered detection rows/s", "value": round(frames / elapsed, 2),
                          "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "launch_check": True, "gather_ok": ok,
                          "config": {"workload": "launch check", "clips_per_gpu": args.clips, "frames_per_step": world * args.clips,
                                     "parallelism": f"clip-dp{world}", "backend": args.backend}}), flush=True)
    return 0 if ok else 1


_RESULT_FD = None       # the real stdout of a rank under torch.distributed.run (see stdout_to_stderr)


def stdout_to_stderr():
    """RCCL prints a version banner on STDOUT when its communicator comes up (at the first collective); the contract is ONE JSON line on rank 0's
    stdout.  File descriptor 1 is pointed at stderr for the run, emit() writes the result line to the saved descriptor."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


# ---------------------------------------------------------------------------------------------------------------------
# CPU binding of a rank: eight Python launch loops on unpinned cores are the likeliest loss of an 8-GPU weak-scaling run
def parse_cpulist(text):
    """'0-15,128-143' -> [0..15, 128..143] (the format of sysfs local_cpulist)."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def rank_core_slice(cores, idx, n):
    """Rank idx's share of `cores` among n ranks: the idx-th n-th of EVERY run of consecutive core numbers -- a node's list is usually its physical
    cores followed by their SMT siblings ('0-63,128-191'), and a rank should get cores together with their own siblings, not another rank's.  Never
    empty while cores is not (with fewer cores than ranks they share)."""
    cores = sorted(cores)
    if not cores or n <= 0:
        return cores
    runs, start = [], 0
    for i in range(1, len(cores) + 1):
        if i == len(cores) or cores[i] != cores[i - 1] + 1:
            runs.append(cores[start:i])
            start = i
    out = []
    for run in runs:
        per = len(run) // n
        out += run[idx * per:(idx + 1) * per]
    return out if out else [cores[idx % len(cores)]]


def gpu_local_cpulist(pci_domain, pci_bus, pci_device, sysfs="/sys/bus/pci/devices"):
    """Cores of the NUMA node the GPU hangs off (sysfs local_cpulist of its PCI function); [] when the file is absent (containers may hide it)."""
    try:
        with open(os.path.join(sysfs, "%04x:%02x:%02x.0" % (pci_domain, pci_bus, pci_device), "local_cpulist")) as fh:
            return parse_cpulist(fh.read())
    except (OSError, ValueError):
        return []


def bind_rank_to_gpu_cores(local_rank, world):
    """Pin this rank (and the OpenMP / torch intra-op threads it starts later) to its share of the cores local to its GPU's NUMA node: the cores of
    that node that this process may use, cut into one slice per local rank whose GPU sits on the same node.  Returns a description for the result
    line; any failure leaves the affinity alone.  STM_BIND_CORES=0 switches it off."""
    if os.environ.get("STM_BIND_CORES", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "why": "switched off or no sched_setaffinity"}
    try:
        allowed = sorted(os.sched_getaffinity(0))
        lists = []
        for i in range(torch.cuda.device_count()):
            pr = torch.cuda.get_device_properties(i)
            lists.append(gpu_local_cpulist(getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id))
        mine = lists[local_rank] if local_rank < len(lists) else []
        local = [c for c in mine if c in allowed]
        if not local:
            return {"bound": False, "why": "no local_cpulist for this GPU (or none of its cores allowed)", "allowed_cores": len(allowed)}
        peers = [r for r in range(min(world, len(lists))) if lists[r] == mine]          # local ranks on the same NUMA node
        cores = rank_core_slice(local, peers.index(local_rank) if local_rank in peers else 0, max(len(peers), 1))
        os.sched_setaffinity(0, cores)
        torch.set_num_threads(max(1, min(len(cores), 16)))
        return {"bound": True, "cores": f"{cores[0]}-{cores[-1]}" if cores == list(range(cores[0], cores[-1] + 1)) else cores, "n_cores": len(cores),
                "numa_local_cores": len(local), "ranks_on_node": len(peers)}
    except Exception as e:  # never take the run down for a placement hint
        return {"bound": False, "why": repr(e)[:120]}


def emit(line):
    """The result line, on the process's real stdout."""
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        sys.stdout.flush()
        os.write(_RESULT_FD, (line + "\n").encode())

