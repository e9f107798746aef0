"""CPU baseline (the oracle path on the host cores), the parity block against it, the two-ranks-on-one-GPU report.
Only this module of the benchmark imports oracle/ -- as the checker and the reported baseline, never inside a timed GPU region."""
import json
import os
import time

import torch
import torch.distributed as dist

from .launch import barrier, emit
from .runner import Runner

PARITY_MIN_MATCHED = 0.98   # bench.py exits 3 when its parity block finds fewer of the oracle's instances ...
PARITY_MAX_MASK_L2 = 1e-4   # ... or a soft mask further than north_star's 1e-4 (RMS) from the oracle's


def cpu_baseline(args, budget_s=25.0, n_frames=4):
    """Oracle path on the host cores: same model / weights / clip, CPU tensors, oracle kernels.  Returns the baseline object
    and the per-frame detection dicts (the parity block compares the HIP path against them)."""
    import oracle
    from oracle.cpu_path import oracle_ops
    from stmask_amd import synthetic
    from stmask_amd.config import get_cfg
    from stmask_amd.model import STMask
    cores = min(len(os.sched_getaffinity(0)), 32)  # more threads than this slow the small convs down
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    oracle.set_num_threads(cores)
    net = STMask(get_cfg(args.config))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    frames = synthetic.synthetic_clip(args.frames, args.height, args.width, seed=0)[:n_frames]
    n, t_total, dets = 0, 0.0, []
    with oracle_ops(), torch.no_grad():
        for t in range(frames.shape[0]):
            t0 = time.perf_counter()
            out = net(frames[t:t + 1], img_meta=[{"is_first": t == 0, "video_id": 0, "frame_id": t}])
            dt = time.perf_counter() - t0
            dets.append({k: v.clone() for k, v in out[0]["detection"].items() if torch.is_tensor(v)})
            if t > 0:  # frame 0 carries one-off costs (prior cache, oneDNN primitive creation)
                n += 1
                t_total += dt
            if t_total > budget_s:
                break
    base = {"value": round(n / t_total, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} frames of one {args.height}x{args.width} clip after 1 warm-up frame, batch 1, "
                      f"torch-CPU trunk + oracle C kernels ({oracle.num_threads()} OpenMP threads)"}
    return base, dets


def parity_block(args, dev, net, ref_dets):
    """BASELINE.json's "mask L2 vs ref": clip 0 through the HIP path (the benchmark's inference graph, batch 1) against the CPU
    oracle run of the same clip and weights.  Instances are matched by box IoU (> 0.5, same class); reported over all frames:
    matched fraction, max box delta, per-mask RMS L2 and max-abs of the soft masks [n,96,160].  The kernel-level figure
    (`mask_*_same_inputs`: HIP lincomb + crop fed the oracle's own prototypes / coefficients / boxes) is north_star's
    contract; the end-to-end figure adds the fp32 rounding differences of the two trunks."""
    import oracle
    from stmask_amd import ops, synthetic
    from stmask_amd.pipeline import BatchedClipPipeline
    frames = synthetic.synthetic_clip(args.frames, args.height, args.width, seed=0)[:len(ref_dets)].to(dev)
    fmt = torch.channels_last if args.channels_last else torch.contiguous_format
    pipe = BatchedClipPipeline(net, 1)
    n_ref = n_hip = n_match = 0
    box_d = l2 = mx = l2_k = mx_k = 0.0
    for t, ref in enumerate(ref_dets):
        pipe.step(frames[t:t + 1].contiguous(memory_format=fmt), is_first=(t == 0))
        got = pipe.detections()[0]
        gb, rb = got["box"].cpu(), ref["box"]
        n_ref += rb.shape[0]
        n_hip += gb.shape[0]
        if rb.shape[0] == 0 or gb.shape[0] == 0:
            continue
        iou = oracle.jaccard(gb, rb)
        iou = iou * (got["class"].cpu()[:, None] == ref["class"][None, :]).float()
        best, j = iou.max(dim=1)
        sel = torch.nonzero(best > 0.5).view(-1)
        if sel.numel() == 0:
            continue
        n_match += int(sel.numel())
        gm, rm = got["mask"].cpu()[sel], ref["mask"][j[sel]]
        d = gm - rm
        box_d = max(box_d, float((gb[sel] - rb[j[sel]]).abs().max()))
        l2 = max(l2, float(d.pow(2).mean(dim=(1, 2)).sqrt().max()))
        mx = max(mx, float(d.abs().max()))
        # kernel-level: the oracle's own inputs through the HIP lincomb + crop
        km = ops.lincomb_sigmoid_crop(ref["proto"].to(dev), ref["mask_coeff"].to(dev), ref["box"].to(dev), apply_tanh=True).cpu()
        dk = km - ref["mask"]
        l2_k = max(l2_k, float(dk.pow(2).mean(dim=(1, 2)).sqrt().max()))
        mx_k = max(mx_k, float(dk.abs().max()))
    return {"frames": len(ref_dets), "instances_ref": n_ref, "instances_hip": n_hip, "matched": n_match,
            "matched_frac": round(n_match / max(n_ref, 1), 4), "box_max_abs": box_d,
            "mask_l2": l2, "mask_max_abs": mx, "mask_l2_same_inputs": l2_k, "mask_max_abs_same_inputs": mx_k,
            "mask_l2_def": "max over matched instances of sqrt(mean((m_hip - m_ref)^2)) over the 96x160 soft mask",
            "ref": "CPU oracle path (cpu_baseline leg), same clip / weights; arithmetic of the HIP side: " + args.planes}


def world2_report(args, run, dev, rank, world, elapsed, use_dist):
    """--world2-one-gpu: the real model path ran with `world` ranks (clip sharding, per-step all-gather, barrier + max-over-ranks
    timing).  Rank 0 now replays every rank's shard ALONE (no process group in the data path: a fresh Runner over the same global
    clips, same batch shape, same kernels) and compares each step's gathered block with it, row for row."""
    gathered = [g.clone() for g in run.keep]            # per step: [world * clips, top_k, 40], rank-major
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        tmax = tmax.to(dev) if dist.get_backend() == "nccl" else tmax
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        barrier()
    rc = 0
    if rank == 0:
        from stmask_amd import dist as sdist
        pg_backend = dist.get_backend() if use_dist else None
        ok, max_abs, rows, per_rank = True, 0.0, 0, []
        net = run.net
        del run
        torch.cuda.empty_cache()
        for r in range(world):
            solo = Runner(args, dev, r, world, args.clips, net=net)
            solo.gatherer = sdist.DetectionGatherer(dev)
            solo.gatherer.gather = lambda packed: packed          # no exchange: this rank's rows only
            solo.keep = []
            solo.timed(args.warmup, args.steps)
            eq, n_valid = True, 0
            for t, mine in enumerate(solo.keep):
                blk = gathered[t][r * args.clips:(r + 1) * args.clips]
                eq = eq and bool(torch.equal(blk, mine))
                max_abs = max(max_abs, float((blk - mine).abs().max()))
                n_valid += int((mine[..., 7] > 0).sum())
            rows += n_valid
            per_rank.append({"rank": r, "global_clips": [r + c * world for c in range(args.clips)], "bit_equal_to_solo_run": eq,
                             "valid_detection_rows": n_valid})
            ok = ok and eq
            del solo
        frames = world * args.clips * args.steps
        res = {"metric": "two ranks of the model path on one GPU (plumbing check, not a throughput figure)",
               "value": round(frames / elapsed, 2), "unit": "frames/s", "n_gpus": 1, "ranks": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "world2_one_gpu": True, "backend": pg_backend,
               "gather_ok": ok, "max_abs_diff_vs_solo": max_abs, "compared_steps": len(gathered), "valid_rows_compared": rows,
               "per_rank": per_rank,
               "config": {"workload": f"{args.config}, {args.height}x{args.width}, {args.clips} clips per rank x {world} ranks, both ranks on "
                                      f"device 0, T={args.frames}", "clips_per_gpu": args.clips, "parallelism": f"clip-dp{world} on 1 GPU"},
               "what": "Runner + BatchedClipPipeline + clip sharding (clip i -> rank i mod N) + one fixed-shape all-gather per step + "
                       "barrier / max-over-ranks timing executed with 2 processes; every step's gathered block of every rank compared "
                       "bit for bit with a single-process run of that rank's clips"}
        emit(json.dumps(res))
        rc = 0 if ok else 4
    if use_dist:
        barrier()
        dist.destroy_process_group()
    return rc

