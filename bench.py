#!/usr/bin/env python3
"""Benchmark of the STMask hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frames/sec at 360x640 (tensor 384x640 after /32 padding), R50-DCN-FPN, fp32.
Workload (configs[1]): STMask_plus_resnet50_config (FCA, DCNv2 backbone, temporal fusion as the config has it),
random seeded weights, synthetic clips.  A "step" advances every local clip by one frame: the frames of all local
clips go through backbone / FPN / proto-net / heads as one batch, then candidate generation, Fast NMS, mask lincomb,
correlation + RoIAlign + TemporalNet temporal fusion and the tracker run per clip.  Inputs are resident in HBM before
the timed region.  N > 1: clips are sharded over ranks (weak scaling, `--clips` per GPU) with one RCCL all-gather of
fixed-shape detections per step.

Extra objects on the JSON line:
  roofline     deformable-im2col kernel, timed live with HIP events on its launch stream inside the timed region;
               achieved = algorithmic bytes / measured time; peak 8 TB/s HBM3E.
  cpu_baseline the CPU oracle path ("port": torch-CPU trunk + oracle C kernels) on this box's host cores, rank 0, N=1,
               on a bounded sample of the same workload (a few frames of one clip).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stmask_amd import dist as sdist  # noqa: E402
from stmask_amd import ops, synthetic  # noqa: E402
from stmask_amd.config import get_cfg  # noqa: E402
from stmask_amd.model import STMask  # noqa: E402
from stmask_amd.pipeline import BatchedClipPipeline, ClipPipeline  # noqa: E402

BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


DEFAULT_CLIPS = 32   # the PMC passes behind profiles/r01_pmc_traffic.json are taken at this batch


def pmc_traffic(kernel=None):
    """HBM bytes per launch (im2col by default, or the named kernel entry) from the committed rocprofv3 PMC passes
    (bench.py cannot collect PMC counters about itself): profiles/r01_pmc_traffic.json, produced by
    `scripts/gpu_round.sh pmc` on the same command and batch."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        return int((d[kernel] if kernel else d)["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(cfg_name, h, w, budget_s=25.0):
    """Oracle path on the host cores: same model / weights / clip, CPU tensors, oracle kernels."""
    import oracle
    from oracle.cpu_path import oracle_ops
    cores = min(len(os.sched_getaffinity(0)), 32)  # more threads than this slow the small convs down
    torch.set_num_threads(cores)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    oracle.set_num_threads(cores)
    net = STMask(get_cfg(cfg_name))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    frames = synthetic.synthetic_clip(4, h, w, seed=0)
    n, t_total = 0, 0.0
    with oracle_ops(), torch.no_grad():
        for t in range(frames.shape[0]):
            t0 = time.perf_counter()
            net(frames[t:t + 1], img_meta=[{"is_first": t == 0, "video_id": 0, "frame_id": t}])
            dt = time.perf_counter() - t0
            if t > 0:  # frame 0 carries one-off costs (prior cache, oneDNN primitive creation)
                n += 1
                t_total += dt
            if t_total > budget_s:
                break
    return {"value": round(n / t_total, 3), "unit": "frames/s", "cores": cores,
            "kind": "port", "sample": f"{n} frames of one {h}x{w} clip after 1 warm-up frame, batch 1, "
            f"torch-CPU trunk + oracle C kernels ({oracle.num_threads()} OpenMP threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--clips", type=int, default=DEFAULT_CLIPS,
                    help="clips per GPU = frames per step per GPU (throughput: 919 frames/s at 8, 1060 at 16, 1100-1120 from 32 up; "
                         "32 clips step in 29 ms, i.e. 32 live 30-fps streams per GPU)")
    ap.add_argument("--config", default="STMask_plus_resnet50_config")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nchw", dest="channels_last", action="store_false", help="keep the dense convs in NCHW")
    ap.add_argument("--no-fuse", dest="fuse", action="store_false", help="keep BatchNorm / bias / ReLU as separate kernels")
    ap.add_argument("--overlap", choices=("late", "early", "off"), default="late",
                    help="next frame's trunk on a second stream: 'late' = after this frame's temporal-fusion convolutions are "
                         "enqueued (default: big kernels never share the GPU, per-kernel timings stay clean), 'early' = at the "
                         "start of the step (about +10 %% frames/s, but kernels of the two streams stretch each other), 'off'")
    ap.add_argument("--no-overlap", dest="overlap", action="store_const", const="off", help="same as --overlap off")
    ap.add_argument("--planes", choices=["fp16x2", "bf16x3"], default="fp16x2",
                    help="operand split of the planar MFMA convolutions: two fp16 planes / 3 products (default) or three bf16 "
                         "planes / 6 products (no fp16 range limit); both are fp32-equivalent to 2e-6 of sum|x w|")
    ap.add_argument("--layer-table", action="store_true", help="per-layer-shape timing table of the planar convolution on stderr")
    ap.add_argument("--no-planar", dest="planar", action="store_false",
                    help="FPN / proto-net / head convolutions through MIOpen instead of the split-operand matrix-core kernel")
    ap.add_argument("--fp16-backbone", action="store_true",
                    help="BASELINE config 5 flavour: ResNet trunk under fp16 autocast (implies --no-fuse); NOT the headline metric")
    ap.add_argument("--pipeline", default="batched", choices=["batched", "per-clip"],
                    help="batched: all clips' post-processing in concatenated tensors; per-clip: reference-shaped layer API")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm

    # MIOpen immediate mode: measured identical steady-state speed to find mode on this trunk (scripts/bench_trunk.py:
    # 20.6 vs 21.0 ms at batch 8) without minutes of solver search per new shape
    torch.backends.cudnn.benchmark = False
    net = STMask(get_cfg(args.config))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    net = net.to(dev)
    if args.fp16_backbone:
        args.fuse = False
        net.backbone_fp16 = True
    if args.fuse:
        from stmask_amd.fuse import optimize_for_inference
        # BN folded into conv / DCN weights, bias (+residual) + ReLU as one epilogue pass; FPN / proto-net / shared head
        # on stm_conv2d_planar_f32 (fp32-equivalent split-operand MFMA convolution, all FPN levels per launch)
        optimize_for_inference(net, planar=args.planar and args.channels_last, planes=args.planes)
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)  # dense convs NHWC (17.0 vs 20.6 ms trunk at batch 8)
        # ... except TemporalNet: on 7x7 RoI tiles MIOpen is 1.5x faster in NCHW (scripts/bench_temporalnet.py)
        net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    T = 8
    # clip c of this rank = global clip rank + c*world (stmask_amd.dist.shard_clips); inputs resident in HBM
    clips = torch.stack([synthetic.synthetic_clip(T, args.height, args.width, seed=rank + c * world)
                         for c in range(args.clips)]).to(dev)  # [clips, T, 3, H, W]
    pipe = BatchedClipPipeline(net, args.clips) if args.pipeline == "batched" else ClipPipeline(net, args.clips)
    if args.pipeline == "batched":
        pipe.prefetch_early = args.overlap == "early"

    fmt = torch.channels_last if args.channels_last else torch.contiguous_format
    frames_t = [clips[:, t].contiguous(memory_format=fmt) for t in range(T)]  # resident, in the trunk's layout

    def step(t):
        if args.pipeline == "batched" and args.overlap != "off":
            # the next frame's trunk starts on a second stream while this frame's tracker logic (tiny launches, two host
            # reads) runs; every step still enqueues exactly one trunk
            out = pipe.step(frames_t[t % T], is_first=(t % T == 0), next_frames=frames_t[(t + 1) % T])
        else:
            out = pipe.step(frames_t[t % T], is_first=(t % T == 0))
        packed = out if args.pipeline == "batched" else sdist.pack_detections(out, top_k=net.cfg.nms_top_k, device=dev)
        return sdist.all_gather_detections(packed)

    for t in range(args.warmup):
        step(t)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    ops.im2col_timing(True)
    ops.conv_timing(True)
    if getattr(pipe, "timer", None) is not None and pipe.timer.on:
        pipe.timer.acc.clear()   # diagnosis runs: stage times of the timed steps only
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        out = step(t)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing = ops.im2col_timing(False)
    conv_t = ops.conv_timing(False) or []
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if getattr(pipe, "timer", None) is not None and pipe.timer.on and rank == 0:
        print("stage ms/step:", {k: round(v / args.steps * 1e3, 2) for k, v in pipe.timer.acc.items()},
              file=sys.stderr, flush=True)
    frames = world * args.clips * args.steps
    n_det = int((out[..., 7] > 0).sum().item())
    if rank == 0:
        ker_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timing)
        ker_bytes = sum(b for _, _, b in timing)
        n_launch = max(len(timing), 1)
        achieved = ker_bytes / (ker_ms * 1e-3) / 1e9 if ker_ms > 0 else 0.0
        res = {
            "metric": "frames/sec at 360x640 R50-DCN-FPN (STMask hot path, fp32)", "value": round(frames / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not args.fp16_backbone else "f16-backbone/f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: R50-DCN-FPN FCA + temporal fusion, {args.height}x{args.width} "
                                   f"tensor (360x640 padded), {args.clips} clips/GPU x 1 frame per step, random seeded weights",
                       "clips_per_gpu": args.clips, "frames_per_step": world * args.clips,
                       "detections_last_step": n_det, "parallelism": f"clip-dp{world}",
                       "pipeline": args.pipeline + (f"+next-trunk-overlap-{args.overlap}" if (args.pipeline == "batched" and args.overlap != "off") else ""),
                       "inference_graph": ("bn-folded+fused-epilogues" + ("+planar-%s-convs" % args.planes if (args.planar and args.channels_last) else ""))
                                          if args.fuse else "reference-ops",
                       "arithmetic": ("fp32 in / fp32 out / fp32 accumulate; dense convs as "
                                      + ("2 fp16 planes x 3 MFMA products" if args.planes == "fp16x2" else "3 bf16 planes x 6 MFMA products")
                                      + " (max error 2e-6 of sum|x w| vs fp64, tests/test_gpu_conv.py)") if (args.fuse and args.planar and args.channels_last)
                                     else "fp32",
                       "memory_format": "channels_last" if args.channels_last else "nchw"},
        }
        planar_dcn = args.fuse and args.planar and args.channels_last
        im2col_roof = {"bound": "hbm", "kernel": ("dcn_sample_planar_kernel (deformable im2col of the 7 DCN layers, NHWC in, plane columns out)"
                                                   if planar_dcn else "deform_im2col_lds (7 DCN layers)") + ", all launches of the timed region",
                       "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 4),
                       "traffic": pmc_traffic("dcn_sample_planar" if planar_dcn else None)
                                  if (args.clips == DEFAULT_CLIPS and args.config == "STMask_plus_resnet50_config") else None,
                       "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)",
                       "launches": len(timing), "avg_launch_us": round(ker_ms * 1e3 / n_launch, 2),
                       "algorithmic_bytes_per_launch": int(ker_bytes / n_launch)}
        if conv_t:
            # dominant kernel of the step: the split-operand convolution.  achieved = fp32-equivalent algorithmic flops
            # (2*M*Cout*Cin*kh*kw of the reference layers, zero-padded channels excluded) / launch time; peak = dense 16-bit
            # MFMA peak / n_prod, because each fp32 product is carried by n_prod MFMA products (3 fp16 or 6 bf16).
            n_prod = 3 if args.planes == "fp16x2" else 6
            split = "fp16x2-plane" if args.planes == "fp16x2" else "bf16x3-plane"
            c_ms = sum(t[0].elapsed_time(t[1]) for t in conv_t)
            c_fl = sum(t[2] for t in conv_t)
            tf = c_fl / (c_ms * 1e-3) / 1e12 if c_ms > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "kernel": f"conv_planar_kernel ({split} split conv: stem, backbone 1x1/3x3 and DCN GEMMs, FPN, proto-net, shared head, TemporalNet; all launches of the timed region)",
                               "achieved": round(tf, 1), "peak": round(BF16_MFMA_PEAK_TF / n_prod, 1), "unit": "TFLOP/s",
                               "frac": round(tf / (BF16_MFMA_PEAK_TF / n_prod), 4),
                               "traffic": pmc_traffic("conv_planar") if (args.clips == DEFAULT_CLIPS and args.config == "STMask_plus_resnet50_config") else None,
                               "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction); average over all launches",
                               "peak_note": f"fp32-equivalent: 2500 TFLOP/s dense 16-bit MFMA / {n_prod} products per fp32 product (fp32 MFMA peak is 157)",
                               "launches": len(conv_t), "avg_launch_us": round(c_ms * 1e3 / len(conv_t), 2),
                               "ms_per_step": round(c_ms / args.steps, 3),
                               "algorithmic_gflop_per_launch": round(c_fl / len(conv_t) / 1e9, 2)}
            res["roofline_im2col"] = im2col_roof
            if args.layer_table:
                # per-layer-shape table of the dominant kernel (stderr; the JSON line stays alone on stdout)
                agg = {}
                for t in conv_t:
                    a = agg.setdefault(t[3], [0, 0.0, 0.0])
                    a[0] += 1; a[1] += t[0].elapsed_time(t[1]); a[2] += t[2]
                print("%9s %5s %5s %2s %2s %2s %4s %6s %9s %8s %7s" % ("M", "C", "O", "k", "s", "g", "tile", "calls", "us/call", "TF", "ms/step"), file=sys.stderr)
                for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    print("%9d %5d %5d %2d %2d %2d %4d %6d %9.1f %8.1f %7.3f" % (*key, n, ms * 1e3 / n, fl / (ms * 1e-3) / 1e12, ms / args.steps),
                          file=sys.stderr)
        else:
            res["roofline"] = im2col_roof
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.config, args.height, args.width)
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
