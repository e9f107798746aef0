#!/usr/bin/env python3
"""Benchmark of the STMask hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frames/sec at 360x640 (tensor 384x640 after /32 padding), R50-DCN-FPN, fp32.
Workload (configs[1]): STMask_plus_resnet50_config (FCA, DCNv2 backbone, temporal fusion as the config has it),
random seeded weights, synthetic clips of T = 16 frames.  A "step" advances every local clip by one
frame: the frames of all local clips go through backbone / FPN / proto-net / heads as one batch, then candidate
generation, Fast NMS, mask lincomb, correlation + RoIAlign + TemporalNet temporal fusion and the tracker.  Inputs are
resident in HBM before the timed region.

N > 1: clips are sharded over ranks (weak scaling, `--clips` per GPU) with one RCCL all-gather of fixed-shape detections
per step.  `python bench.py --gpus N` run as typed (no RANK in the environment) starts the N ranks itself -- the parent,
before it touches the GPU, runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`
as a child process, relays rank 0's JSON line and exits with the child's code.  Under torch.distributed.run (RANK set)
it is one rank.

Extra objects on the JSON line:
  roofline        dominant kernel (conv_planar_kernel), timed live with HIP events on its launch stream inside the timed region
  roofline_im2col deformable sampler (the kernel north_star names), same method, HBM bound
  cpu_baseline    the CPU oracle path ("port": torch-CPU trunk + oracle C kernels) on this box's host cores, rank 0, N=1,
                  on a bounded sample of the same workload (a few frames of one clip)
  parity          masks / boxes of the HIP path against that oracle run on the same clip and weights (outside the timed region)
  extras          N=1 only, short untimed-by-the-driver side measurements: `realistic` (the headline workload capped at 8 tracked
                  instances per clip: SURVEY 8(d)'s n ~ 5-10 regime), 8 clips / 1 clip per GPU (each with its own roofline), bf16x3 planes
The roofline object carries `frac_trunk_only` (TemporalNet excluded) and the launches split into MFMA-bound and HBM-bound ones
(`mfma_bound_launches`, `hbm_bound_launches`).  Exit code 3: the parity block failed (matched_frac < 0.98 or mask L2 >= 1e-4).
`--world2-one-gpu`: two ranks of the real model path on one GPU (gloo exchange), compared bit for bit with single-process runs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
DEFAULT_CLIPS = 32          # plateau of the throughput curve (999 frames/s at 8 clips, 1167 at 16, 1250-1280 at 32, 1300 at 64);
                            # SURVEY §8(d) names 8 clips/GPU: that line and the single-stream (1 clip) line ride along in `extras`
CLIP_FRAMES = 16            # clips of T = 16 frames
PMC_FILE = "r05_pmc_traffic.json"
PARITY_MIN_MATCHED = 0.98   # bench.py exits 3 when its parity block finds fewer of the oracle's instances ...
PARITY_MAX_MASK_L2 = 1e-4   # ... or a soft mask further than north_star's 1e-4 (RMS) from the oracle's


def barrier():
    """dist.barrier() that names this rank's GPU under the RCCL backend: the process group is created WITHOUT device_id (see main), so the first
    collective -- usually this barrier -- is what creates the communicator, and it must not have to guess the device."""
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        dist.barrier(device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--clips", type=int, default=DEFAULT_CLIPS, help="clips per GPU = frames per step per GPU")
    ap.add_argument("--frames", type=int, default=CLIP_FRAMES, help="frames per synthetic clip (the tracker resets every T steps)")
    ap.add_argument("--config", default="STMask_plus_resnet50_config")
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (and the parity block that needs it)")
    ap.add_argument("--no-extras", action="store_true", help="skip the 8-clip / 1-clip / bf16x3 side measurements")
    ap.add_argument("--events-in-timed-region", action="store_true",
                    help="record the per-launch HIP events of the roofline objects INSIDE the timed region (rounds 1-4 did; two event records around each of "
                         "~190 launches cost ~1.1 ms per step at 32 clips, so by default they are taken in an eager pass of the same K steps right after it)")
    ap.add_argument("--no-sampler-pass", action="store_true",
                    help="skip the short eager pass with the fused deformable convolution switched off that only serves roofline_im2col (kernel traces: one kind of step)")
    ap.add_argument("--extras", default="realistic,clips8,clips1,bf16x3,per_class_nms,non_tf,e2e", help="which side measurements to run (comma-separated)")
    ap.add_argument("--nchw", dest="channels_last", action="store_false", help="keep the dense convs in NCHW")
    ap.add_argument("--no-fuse", dest="fuse", action="store_false", help="keep BatchNorm / bias / ReLU as separate kernels")
    ap.add_argument("--overlap", choices=("late", "early", "off"), default="early",
                    help="next frame's trunk on a second stream: 'early' = at the start of the step (default since round 5: +0.5 %% at 32 clips, +5.7 %% at 8, "
                         "+13 %% single-stream; the passes that record per-launch events run 'late'), 'late' = after this frame's temporal-fusion "
                         "convolutions are enqueued (the two big kernel groups never share the GPU: per-kernel timings stay clean), 'off'")
    ap.add_argument("--no-overlap", dest="overlap", action="store_const", const="off", help="same as --overlap off")
    ap.add_argument("--planes", choices=["fp16x2", "bf16x3", "fp16x1"], default="fp16x2",
                    help="operand format of the planar MFMA convolutions: two fp16 planes / 3 products (default, fp32-equivalent), "
                         "three bf16 planes / 6 products (fp32-equivalent, no fp16 range limit), or fp16x1 = ONE fp16 plane / one "
                         "product (BASELINE config 5: genuine fp16 convolutions with fp32 accumulation, ~1e-3 relative)")
    ap.add_argument("--layer-table", action="store_true", help="per-layer-shape timing table of the planar convolution on stderr")
    ap.add_argument("--no-planar", dest="planar", action="store_false",
                    help="dense convolutions through MIOpen instead of the planar matrix-core kernel")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="replay the trunk from captured HIP graphs (stmask_amd/pipeline.py _trunk).  auto = up to 8 clips per GPU, "
                         "where the ~110 Python-driven launches of the trunk cost more host time than GPU time; at 32 clips the step is "
                         "GPU-bound and the eager launches keep the per-kernel HIP-event timing of the roofline inside the timed region")
    ap.add_argument("--pipeline", default="batched", choices=["batched", "per-clip"],
                    help="batched: all clips' post-processing in concatenated tensors; per-clip: reference-shaped layer API")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help='"nccl" is RCCL on ROCm; gloo for the CPU launch test')
    ap.add_argument("--max-instances", type=int, default=0,
                    help="workload knob (SURVEY 8(d)): at most N detections per frame and N tracked instances per clip; 0 = the "
                         "reference's behaviour (its tracker never prunes: ~114 tracked instances per clip with the synthetic weights). "
                         "The default run reports N = 8 as extras.realistic")
    ap.add_argument("--world2-one-gpu", action="store_true",
                    help="run TWO ranks of the real model path on ONE GPU (both map to device 0, gloo all-gather through host staging): "
                         "executes Runner + clip sharding + all-gather + max-over-ranks timing for real where only one GPU exists; "
                         "writes profiles-style JSON with gather_ok and a comparison against the N = 1 run of the same clips")
    ap.add_argument("--launch-check", action="store_true",
                    help="host-only check of the multi-rank plumbing (self-launch, rendezvous, clip sharding, all-gather, max-over-"
                         "ranks timing, JSON relay) with synthetic detection rows instead of the model; needs no GPU")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` without a launcher
def self_launch(args, argv):
    """Parent of an N-rank run.  Touches no GPU API; starts torch.distributed.run as a child and relays its output."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
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


# ---------------------------------------------------------------------------------------------------------------------
def pmc_traffic(kernel):
    """HBM bytes per launch of the named kernel from the committed rocprofv3 PMC passes (bench.py cannot collect PMC counters
    about itself): profiles/r02_pmc_traffic.json, produced by `scripts/gpu_round.sh pmc` on this command and batch."""
    for name in (PMC_FILE, "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                d = json.load(fh)
            return int(d[kernel]["traffic_bytes_per_launch"]), name
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None, None


def conv_roofline(conv_t, steps, planes, traffic=None, traffic_src=None, timed_in="the timed region"):
    """Roofline objects of the dominant kernel family (conv_planar_kernel) from the live HIP-event records of ops.conv_timing:
    (start, end, algorithmic flops, layer key, MFMA products per reference product, role, algorithmic HBM bytes) per launch.

    achieved = fp32-equivalent algorithmic flops (2*M*Cout*Cin*kh*kw of the reference layers, zero-padded channels excluded) /
    launch time; peak = dense 16-bit MFMA peak / n_prod, because each product of the reference is carried by n_prod MFMA products
    (3 fp16x2, 6 bf16x3, 1 fp16x1) -- i.e. frac = the format's MFMA products for the reference's flops / time / 2500 (equal to the issued
    MFMA rate except for TemporalNet's window sets, which skip the products of the padded taps: `mfma_tflops_issued` reports those).
    Beside the overall figure: `frac_trunk_only` (TemporalNet's launches excluded: with the synthetic weights the tracker keeps
    ~114 instances per clip, whose 0.98 GF each are the most efficient launches of the step), and the launches split by what bounds
    each one algorithmically -- a launch whose algorithmic bytes / 8 TB/s exceed its issued flops / 2500 TF is HBM-bound (the
    bottlenecks' 1x1 convolutions with their residual) and is priced in GB/s against the HBM peak, the others against the MFMA peak."""
    def ms(t):
        return t[0].elapsed_time(t[1])

    def mfma_obj(sel):
        c_ms = sum(ms(t) for t in sel)
        c_fl = sum(t[2] for t in sel)
        c_mfma = sum(t[2] * t[4] for t in sel)
        c_issued = sum(t[2] * t[4] * (t[7] if len(t) > 7 else 1.0) for t in sel)      # (window sets skip the taps that lie in the zero padding)
        if not sel or c_ms <= 0 or c_mfma <= 0:
            return None
        tf = c_fl / (c_ms * 1e-3) / 1e12
        peak = BF16_MFMA_PEAK_TF * c_fl / c_mfma
        return {"achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tf / peak, 4),
                "mfma_tflops_issued": round(c_issued / (c_ms * 1e-3) / 1e12, 1),
                # MFMA products actually ISSUED / time / 2500: moves only when the hardware runs faster, never with an accounting change
                "frac_issued": round(c_issued / (c_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4), "launches": len(sel),
                "ms_per_step": round(c_ms / steps, 3), "tflop_per_step": round(c_fl / steps / 1e12, 3)}

    hbm_sel = [t for t in conv_t if t[6] / (HBM_PEAK_GBS * 1e9) > t[2] * t[4] / (BF16_MFMA_PEAK_TF * 1e12)]
    mfma_sel = [t for t in conv_t if not (t[6] / (HBM_PEAK_GBS * 1e9) > t[2] * t[4] / (BF16_MFMA_PEAK_TF * 1e12))]
    trunk_sel = [t for t in conv_t if t[5] != "temporal"]
    allo, trunk, mf = mfma_obj(conv_t), mfma_obj(trunk_sel), mfma_obj(mfma_sel)
    h_ms, h_by = sum(ms(t) for t in hbm_sel), sum(t[6] for t in hbm_sel)
    hbm = None
    if hbm_sel and h_ms > 0:
        gbs = h_by / (h_ms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
               "launches": len(hbm_sel), "ms_per_step": round(h_ms / steps, 3), "gbyte_per_step": round(h_by / steps / 1e9, 3),
               "what": "launches whose algorithmic bytes / 8 TB/s exceed their issued MFMA flops / 2500 TF (bottleneck 1x1 convolutions "
                       "with residual, stem): inputs, residual, outputs and weights once, in their stored formats"}
    obj = dict(allo)
    obj.update({"bound": "mfma",
                "kernel": f"conv_planar_kernel / conv_planar_kx3_kernel / conv_kxr_kernel / conv_chain_kernel ({planes} planes: stem, backbone 1x1/3x3 (the deformable layers: roofline_dcn_fused), "
                          " FPN, proto-net, shared head, TemporalNet; all launches of " + timed_in + ")",
                "traffic": traffic,
                "traffic_source": (f"profiles/{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 "
                                   "correction); average over all launches") if traffic_src else None,
                "peak_note": "algorithmic (reference) flops against 2500 TFLOP/s dense 16-bit MFMA divided by the MFMA products issued per "
                             "reference product (3 for fp16x2 layers, 6 for bf16x3, 1 for fp16x1 layers; flop-weighted over the launches).  TemporalNet's 3x3 layers keep the "
                             "reference's flop count (2 M Cout Cin 9, padded taps included like every layer's) while their border-class windows ISSUE 361 / 441 of "
                             "the products: `mfma_tflops_issued` counts what is issued, `achieved` what the reference computes "
                             "-- frac = MFMA products the format needs for the reference's flops / time / 2500 (fp32 MFMA peak is 157)",
                "avg_launch_us": round(allo["ms_per_step"] * steps * 1e3 / len(conv_t), 2),
                "algorithmic_gflop_per_launch": round(allo["tflop_per_step"] * steps * 1e3 / len(conv_t), 2),
                "timed_in": timed_in,
                "frac_trunk_only": trunk["frac"] if trunk else None,
                "trunk_only": trunk, "mfma_bound_launches": mf, "hbm_bound_launches": hbm})
    return obj


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


def backbone_tag(cfg):
    depth = {(3, 4, 6, 3): "R50", (3, 4, 23, 3): "R101"}.get(tuple(cfg.backbone_layers), "ResNet")
    return depth + ("-DCN" if any(cfg.backbone_dcn_layers) else "") + "-FPN"


def heads_tag(cfg):
    fcb = ("+FCB(ada)" if cfg.use_pred_offset else "+FCB(ali)") if cfg.use_dcn_class else ""
    return "FCA" + fcb + (" + temporal fusion" if cfg.temporal_fusion_module else "")


def image_tag(h, w):
    """Tensor size -> the image size it is the /32 padding of (360x640 -> 384x640, 720x1280 -> 736x1280)."""
    known = {(384, 640): "360x640", (736, 1280): "720x1280"}
    return known.get((h, w), f"{h}x{w}")


def build_net(args, dev, planes=None):
    from stmask_amd import synthetic
    from stmask_amd.config import get_cfg
    from stmask_amd.model import STMask
    planes = planes or args.planes
    net = STMask(get_cfg(args.config))
    net.eval()
    synthetic.fill_state_dict(net, seed=0, bg_bias=synthetic.BENCH_BG_BIAS)
    net = net.to(dev)
    if args.fuse:
        from stmask_amd.fuse import optimize_for_inference
        # BN folded into conv / DCN weights, bias (+residual) + ReLU as one epilogue pass; every dense convolution on
        # stm_conv2d_planar_f32 (split-operand MFMA convolution, all FPN levels per launch)
        optimize_for_inference(net, planar=args.planar and args.channels_last, planes=planes)
    if args.channels_last:
        net = net.to(memory_format=torch.channels_last)
        net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    return net


class Runner:
    """One pipeline over resident synthetic clips; step(t) = every local clip advances one frame + the detection all-gather."""

    def __init__(self, args, dev, rank, world, clips, planes=None, net=None, max_instances=None):
        from stmask_amd import synthetic
        from stmask_amd.pipeline import BatchedClipPipeline, ClipPipeline
        self.args, self.dev, self.clips_n, self.T = args, dev, clips, args.frames
        self.net = net if net is not None else build_net(args, dev, planes)
        # clip c of this rank = global clip rank + c*world (stmask_amd.dist.shard_clips); inputs resident in HBM
        clip_t = torch.stack([synthetic.synthetic_clip(self.T, args.height, args.width, seed=rank + c * world)
                              for c in range(clips)]).to(dev)                      # [clips, T, 3, H, W]
        fmt = torch.channels_last if args.channels_last else torch.contiguous_format
        self.frames_t = [clip_t[:, t].contiguous(memory_format=fmt) for t in range(self.T)]   # in the trunk's layout
        del clip_t
        self.batched = args.pipeline == "batched"
        self.pipe = BatchedClipPipeline(self.net, clips) if self.batched else ClipPipeline(self.net, clips)
        if self.batched:
            self.pipe.max_instances = args.max_instances if max_instances is None else max_instances
            self.pipe.prefetch_early = args.overlap == "early"
            gm = getattr(args, "graph", "auto")
            self.pipe.use_graph = (gm == "on" or (gm == "auto" and clips <= 8)) and args.fuse and args.planar and args.channels_last
        self.tracked_sum = 0.0
        self.tracked_steps = 0
        from stmask_amd.dist import DetectionGatherer
        self.gatherer = DetectionGatherer(dev)
        self.keep = None             # a list: the gathered detections of every step are kept (the two-rank check compares them)

    def step(self, t):
        from stmask_amd import dist as sdist
        T, pipe = self.T, self.pipe
        if self.batched and self.args.overlap != "off":
            # the next frame's trunk starts on a second stream while this frame's tracker logic (tiny launches, two host
            # reads) runs; every step still enqueues exactly one trunk
            # the frames of the next two calls: under graph replay (small batches) two trunks run ahead on two side streams (BatchedClipPipeline._prefetch_trunk)
            out = pipe.step(self.frames_t[t % T], is_first=(t % T == 0), next_frames=[self.frames_t[(t + k) % T] for k in range(1, 1 + max(2, pipe.PREFETCH_DEPTH))])
        else:
            out = pipe.step(self.frames_t[t % T], is_first=(t % T == 0))
        if self.batched:
            self.tracked_sum += sum(pipe.prev_n) / max(self.clips_n, 1)
            self.tracked_steps += 1
        packed = out if self.batched else sdist.pack_detections(out, top_k=self.net.cfg.nms_top_k, device=self.dev)
        # the all-gather rides on its own stream (stmask_amd.dist.DetectionGatherer): neither this step's tail nor the next trunk waits
        full = self.gatherer.gather(packed)
        if self.keep is not None:
            self.keep.append(full)
        return full

    def timed(self, warmup, steps, use_dist=False, collect=False):
        """W untimed steps, then exactly K steps bracketed by barrier + synchronize; returns (seconds, last output, timings)."""
        from stmask_amd import ops
        for t in range(warmup):
            self.step(t)
        t_first = warmup
        if self.batched and self.pipe.use_graph and not collect:
            # the trunk graphs are captured lazily, one slot per trunk call (two eager calls first): keep the captures out of the timed region
            while len(self.pipe._graphs) < self.pipe.N_GRAPH_SLOTS and t_first < warmup + self.pipe.N_GRAPH_SLOTS + 4:
                self.step(t_first)
                t_first += 1
        torch.cuda.synchronize()
        if use_dist:
            barrier()
        if collect:
            ops.im2col_timing(True)
            ops.conv_timing(True)
            ops.fused_dcn_timing(True)
        self.tracked_sum, self.tracked_steps = 0.0, 0
        tm = getattr(self.pipe, "timer", None)
        if tm is not None and tm.on:
            tm.acc.clear()   # diagnosis runs: stage times of the timed steps only
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for t in range(t_first, t_first + steps):
            out = self.step(t)
        self.gatherer.wait()
        torch.cuda.synchronize()
        if use_dist:
            barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timing = ops.im2col_timing(False) if collect else None
        conv_t = (ops.conv_timing(False) or []) if collect else None
        if collect:
            self.fused_t = ops.fused_dcn_timing(False) or []       # launches of the fused deformable convolution (csrc/dcn_fused.hip) of this pass
        return elapsed, out, timing, conv_t


def e2e_block(args, dev, net, cap, steps, warmup=3):
    """SURVEY 8(f) rows f3 + f1 around the step: uint8 720x1280 frames resident in HBM -> stm_preprocess_u8_f32 (resize to the test scale, normalise, pad
    to /32, CHW; eval.py:703-717) -> the step -> keep rule -> stm_mask_resize_rle_f32 (un-pad, bilinear upsample to 720x1280, > 0.5, COCO run lengths;
    output_utils.py:85-106) -> D2H of the run lengths -> RLE strings (library host function).  Wall-clock frames/s of the whole chain and per-stage GPU
    time from HIP events; the next frame's pre-processing is enqueued before the step so that the step can start the next trunk beside its tracker tail."""
    import numpy as np
    from stmask_amd import ops, output_utils, synthetic
    from stmask_amd.pipeline import BatchedClipPipeline
    from stmask_amd.preprocess import MEANS, STD, preprocess_eval_frames
    clips, T = args.clips, args.frames
    img_h = {384: 360, 736: 720}.get(args.height, args.height)
    img_w = args.width
    OH, OW = 720, 1280
    mean = torch.tensor(MEANS).view(1, 3, 1, 1)
    std = torch.tensor(STD).view(1, 3, 1, 1)
    u8 = []
    clip_t = torch.stack([synthetic.synthetic_clip(T, args.height, args.width, seed=c) for c in range(clips)])      # [clips, T, 3, H, W], normalised
    for t in range(T):
        x = clip_t[:, t, :, :img_h, :img_w] * std + mean
        x = x.round().clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1)                                          # [clips, img_h, img_w, 3]
        if (img_h, img_w) != (OH, OW):
            x = x.repeat_interleave(OH // img_h, 1).repeat_interleave(OW // img_w, 2)
        u8.append(x.contiguous().to(dev))
    del clip_t
    fmt = torch.channels_last if args.channels_last else torch.contiguous_format

    def pre(t):
        x, _ = preprocess_eval_frames(u8[t % T], size=(img_w, img_h))
        return x.contiguous(memory_format=fmt)

    pipe = BatchedClipPipeline(net, clips)
    pipe.max_instances = cap or 0
    pipe.prefetch_early = args.overlap == "early"
    thr = net.cfg.eval_conf_thresh
    acc = {"pre": 0.0, "step": 0.0, "keep": 0.0, "rle": 0.0, "host": 0.0}
    n_masks = n_bytes = 0
    tracked_sum = 0.0
    crop_h = crop_w = 0
    x_next = pre(0)
    torch.cuda.synchronize()
    t0 = None
    for t in range(warmup + steps):
        if t == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        x, x_next = x_next, pre(t + 1)
        ev[1].record()
        pipe.step(x, is_first=(t % T == 0), next_frames=x_next if args.overlap != "off" else None)
        ev[2].record()
        prev = pipe.prev
        n = 0
        if prev is not None and sum(pipe.prev_n):
            tm = torch.tensor([v for tr in pipe.tracked for v in tr], device=dev)
            keep = (tm <= 10) & (prev["mask"].gt(0.5).sum([1, 2]) > 1) & (prev["score"] > thr)       # track_TF.py:158-165
            masks = prev["mask"].index_select(0, torch.nonzero(keep).view(-1))
            n = masks.shape[0]
        ev[3].record()
        th = time.perf_counter()
        if n:
            mh, mw = masks.shape[1:]
            crop_h, crop_w = int(img_h / args.height * mh), int(img_w / args.width * mw)
            counts, n_runs = ops.mask_resize_rle(masks, crop_h, crop_w, OH, OW)
            ev[4].record()
            nr = n_runs.cpu()
            host = counts[:, :max(int(nr.max()), 1)].contiguous().cpu()
            strings = output_utils.rle_strings(host, nr)
        else:
            ev[4].record()
            strings = []
        host_s = time.perf_counter() - th
        if t >= warmup:
            torch.cuda.synchronize()
            acc["pre"] += ev[0].elapsed_time(ev[1]); acc["step"] += ev[1].elapsed_time(ev[2]); acc["keep"] += ev[2].elapsed_time(ev[3])
            acc["rle"] += ev[3].elapsed_time(ev[4]); acc["host"] += host_s * 1e3
            n_masks += n
            n_bytes += sum(len(b) for b in strings)
            tracked_sum += sum(pipe.prev_n) / clips
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    per = {k: v / steps for k, v in acc.items()}
    nm = n_masks / steps
    rle_bytes = nm * (crop_h * crop_w * 4 + 2 * OH * OW / 8) if n_masks else 0.0
    return {"value": round(clips * steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3), "clips_per_gpu": clips, "steps": steps,
            "max_instances": cap or None, "frames_in": f"uint8 {OH}x{OW}x3 resident in HBM", "masks_per_step": round(nm, 1),
            "tracked_instances_mean": round(tracked_sum / steps, 1),
            "rle_bytes_per_mask": round(n_bytes / max(n_masks, 1), 1),
            "stages_ms_per_step": {"preprocess_u8 (next frame: resize + normalise + pad + layout)": round(per["pre"], 3),
                                   "step (trunk .. tracker, two host reads)": round(per["step"], 3),
                                   "keep rule + mask gather": round(per["keep"], 3),
                                   "mask_resize_rle kernels (resize_threshold_pack + rle_runs)": round(per["rle"], 3),
                                   "D2H of run lengths + RLE strings (host wall clock, includes the wait for the kernels)": round(per["host"], 3)},
            "preprocess_gbs": round(clips * (OH * OW * 3 + 3 * args.height * args.width * 4 * 3) / (per["pre"] * 1e-3) / 1e9, 1) if per["pre"] > 0 else None,
            "mask_resize_rle_gpixel_s": round(nm * OH * OW / (per["rle"] * 1e-3) / 1e9, 1) if per["rle"] > 0 and nm else None,
            "mask_resize_rle_gbs": round(rle_bytes / (per["rle"] * 1e-3) / 1e9, 1) if per["rle"] > 0 and nm else None,
            "what": "frame bytes -> COCO RLE strings: the reference's FPS meter wraps the same span (eval.py:600-665, output_utils.py:85-106)"}


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


_RESULT_FD = None       # the real stdout of a rank under torch.distributed.run (see main)


def emit(line):
    """The result line, on the process's real stdout."""
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        sys.stdout.flush()
        os.write(_RESULT_FD, (line + "\n").encode())


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.world2_one_gpu and "RANK" not in os.environ:
        # two ranks of the real model path on whatever GPUs exist (both on device 0 of a 1-GPU box), gloo for the exchange
        args.gpus = 2
        extra = [] if "--backend" in argv else ["--backend", "gloo"]
        argv2 = [a for i, a in enumerate(argv) if not (a == "--gpus" or (i > 0 and argv[i - 1] == "--gpus"))] + ["--gpus", "2"] + extra
        sys.exit(self_launch(args, argv2))       # nothing above touched the GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args, argv))        # nothing above touched the GPU

    rc_final = 0
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run

    if args.launch_check:
        if use_dist:
            dist.init_process_group(args.backend if args.backend == "gloo" or torch.cuda.is_available() else "gloo",
                                    rank=rank, world_size=world)
        rc = launch_check(args, rank, world)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(rc)

    from stmask_amd import ops
    if args.world2_one_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)     # both ranks on device 0 of a 1-GPU box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        # RCCL prints a version banner on STDOUT when its communicator comes up (at the first collective); the contract is ONE JSON line on
        # rank 0's stdout.  File descriptor 1 is pointed at stderr for the run, the result line goes to the saved descriptor.
        global _RESULT_FD
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)
        # NO device_id: with it torch binds the communicator eagerly at start-up, and on this stack that alone -- no collective issued -- costs a rank
        # 1.3-1.5 ms of every 21-ms step (profiles/r05_launcher_overhead.txt: plain 21.19, eager communicator without any gather 22.56, lazy
        # communicator with a gather every step 21.20).  The device is set above; barriers name it.  STM_PG_EAGER=1 restores the eager form (A/B).
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=dev if (args.backend == "nccl" and os.environ.get("STM_PG_EAGER", "0") != "0") else None)

    # MIOpen immediate mode (only --no-planar graphs reach the library at all)
    torch.backends.cudnn.benchmark = False
    run = Runner(args, dev, rank, world, args.clips)
    net = run.net
    cfg = net.cfg
    graphed = run.batched and run.pipe.use_graph
    if args.world2_one_gpu:
        run.keep = []
    # The timed region carries no instrumentation: the per-launch HIP events the roofline objects are built from are recorded in a second pass of the same
    # K steps right after it (eager; under a trunk graph that was always so).  --events-in-timed-region puts them back inside.
    events_inside = args.events_in_timed_region and not graphed and not args.world2_one_gpu
    elapsed, out, timing, conv_t = run.timed(args.warmup, args.steps, use_dist, collect=events_inside)
    if args.world2_one_gpu:
        sys.exit(world2_report(args, run, dev, rank, world, elapsed, use_dist))
    coll_local = coll_delta = None
    if use_dist and run.batched:
        # the collective's report, taken HERE: the block this rank packed in the last timed step (before any further pass advances the pipeline), and
        # the same K steps once more with the exchange switched off (the process group stays): what the all-gather costs a step
        run.gatherer.wait()
        coll_local = run.pipe._pack_outputs(dev).clone()
        saved_mode = run.gatherer.mode
        run.gatherer.mode = 2
        el_nog, _, _, _ = run.timed(args.warmup, args.steps, use_dist)         # the same step indices as the timed region: same frames, same tracker phase
        run.gatherer.mode = saved_mode
        coll_delta = (elapsed - el_nog) / args.steps * 1e3
    instrumented_s = None
    if not events_inside and not args.world2_one_gpu:
        # the per-kernel HIP-event timing of the roofline objects: a second, eager pass of the same K steps right after the timed region (same kernels,
        # same shapes, same process; a graph replay cannot be bracketed kernel by kernel at all)
        if graphed:
            run.pipe.use_graph = False
        early = run.batched and run.pipe.prefetch_early
        if early:
            run.pipe.prefetch_early = False      # 'late': the trunk and the temporal-fusion convolutions never share the GPU, so a launch's events bracket that launch alone
        instrumented_s, _, timing, conv_t = run.timed(args.warmup, args.steps, use_dist, collect=True)     # the same step indices: same frames, same tracker phase
        if early:
            run.pipe.prefetch_early = True
        if graphed:
            run.pipe.use_graph = True
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    tm = getattr(run.pipe, "timer", None)
    if tm is not None and tm.on and rank == 0:
        print("stage ms/step:", {k: round(v / args.steps * 1e3, 2) for k, v in tm.acc.items()}, file=sys.stderr, flush=True)
    frames = world * args.clips * args.steps
    n_det = int((out[..., 7] > 0).sum().item())
    if rank == 0:
        planar_graph = args.fuse and args.planar and args.channels_last
        arith = {"fp16x2": "fp32 in / fp32 out / fp32 accumulate; dense convs as 2 fp16 planes x 3 MFMA products (max error 2e-6 of "
                           "sum|x w| vs fp64, tests/test_gpu_conv.py)",
                 "bf16x3": "fp32 in / fp32 out / fp32 accumulate; dense convs as 3 bf16 planes x 6 MFMA products (max error 2e-6 of "
                           "sum|x w| vs fp64, no range limit)",
                 "fp16x1": "fp16 activations and weights in the dense convolutions, fp32 accumulate / bias / residual / "
                           "post-processing (max error 2e-3 of sum|x w| vs fp64, tests/test_gpu_conv.py)"}[args.planes]
        img = image_tag(args.height, args.width)
        res = {
            "metric": f"frames/sec at {img} {backbone_tag(cfg)} (STMask hot path, "
                      + ("fp32" if args.planes != "fp16x1" or not planar_graph else "fp16 convs / fp32 accumulate") + ")",
            "value": round(frames / elapsed, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "timing": {"timed_region": "exactly K steps between barrier + synchronize pairs, " + ("per-launch HIP events recorded inside" if events_inside else
                                       "no instrumentation inside"),
                       "instrumented_pass_ms_per_step": round(instrumented_s / args.steps * 1e3, 3) if instrumented_s is not None else None,
                       "note": "the roofline objects' per-launch durations come from HIP events recorded on the launch stream around every kernel launch; "
                               "recorded inside the timed region (rounds 1-4, --events-in-timed-region) they add ~1.1 ms per step at 32 clips"},
            "vs_baseline": None, "dtype": "f32" if (args.planes != "fp16x1" or not planar_graph) else "f16-convs/f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {backbone_tag(cfg)} {heads_tag(cfg)}, {args.height}x{args.width} tensor "
                                   f"({img} image padded to /32), {args.clips} clips/GPU x 1 frame per step, clips of "
                                   f"T={args.frames} frames, random seeded weights",
                       "clips_per_gpu": args.clips, "frames_per_clip": args.frames, "frames_per_step": world * args.clips,
                       "detections_last_step": n_det,
                       "tracked_instances_mean": round(run.tracked_sum / max(run.tracked_steps, 1), 1),
                       "parallelism": f"clip-dp{world}",
                       "pipeline": args.pipeline + (f"+next-trunk-overlap-{args.overlap}" if (run.batched and args.overlap != "off") else "")
                                   + ("+trunk-hip-graph" if getattr(run.pipe, "graph_active", False) else ""),
                       "inference_graph": ("bn-folded+fused-epilogues" + ("+planar-%s-convs" % args.planes if planar_graph else ""))
                                          if args.fuse else "reference-ops",
                       "arithmetic": arith if planar_graph else "fp32",
                       "memory_format": "channels_last" if args.channels_last else "nchw",
                       "max_instances": args.max_instances or None,
                       "why_32_clips": "SURVEY §8(d) names 8 clips/GPU; throughput plateaus from 32 (extras.clips8 / extras.clips1 "
                                       "carry the 8-clip and single-stream lines)"},
        }
        if use_dist:
            # the data-path collective of SURVEY 8(e) as it ran: backend, how many all-gathers went out on the communication stream, and --
            # at world size 1, where the gathered block must BE the local block -- whether the last one came back bit-equal
            g = run.gatherer
            run.gatherer.wait()
            last_local = coll_local
            res["collective"] = {"backend": dist.get_backend(), "world_size": world, "all_gathers_on_comm_stream": g.n_collectives,
                                 "ms_per_step_delta_vs_no_gather": round(coll_delta, 3) if coll_delta is not None else None,
                                 "delta_note": "timed region minus the same K steps run right after it with the exchange switched off (process group kept); "
                                               "profiles/r05_launcher_overhead.txt has the same-box runs against a plain process without any process group",
                                 "comm_stream": (g._comm is not None and g._comm != torch.cuda.default_stream(dev)),
                                 "gathered_shape": list(out.shape),
                                 "last_gather_equals_local_block": (bool(torch.equal(out[rank * args.clips:(rank + 1) * args.clips], last_local))
                                                                    if last_local is not None else None)}
        ker_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timing)
        ker_bytes = sum(b for _, _, b in timing)
        n_launch = max(len(timing), 1)
        achieved = ker_bytes / (ker_ms * 1e-3) / 1e9 if ker_ms > 0 else 0.0
        default_wl = args.clips == DEFAULT_CLIPS and args.config == "STMask_plus_resnet50_config" and args.planes == "fp16x2"
        tr_s, src_s = pmc_traffic("dcn_sample_planar") if (default_wl and planar_graph) else (None, None)
        im2col_roof = {"bound": "hbm", "kernel": ("dcn_sample_planar_kernel (deformable im2col of the DCN layers, NHWC in, plane columns out)"
                                                   if planar_graph else "deform_im2col_lds (DCN layers)") + ", all launches of the timed region",
                       "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr_s,
                       "traffic_source": f"profiles/{src_s} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)" if src_s else None,
                       "launches": len(timing), "avg_launch_us": round(ker_ms * 1e3 / n_launch, 2),
                       "algorithmic_bytes_per_launch": int(ker_bytes / n_launch)}
        fused_t = getattr(run, "fused_t", None) or []
        if fused_t:
            # the deformable layers as ONE kernel (sampler + plane split + MFMA product, no column buffer): priced both ways -- against the HBM peak
            # with SURVEY 8(d)'s FUSED byte formula (input + offsets + output + weights; the 78 % of a DCN layer's bytes that were columns are gone,
            # so this kernel is nowhere near HBM-bound) and against the matrix peak with the layer's reference flops
            f_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in fused_t)
            f_by, f_fl = sum(t[2] for t in fused_t), sum(t[3] for t in fused_t)
            f_mf = sum(t[3] * t[4] for t in fused_t)
            tr_f, src_f = pmc_traffic("dcn_fused") if default_wl else (None, None)
            res["roofline_dcn_fused"] = {
                "kernel": "dcn_fused_kernel (deformable convolution of the DCN layers: corner gathers, blend, fp16 plane split and the three plane products "
                          "in one kernel; all launches of the pass the other roofline objects come from)",
                "launches": len(fused_t), "avg_launch_us": round(f_ms * 1e3 / len(fused_t), 2), "ms_per_step": round(f_ms / args.steps, 3),
                "bound": "mfma", "achieved": round(f_fl / (f_ms * 1e-3) / 1e12, 1), "peak": round(BF16_MFMA_PEAK_TF * f_fl / f_mf, 1), "unit": "TFLOP/s",
                "frac": round(f_mf / (f_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
                "hbm": {"bound": "hbm", "achieved": round(f_by / (f_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(f_by / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(f_by / len(fused_t)),
                        "formula": "4 B (C H W + 27 Ho Wo) + planes (Cout Ho Wo + 9 C Cout): SURVEY 8(d), fused im2col + GEMM"},
                "traffic": tr_f,
                "traffic_source": f"profiles/{src_f} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)" if src_f else None,
                "replaces": "dcn_sample_planar_kernel (roofline_im2col: 412 MB of columns per launch at 0.48-0.50 of the HBM peak) + the 1x1 product over 9C "
                            "channels; what bounds the fused kernel instead: profiles/r05_dcn_fused_forms.txt"}
            if not timing and not args.no_sampler_pass:
                # the sampler the north star names did not run in that pass (every DCN layer took the fused kernel): its own figure from a short eager
                # pass of the same pipeline with fusion switched off (same process, right after; nothing of it enters the headline)
                from stmask_amd import planar as _plf
                saved_mt, saved_g = _plf.DCN_FUSED_MIN_TILES, (run.pipe.use_graph if run.batched else None)
                _plf.DCN_FUSED_MIN_TILES = 1 << 30
                if run.batched:
                    run.pipe.use_graph = False
                try:
                    early_s = run.batched and run.pipe.prefetch_early
                    if early_s:
                        run.pipe.prefetch_early = False
                    _, _, timing, _ = run.timed(1, min(args.steps, 8), use_dist, collect=True)
                    if early_s:
                        run.pipe.prefetch_early = True
                finally:
                    _plf.DCN_FUSED_MIN_TILES = saved_mt
                    if run.batched:
                        run.pipe.use_graph = saved_g
                ker_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timing)
                ker_bytes = sum(b for _, _, b in timing)
                n_launch = max(len(timing), 1)
                achieved = ker_bytes / (ker_ms * 1e-3) / 1e9 if ker_ms > 0 else 0.0
                im2col_roof.update({"achieved": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBS, 4), "launches": len(timing),
                                    "avg_launch_us": round(ker_ms * 1e3 / n_launch, 2), "algorithmic_bytes_per_launch": int(ker_bytes / n_launch),
                                    "timed_in": "a short eager pass with STM_DCN_FUSED off right after the timed region (the timed region runs the fused kernel: roofline_dcn_fused)"})
        if conv_t:
            tr_c, src_c = pmc_traffic("conv_planar") if default_wl else (None, None)
            res["roofline"] = conv_roofline(conv_t, args.steps, args.planes, tr_c, src_c,
                                            "the timed region" if events_inside else "an eager pass of the same K steps right after the timed region (which "
                                            + ("replays HIP graphs" if graphed else "carries no per-launch events: they cost ~0.8 ms per step") + "), next-trunk overlap 'late' so that "
                                            "no two convolution launches share the GPU")
            res["frac_trunk_only"] = res["roofline"]["frac_trunk_only"]
            if fused_t:
                # `roofline` keeps its definition of the earlier rounds -- the plane-split dense-convolution kernels -- so the deformable layers' products,
                # which those kernels ran until round 4 (0.25 TFLOP per step), left it together with their time; the same figure WITH the fused kernel's
                # launches (whose time also holds the sampling the old sampler kernel did outside any roofline object):
                ro = res["roofline"]
                c_ms, c_fl = ro["ms_per_step"] * args.steps + f_ms, ro["tflop_per_step"] * args.steps * 1e12 + f_fl
                ro["with_dcn_fused"] = {"achieved": round(c_fl / (c_ms * 1e-3) / 1e12, 1), "frac": round(c_fl / (c_ms * 1e-3) / 1e12 / ro["peak"], 4),
                                        "ms_per_step": round(c_ms / args.steps, 3), "tflop_per_step": round(c_fl / args.steps / 1e12, 3)}
            res["roofline_im2col"] = im2col_roof
            if args.layer_table:
                # per-layer-shape table of the dominant kernel (stderr; the JSON line stays alone on stdout)
                agg = {}
                for t in conv_t:
                    a = agg.setdefault(t[3], [0, 0.0, 0.0])
                    a[0] += 1; a[1] += t[0].elapsed_time(t[1]); a[2] += t[2]
                # (tile 0 = conv_kxr_kernel, -1 = conv_chain_kernel: conv2 3x3 + conv3 + shortcut + the next conv1 of a 64-channel bottleneck,
                # -2 = the nine border-class windows of a TemporalNet layer in one conv_planar_kernel grid; TF = reference flops / time)
                print("%9s %5s %5s %2s %2s %2s %4s %6s %9s %8s %7s" % ("M", "C", "O", "k", "s", "g", "tile", "calls", "us/call", "TF", "ms/step"), file=sys.stderr)
                for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                    print("%9d %5d %5d %2d %2d %2d %4d %6d %9.1f %8.1f %7.3f" % (*key, n, ms * 1e3 / n, fl / (ms * 1e-3) / 1e12, ms / args.steps),
                          file=sys.stderr)
        else:
            res["roofline"] = im2col_roof
        if world == 1 and not args.no_extras and run.batched:
            # side measurements on the same process and box (short; the headline above is untouched by them)
            extras = {}
            from stmask_amd import planar as _pl
            saved_fmt = (_pl.FMT, _pl.BACKBONE_FMT)
            del run.frames_t
            torch.cuda.empty_cache()
            REAL_N = 8
            # (single stream first: measured after the two 32- / 8-clip side runs it read 410 instead of 462 frames/s on the same box)
            for name, clips, planes, cap in (("clips1", 1, None, None), ("clips8", 8, None, None), ("realistic", args.clips, None, REAL_N),
                                             ("bf16x3", args.clips, "bf16x3", None)):
                if (planes is None and cap is None and clips == args.clips) or (planes == args.planes) or (planes and not planar_graph):
                    continue
                if name not in args.extras.split(","):
                    continue
                if cap is not None and args.max_instances:
                    continue                                  # the headline itself already runs capped
                try:
                    r2 = Runner(args, dev, rank, world, clips, planes=planes, net=(net if planes is None else None), max_instances=cap)
                    steps = args.steps if clips >= 8 else 3 * args.steps
                    el, _, _, _ = r2.timed(args.warmup, steps)
                    extras[name] = {"value": round(clips * steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3),
                                    "clips_per_gpu": clips, "planes": planes or args.planes, "steps": steps,
                                    "tracked_instances_mean": round(r2.tracked_sum / max(r2.tracked_steps, 1), 1)}
                    if cap is not None:
                        extras[name]["max_instances"] = cap
                        extras[name]["what"] = (f"the headline workload with at most {cap} detections per frame and {cap} tracked instances per "
                                                "clip (SURVEY 8(d): the n ~ 5-10 regime of real YouTube-VIS clips; the reference's tracker never "
                                                "prunes, and the synthetic weights make it keep ~114 per clip: TemporalNet is then 39 % of the "
                                                "step's flops)")
                    if planar_graph and planes is None:
                        # this line's own roofline: per-launch HIP events need eager launches, so a short eager pass of the same
                        # pipeline right after its timed region (which replays HIP graphs up to 8 clips)
                        r2.pipe.use_graph = False
                        r2.pipe.prefetch_early = False
                        rsteps = min(steps, 8)
                        _, _, _, ct = r2.timed(1, rsteps, collect=True)
                        if ct:
                            ro = conv_roofline(ct, rsteps, args.planes, None, None, "an eager pass of the same pipeline right after this line's timed region")
                            extras[name]["roofline"] = {k: ro[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_trunk_only", "launches",
                                                                             "ms_per_step", "timed_in", "mfma_bound_launches", "hbm_bound_launches")}
                    del r2
                    torch.cuda.empty_cache()
                except Exception as e:  # a side measurement never takes the headline down
                    extras[name] = {"error": repr(e)[:200]}
            # rows a13 / a18 through the same batched pipeline: the reference's per-class Fast NMS variant (detection_TF.py:136-204: ONE launch pair
            # for all clips) and the non-temporal-fusion flow Detect + Track (detection.py:98-137, track.py:56-179) on the headline's net and clips
            for name in ("per_class_nms", "non_tf"):
                if name not in args.extras.split(",") or args.max_instances:
                    continue
                try:
                    r2 = Runner(args, dev, rank, world, args.clips, net=net)
                    if name == "per_class_nms":
                        net.Detect_TF.use_cross_class_nms = False
                    else:
                        r2.pipe.tf = False
                    try:
                        el, _, _, _ = r2.timed(args.warmup, args.steps)
                    finally:
                        net.Detect_TF.use_cross_class_nms = True
                    extras[name] = {"value": round(args.clips * args.steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / args.steps * 1e3, 3),
                                    "clips_per_gpu": args.clips, "steps": args.steps, "instances_per_clip_mean": round(r2.tracked_sum / max(r2.tracked_steps, 1), 1),
                                    "what": ("Detect_TF.use_cross_class_nms = False: 40 class-wise Fast NMS per frame, top 100 (stm_fast_nms_batched_f32)"
                                             if name == "per_class_nms" else
                                             "no temporal fusion: Detect + Track (binary-mask tracker, track.py:162 update gate), the frame's detections as output")}
                    del r2
                    torch.cuda.empty_cache()
                except Exception as e:
                    extras[name] = {"error": repr(e)[:200]}
            # frame bytes -> RLE strings around the same pipeline: the realistic regime (8 instances per clip) and the headline's tracked set
            if "e2e" in args.extras.split(",") and not args.max_instances and args.pipeline == "batched":
                try:
                    extras["e2e"] = {"realistic": e2e_block(args, dev, net, 8, max(args.steps // 2, 5)),
                                     "uncapped": e2e_block(args, dev, net, 0, max(args.steps // 2, 5)),
                                     "note": "the frames are the headline's synthetic clips quantised to uint8 (values beyond 0..255 clipped), doubled to "
                                             "720x1280 and zero-padded by the pre-processing: not bit-identical inputs, so the tracked set (and with it "
                                             "TemporalNet's share of the step) differs from the headline's -- tracked_instances_mean says by how much"}
                    torch.cuda.empty_cache()
                except Exception as e:
                    extras["e2e"] = {"error": repr(e)[:300]}
            if "clips1" in extras and "value" in extras["clips1"]:
                extras["clips1"]["context"] = "single-stream regime of the reference's own FPS table (README.md:102: 29.3 FPS on a 2080 Ti, batch 1)"
            if "bf16x3" in extras and "value" in extras["bf16x3"]:
                res["value_bf16x3"] = extras["bf16x3"]["value"]
            _pl.set_format(*saved_fmt)
            res["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            base, ref_dets = cpu_baseline(args)
            res["cpu_baseline"] = base
            try:
                res["parity"] = parity_block(args, dev, net, ref_dets)
                res["mask_l2"], res["mask_max_abs"] = res["parity"]["mask_l2"], res["parity"]["mask_max_abs"]
            except Exception as e:
                res["parity"] = {"error": repr(e)[:300]}
        emit(json.dumps(res))
        par = res.get("parity")
        if par is not None:
            # the benchmark's own parity block is a gate, not a report: a fast run with wrong results exits non-zero
            bad = ("error" in par) or par["matched_frac"] < PARITY_MIN_MATCHED or not (par["mask_l2"] < PARITY_MAX_MASK_L2)
            if bad:
                sys.stderr.write(f"bench.py: PARITY FAILED against the CPU oracle: {json.dumps(par)[:400]}\n")
                rc_final = 3
    if use_dist:
        barrier()
        dist.destroy_process_group()
    sys.exit(rc_final)


if __name__ == "__main__":
    main()
