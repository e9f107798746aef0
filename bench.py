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
                  instances per clip: SURVEY 8(d)'s n ~ 5-10 regime), 8 clips / 1 clip per GPU (each with its own roofline), bf16x3 planes,
                  `sustained` (the headline pipeline for >= 500 steps / >= 10 s with board power and shader clock sampled every 100 ms),
                  `config3` / `config4` / `config5` (one short line each for BASELINE.json's other configurations, with roofline objects)
The roofline object carries `frac_trunk_only` (TemporalNet excluded) and the launches split into MFMA-bound and HBM-bound ones
(`mfma_bound_launches`, `hbm_bound_launches`).  Exit code 3: the parity block failed (matched_frac < 0.98 or mask L2 >= 1e-4).
`--world2-one-gpu`: two ranks of the real model path on one GPU (gloo exchange), compared bit for bit with single-process runs.

The pieces live in benchlib/: launch.py (self-launch, CPU binding of a rank, barriers, the result line), runner.py (model build, Runner), roofline.py
(peaks, PMC traffic file, roofline objects), extras.py (side measurements), checks.py (CPU baseline, parity block, two-ranks-on-one-GPU report).
"""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import launch  # noqa: E402
from benchlib.launch import barrier, emit  # noqa: E402
from benchlib.roofline import BF16_MFMA_PEAK_TF, HBM_PEAK_GBS, PMC_FILE  # noqa: E402,F401
from benchlib.runner import Runner, build_net  # noqa: E402,F401  (tests/test_gpu_parity.py and scripts/gpu_hammer.py drive the benchmark's pipeline through these)

DEFAULT_CLIPS = 32          # plateau of the throughput curve (999 frames/s at 8 clips, 1167 at 16, 1250-1280 at 32, 1300 at 64);
                            # SURVEY §8(d) names 8 clips/GPU: that line and the single-stream (1 clip) line ride along in `extras`
CLIP_FRAMES = 16            # clips of T = 16 frames
DEFAULT_EXTRAS = "realistic,clips8,clips1,bf16x3,per_class_nms,non_tf,e2e,sustained,config3,config4,config5"


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
    ap.add_argument("--extras", default=DEFAULT_EXTRAS, help="which side measurements to run (comma-separated)")
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
                    help="replay the trunk from captured HIP graphs (stmask_amd/pipeline.py _trunk).  auto = on (round 6; until round 5: up to 8 clips per GPU, "
                         "where the ~110 Python-driven launches of the trunk cost more host time than GPU time -- at 32 clips two replayed trunks in flight "
                         "are worth +2 %%); the roofline objects' per-launch HIP events come from an eager pass right after the timed region either way")
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
    ap.add_argument("--share-gpu", action="store_true",
                    help="ranks under a launcher map to device local_rank %% device_count (with --backend gloo: RCCL refuses two ranks on one device): the "
                         "WHOLE N-rank flow of this file -- every pass, barrier, all-gather and the result line -- on a box with fewer GPUs than ranks "
                         "(tests/test_gpu_bench_ranks.py); a plumbing check, not a throughput figure")
    ap.add_argument("--launch-check", action="store_true",
                    help="host-only check of the multi-rank plumbing (self-launch, rendezvous, clip sharding, all-gather, max-over-"
                         "ranks timing, JSON relay) with synthetic detection rows instead of the model; needs no GPU")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    script = os.path.abspath(__file__)
    if args.world2_one_gpu and "RANK" not in os.environ:
        # two ranks of the real model path on whatever GPUs exist (both on device 0 of a 1-GPU box), gloo for the exchange
        args.gpus = 2
        extra = [] if "--backend" in argv else ["--backend", "gloo"]
        argv2 = [a for i, a in enumerate(argv) if not (a == "--gpus" or (i > 0 and argv[i - 1] == "--gpus"))] + ["--gpus", "2"] + extra
        sys.exit(launch.self_launch(args, argv2, script))       # nothing above touched the GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch.self_launch(args, argv, script))        # nothing above touched the GPU

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
        rc = launch.launch_check(args, rank, world)
        if use_dist:
            dist.destroy_process_group()
        sys.exit(rc)

    from benchlib import checks, extras as bx, roofline as rf
    from benchlib.runner import Runner, backbone_tag, heads_tag, image_tag
    from stmask_amd import ops  # noqa: F401  (fails loudly here when the HIP library is missing)
    if args.world2_one_gpu or args.share_gpu:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)     # both ranks on device 0 of a 1-GPU box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a rank under a launcher keeps to the cores of its GPU's NUMA node (its share of them); the plain N = 1 run keeps every core: its CPU baseline
    # leg wants them
    binding = launch.bind_rank_to_gpu_cores(local_rank, world) if (use_dist and world > 1 and not args.world2_one_gpu and not args.share_gpu) else {"bound": False, "why": "single process or shared GPU"}
    if use_dist:
        launch.stdout_to_stderr()
        # NO device_id: with it torch binds the communicator eagerly at start-up, and on this stack that alone -- no collective issued -- costs a rank
        # 1.3-1.5 ms of every 21-ms step (profiles/r05_launcher_overhead.txt: plain 21.19, eager communicator without any gather 22.56, lazy
        # communicator with a gather every step 21.20).  The device is set above; barriers name it.
        dist.init_process_group(args.backend, rank=rank, world_size=world)

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
    tracked_mean = round(run.tracked_sum / max(run.tracked_steps, 1), 1)      # of the timed region (the later passes reset the counters)
    if args.world2_one_gpu:
        sys.exit(checks.world2_report(args, run, dev, rank, world, elapsed, use_dist))
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
    if not events_inside:
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

    # EVERY rank runs the passes below: they contain barriers and the per-step all-gather (a pass on rank 0 alone would wait for ranks that have left)
    fused_t = getattr(run, "fused_t", None) or []
    im2col_in = None
    if fused_t and not timing and not args.no_sampler_pass:
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
        im2col_in = "a short eager pass with STM_DCN_FUSED off right after the timed region (the timed region runs the fused kernel: roofline_dcn_fused)"

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
                       "tracked_instances_mean": tracked_mean,
                       "parallelism": f"clip-dp{world}", "cpu_binding": binding,
                       "hbm_reserved_gb_after_timed_region": round(torch.cuda.memory_reserved(dev) / 1e9, 1),
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
            res["collective"] = {"backend": dist.get_backend(), "world_size": world, "all_gathers_on_comm_stream": g.n_collectives,
                                 "ms_per_step_delta_vs_no_gather": round(coll_delta, 3) if coll_delta is not None else None,
                                 "delta_note": "timed region minus the same K steps run right after it with the exchange switched off (process group kept); "
                                               "profiles/r05_launcher_overhead.txt has the same-box runs against a plain process without any process group",
                                 "comm_stream": (g._comm is not None and g._comm != torch.cuda.default_stream(dev)),
                                 "comm_stream_is_a_trunk_stream": (g._comm is not None and any(g._comm == s_ for s_ in getattr(run.pipe, "_sides", []))),
                                 "gathered_shape": list(out.shape),
                                 "last_gather_equals_local_block": (bool(torch.equal(out[rank * args.clips:(rank + 1) * args.clips], coll_local))
                                                                    if coll_local is not None else None)}
        default_wl = args.clips == DEFAULT_CLIPS and args.config == "STMask_plus_resnet50_config" and args.planes == "fp16x2"
        f_ms = f_fl = 0.0
        if fused_t:
            tr_f, src_f, why_f = rf.pmc_traffic("dcn_fused", len(fused_t) / args.steps) if default_wl else (None, None, "not the profiled workload")
            res["roofline_dcn_fused"], f_ms, f_fl = rf.dcn_fused_roofline(fused_t, args.steps, tr_f, src_f, why_f)
        n_samp_steps = min(args.steps, 8) if im2col_in else args.steps
        tr_s, src_s, why_s = (rf.pmc_traffic("dcn_sample_planar", len(timing) / n_samp_steps) if (default_wl and planar_graph and timing)
                              else (None, None, "not the profiled workload"))
        im2col_roof = rf.im2col_roofline(timing, planar_graph, n_samp_steps, tr_s, src_s, why_s, im2col_in)
        if conv_t:
            tr_c, src_c, why_c = rf.pmc_traffic("conv_planar", len(conv_t) / args.steps) if default_wl else (None, None, "not the profiled workload")
            res["roofline"] = rf.conv_roofline(conv_t, args.steps, args.planes, tr_c, src_c,
                                               "the timed region" if events_inside else "an eager pass of the same K steps right after the timed region (which "
                                               + ("replays HIP graphs" if graphed else "carries no per-launch events: they cost ~0.8 ms per step") + "), next-trunk overlap 'late' so that "
                                               "no two convolution launches share the GPU", traffic_refused=why_c)
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
                rf.print_layer_table(conv_t, args.steps)
        else:
            res["roofline"] = im2col_roof
        if world == 1 and not args.no_extras and run.batched:
            # side measurements on the same process and box (short; the headline above is untouched by them)
            from stmask_amd import planar as _pl
            saved_fmt = (_pl.FMT, _pl.BACKBONE_FMT)
            names = args.extras.split(",")
            headline_ms = elapsed / args.steps * 1e3
            del run.frames_t
            torch.cuda.empty_cache()
            extras = {}
            if "sustained" in names and default_wl:
                try:
                    extras["sustained"] = bx.sustained_block(args, dev, rank, world, net, headline_ms)
                except Exception as e:
                    extras["sustained"] = {"error": repr(e)[:300]}
            extras.update(bx.side_runs(args, dev, rank, world, net, planar_graph))
            if "bf16x3" in extras and "value" in extras["bf16x3"]:
                res["value_bf16x3"] = extras["bf16x3"]["value"]
            _pl.set_format(*saved_fmt)
            if default_wl and planar_graph:
                extras.update(bx.config_lines(args, dev, rank, world, [n for n in names if n.startswith("config")]))
                _pl.set_format(*saved_fmt)
            res["extras"] = extras
        if world == 1 and not args.no_cpu_baseline:
            base, ref_dets = checks.cpu_baseline(args)
            res["cpu_baseline"] = base
            try:
                res["parity"] = checks.parity_block(args, dev, net, ref_dets)
                res["mask_l2"], res["mask_max_abs"] = res["parity"]["mask_l2"], res["parity"]["mask_max_abs"]
            except Exception as e:
                res["parity"] = {"error": repr(e)[:300]}
        emit(json.dumps(res))
        par = res.get("parity")
        if par is not None:
            # the benchmark's own parity block is a gate, not a report: a fast run with wrong results exits non-zero
            bad = ("error" in par) or par["matched_frac"] < checks.PARITY_MIN_MATCHED or not (par["mask_l2"] < checks.PARITY_MAX_MASK_L2)
            if bad:
                sys.stderr.write(f"bench.py: PARITY FAILED against the CPU oracle: {json.dumps(par)[:400]}\n")
                rc_final = 3
    if use_dist:
        barrier()
        dist.destroy_process_group()
    sys.exit(rc_final)


if __name__ == "__main__":
    main()
