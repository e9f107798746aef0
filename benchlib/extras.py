"""Side measurements of bench.py (N = 1 only; the headline is untouched by them)."""
import json
import os
import sys
import time

import torch

from .roofline import conv_roofline
from .runner import Runner


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



def side_runs(args, dev, rank, world, net, planar_graph):
    """clips1 / clips8 / realistic / bf16x3 (each a Runner of its own on the same box, with its own roofline from a short eager pass), then rows a13 /
    a18 through the batched pipeline and the frame-bytes -> RLE-strings line.  A side measurement never takes the headline down."""
    extras = {}
    REAL_N = 8
    names = args.extras.split(",")
    # (single stream first: measured after the two 32- / 8-clip side runs it read 410 instead of 462 frames/s on the same box)
    for name, clips, planes, cap in (("clips1", 1, None, None), ("clips8", 8, None, None), ("realistic", args.clips, None, REAL_N),
                                     ("bf16x3", args.clips, "bf16x3", None)):
        if (planes is None and cap is None and clips == args.clips) or (planes == args.planes) or (planes and not planar_graph):
            continue
        if name not in names:
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
                extras[name]["roofline"] = short_roofline(r2, args.planes, min(steps, 8))
            del r2
            torch.cuda.empty_cache()
        except Exception as e:
            extras[name] = {"error": repr(e)[:200]}
    # rows a13 / a18 through the same batched pipeline: the reference's per-class Fast NMS variant (detection_TF.py:136-204: ONE launch pair
    # for all clips) and the non-temporal-fusion flow Detect + Track (detection.py:98-137, track.py:56-179) on the headline's net and clips
    for name in ("per_class_nms", "non_tf"):
        if name not in names or args.max_instances:
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
    if "e2e" in names and not args.max_instances and args.pipeline == "batched":
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
    return extras


ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "frac_issued", "frac_trunk_only", "launches", "ms_per_step", "timed_in", "mfma_bound_launches",
             "hbm_bound_launches")


def short_roofline(r2, planes, rsteps):
    """A side line's own roofline: per-launch HIP events need eager launches, so a short eager pass of the same pipeline right after its timed region
    (which replays HIP graphs up to 8 clips), next-trunk overlap 'late'."""
    r2.pipe.use_graph = False
    r2.pipe.prefetch_early = False
    _, _, _, ct = r2.timed(1, rsteps, collect=True)
    if not ct:
        return None
    ro = conv_roofline(ct, rsteps, planes, None, None, "an eager pass of the same pipeline right after this line's timed region")
    out = {k: ro[k] for k in ROOF_KEYS if k in ro}
    fused_t = getattr(r2, "fused_t", None) or []
    if fused_t:
        f_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in fused_t)
        out["dcn_fused_ms_per_step"] = round(f_ms / rsteps, 3)
        out["dcn_fused_launches"] = len(fused_t)
    return out


# ---------------------------------------------------------------------------------------------------------------------
class BoardSampler:
    """Board power and shader clock of one GPU sampled every `interval` seconds on a host thread while a measurement runs.  Sources, first that works:
    the amdgpu hwmon files of the device (power1_average / power1_input in microwatts, freq1_input in Hz: a file read, ~20 us), then one
    `rocm-smi --showpower --showclocks --json` child per sample (then the cadence is what that takes).  Nothing here touches the HIP runtime."""

    def __init__(self, device_index=0, interval=0.1):
        import threading
        self.interval, self.samples, self._stop = interval, [], threading.Event()
        self.source = None
        self._files = self._find_hwmon(device_index)
        if self._files:
            self.source = ("sysfs hwmon (" + ", ".join(os.path.basename(f) for f in self._files.values()) + ") of "
                           + os.path.dirname(next(iter(self._files.values()))))
        else:
            import shutil
            self._smi = shutil.which("rocm-smi")
            self._dev = device_index
            if self._smi:
                self.source = "rocm-smi --showpower --showclocks --json (one child process per sample)"
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _find_hwmon(device_index):
        import glob
        try:
            import torch as _t
            pr = _t.cuda.get_device_properties(device_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            roots = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:
            roots = []
        if not roots:
            roots = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))[device_index:device_index + 1]
        for r in roots:
            files = {}
            for key, names in (("power_uw", ("power1_average", "power1_input")), ("sclk_hz", ("freq1_input",))):
                for n in names:
                    f = os.path.join(r, n)
                    if BoardSampler._read_int(f) is not None:
                        files[key] = f
                        break
            if "power_uw" in files or "sclk_hz" in files:
                return files
        return {}

    @staticmethod
    def _read_int(path):
        try:
            with open(path) as fh:
                return int(fh.read().strip())
        except (OSError, ValueError):
            return None

    def _read(self):
        if self._files:
            out = {key: self._read_int(f) for key, f in self._files.items()}
            return ((out.get("power_uw") or 0) / 1e6 or None, (out.get("sclk_hz") or 0) / 1e6 or None)
        if getattr(self, "_smi", None):
            import subprocess
            try:
                d = json.loads(subprocess.run([self._smi, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
                c = d.get(f"card{self._dev}", {})
                pw = next((float(v) for k, v in c.items() if "Power" in k and "(W)" in k), None)
                ck = next((float(str(v).strip("()Mhz")) for k, v in c.items() if k.startswith("sclk clock speed")), None)
                return pw, ck
            except Exception:
                return None, None
        return None, None

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            pw, ck = self._read()
            self.samples.append((time.perf_counter() - t0, pw, ck))
            self._stop.wait(self.interval)

    def __enter__(self):
        if self.source:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.source:
            self._thread.join(timeout=10)

    def summary(self, windows=()):
        """windows: [(name, t0, t1)] in seconds since the sampler started."""
        def stat(sel, i):
            v = [s[i] for s in sel if s[i] is not None]
            return None if not v else {"mean": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1)}
        n = len(self.samples)
        dur = self.samples[-1][0] if n else 0.0
        out = {"source": self.source, "samples": n, "interval_ms_mean": round(dur / max(n - 1, 1) * 1e3, 1) if n > 1 else None,
               "power_w": stat(self.samples, 1), "sclk_mhz": stat(self.samples, 2)}
        for name, a, b in windows:
            sel = [s for s in self.samples if a <= s[0] <= b]
            out[name] = {"samples": len(sel), "power_w": stat(sel, 1), "sclk_mhz": stat(sel, 2)}
        return out


def sustained_block(args, dev, rank, world, net, headline_ms, min_seconds=10.0, chunk=100, min_chunks=5, max_chunks=14):
    """The headline pipeline for >= 500 steps and >= 10 s, frames/s per 100-step chunk (a synchronize at each chunk boundary), board power and shader
    clock sampled every 100 ms: the 20-step timed region is 0.4 s, and every "board at its power limit" argument needs a window longer than the
    board's thermal / power-management time constants."""
    r2 = Runner(args, dev, rank, world, args.clips, net=net)
    t = 0
    for _ in range(args.warmup):
        r2.step(t)
        t += 1
    # (the trunk graphs are captured lazily, one slot per trunk call after two eager ones: keep the captures out of the first chunk, as Runner.timed does)
    pipe = r2.pipe
    while r2.batched and pipe.use_graph and len(pipe._graphs) < pipe.n_graph_slots and t < args.warmup + pipe.n_graph_slots + 4:
        r2.step(t)
        t += 1
    torch.cuda.synchronize()
    chunks = []
    with BoardSampler(dev.index or 0, 0.1) as smp:
        t_begin = time.perf_counter()
        while len(chunks) < min_chunks or (time.perf_counter() - t_begin < min_seconds and len(chunks) < max_chunks):
            c0 = time.perf_counter()
            for _ in range(chunk):
                r2.step(t)
                t += 1
            r2.gatherer.wait()
            torch.cuda.synchronize()
            chunks.append((c0 - t_begin, time.perf_counter() - t_begin))
    fps = [round(args.clips * chunk / (b - a), 1) for a, b in chunks]
    total_s = chunks[-1][1] - chunks[0][0]
    ms = total_s / (len(chunks) * chunk) * 1e3
    out = {"steps": len(chunks) * chunk, "seconds": round(total_s, 2), "value": round(args.clips * len(chunks) * chunk / total_s, 2), "unit": "frames/s",
           "ms_per_step": round(ms, 3), "frames_per_s_by_100_steps": fps, "first_100": fps[0], "last_100": fps[-1],
           "vs_headline_timed_region": round(headline_ms / ms, 4) if headline_ms else None,
           "clips_per_gpu": args.clips, "tracked_instances_mean": round(r2.tracked_sum / max(r2.tracked_steps, 1), 1),
           "board": smp.summary([("first_100", chunks[0][0], chunks[0][1]), ("last_100", chunks[-1][0], chunks[-1][1])]),
           "what": "same pipeline, net and clips as the headline (the clips wrap every T frames); one synchronize per 100 steps; vs_headline = headline "
                   "ms_per_step / this ms_per_step"}
    del r2
    torch.cuda.empty_cache()
    return out


CONFIG_LINES = (
    # (key, BASELINE.json configs[i], config name, height, width, planes, clips per GPU)
    ("config3", "R50-DCN-FPN FCA+FCB(ada) + TF correlation (configs[2])", "STMask_plus_resnet50_ada_config", 384, 640, "fp16x2", 32),
    ("config4", "R101-DCN-FPN FCA+FCB(ali)+TF, 360x640: one GPU's leg of the 8-GPU run (configs[3])", "STMask_plus_base_ali_config", 384, 640, "fp16x2", 32),
    ("config5", "R101-DCN-FPN at 720x1280, fp16 MFMA backbone convs: one GPU's leg (configs[4])", "STMask_plus_base_ali_config", 736, 1280, "fp16x1", 8),
)


def config_lines(args, dev, rank, world, which, steps=16, warmup=3):
    """(steps = one whole clip cycle of T = 16 frames: the tracked set, and with it TemporalNet's share, grows over a clip, so a shorter window would depend on
    where in the cycle it falls -- under graph replay the runner puts capture steps in front of the timed region.)
    One short line each for the other BASELINE configurations, in this process (their own nets and clips, built and dropped one at a time), each
    with the roofline objects of a short eager pass.  datasets/config.py:789-798 (ada), :757-766 (R101 ali)."""
    import argparse
    from stmask_amd import planar as _pl
    from .runner import backbone_tag, heads_tag
    out = {}
    for key, what, cfg_name, h, w, planes, clips in CONFIG_LINES:
        if key not in which:
            continue
        saved_fmt = (_pl.FMT, _pl.BACKBONE_FMT)
        try:
            a2 = argparse.Namespace(**vars(args))
            a2.config, a2.height, a2.width, a2.planes, a2.clips, a2.max_instances = cfg_name, h, w, planes, clips, 0
            t0 = time.perf_counter()
            r2 = Runner(a2, dev, rank, world, clips, planes=planes)
            el, _, _, _ = r2.timed(warmup, steps)
            line = {"baseline_config": what, "value": round(clips * steps / el, 2), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3),
                    "clips_per_gpu": clips, "steps": steps, "warmup": warmup, "planes": planes,
                    "dtype": "f32" if planes != "fp16x1" else "f16-convs/f32",
                    "workload": f"{cfg_name}: {backbone_tag(r2.net.cfg)} {heads_tag(r2.net.cfg)}, {h}x{w} tensor, {clips} clips/GPU, T={args.frames}",
                    "tracked_instances_mean": round(r2.tracked_sum / max(r2.tracked_steps, 1), 1)}
            line["roofline"] = short_roofline(r2, planes, min(steps, 6))
            line["wall_s_incl_build"] = round(time.perf_counter() - t0, 1)
            out[key] = line
            del r2
        except Exception as e:
            out[key] = {"baseline_config": what, "error": repr(e)[:300]}
        finally:
            _pl.set_format(*saved_fmt)
            torch.cuda.empty_cache()
    return out
