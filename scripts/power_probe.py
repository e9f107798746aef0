#!/usr/bin/env python3
"""Shader clock and board power WHILE one kernel runs back to back for a few seconds (hwmon files of the device, one sample every 20 ms, on a host
thread; benchlib.extras.BoardSampler), next to the kernel's average duration over that window.  Answers, per kernel: does the board hold its
2.4-GHz clock (then a slow kernel is stalled, not power-limited) or does it sit at the 1.4-kW cap with a reduced clock (then its time is energy)?

phases: idle | copy (torch elementwise over 1 GB) | kx3 (conv_planar_kx3_kernel, 256->256 3x3 at 48x80) | dcn (dcn_fused_kernel, layer2.2: 128 ch,
48x80, stride 1) | dcn256 (layer3.2: 256 ch, 24x40, wide tiles) | tn (TemporalNet-shaped 3x3 512->1024 on 7x7 maps through the planar kernel)
usage: power_probe.py [seconds=2.5] [batch=32] [phases=idle,copy,kx3,dcn,dcn256]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.extras import BoardSampler  # noqa: E402
from stmask_amd import ops, planar  # noqa: E402

SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
PHASES = (sys.argv[3] if len(sys.argv) > 3 else "idle,copy,kx3,dcn,dcn256").split(",")
g = torch.Generator(device="cuda").manual_seed(0)
ops.planar_range_flag()


def dcn_case(C, H, W, s, nsets=4):
    xs = [torch.randn(B * H * W, C, device="cuda", generator=g).clamp_min(0) for _ in range(nsets)]
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    M = B * Ho * Wo
    om = torch.cat([torch.rand(1, 18, device="cuda", generator=g) * 4 - 2 + 0.05 * torch.randn(M, 18, device="cuda", generator=g),
                    torch.randn(M, 9, device="cuda", generator=g), torch.zeros(M, 5, device="cuda")], 1).contiguous()
    w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * (9 * C) ** -0.5
    fz = planar.PlanarConv(w, torch.randn(C, device="cuda", generator=g), s, 1, relu=True, fmt=1)
    state = {"i": 0}

    def run():
        fz.deform(xs[state["i"] % nsets], B, H, W, om, s, 1, 1, has_mask=True)
        state["i"] += 1
    return run, 2.0 * M * C * C * 9 * 3      # issued MFMA flops (three plane products)


def conv_case(C, O, H, W, nsets=2):
    w = torch.randn(O, C, 3, 3, device="cuda", generator=g) * (9 * C) ** -0.5
    pc = planar.PlanarConv(w, torch.randn(O, device="cuda", generator=g), 1, 1, relu=True, fmt=1)
    xs = [ops.split_planes(torch.randn(B, H, W, C, device="cuda", generator=g).clamp_min(0), 1) for _ in range(nsets)]
    state = {"i": 0}

    def run():
        pc(xs[state["i"] % nsets], ("img", B, H, W))
        state["i"] += 1
    return run, 2.0 * B * H * W * C * O * 9 * 3


def copy_case():
    a = torch.randn(256 << 20, device="cuda")       # 1 GiB read + 1 GiB written per launch
    b = torch.empty_like(a)
    return (lambda: torch.mul(a, 1.0001, out=b)), 0.0


def phase(name):
    if name == "idle":
        run, fl = (lambda: None), 0.0
    elif name == "copy":
        run, fl = copy_case()
    elif name == "kx3":
        run, fl = conv_case(256, 256, 48, 80)
    elif name == "dcn":
        run, fl = dcn_case(128, 48, 80, 1)
    elif name == "dcn256":
        run, fl = dcn_case(256, 24, 40, 1)
    elif name == "tn":
        run, fl = conv_case(512, 1024, 7, 7 * 40)
    else:
        raise SystemExit("unknown phase " + name)
    for _ in range(8):
        run()
    torch.cuda.synchronize()
    n = 0
    with BoardSampler(torch.cuda.current_device(), 0.02) as smp:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < SEC:
            if name == "idle":
                time.sleep(0.05)
            else:
                for _ in range(20):
                    run()
                torch.cuda.synchronize()
                n += 20
        dt = time.perf_counter() - t0
    # the first 0.5 s are the ramp (clock rising from idle, power averaging window): report the rest
    s = smp.summary([("steady", 0.5, dt)])
    st = s["steady"]
    us = dt / n * 1e6 if n else 0.0
    print("%-7s %6d launches  %8.1f us/launch  %s  | steady (>0.5 s): sclk %s MHz  power %s W  | %d samples, %s ms apart, source %s" % (
        name, n, us, ("%6.0f TF issued (%.3f of 2500)" % (fl / us / 1e6, fl / us / 1e6 / 2500)) if fl and us else " " * 33,
        st["sclk_mhz"], st["power_w"], s["samples"], s["interval_ms_mean"], s["source"]), flush=True)


print(f"batch {B}, {SEC} s per phase; power cap: ", end="")
try:
    import glob
    print([int(open(f).read()) / 1e6 for f in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_cap")][:1], "W")
except Exception:
    print("?")
for p in PHASES:
    phase(p)
