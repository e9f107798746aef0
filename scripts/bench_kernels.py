#!/usr/bin/env python3
"""Per-kernel micro-benchmarks at BASELINE shapes (R50-DCN @384x640), interleaved rounds in ONE process, HIP-event
timing on the launch stream.  Prints achieved GB/s (algorithmic bytes) or TFLOP/s next to the gfx950 peaks.

  python scripts/bench_kernels.py [--batch 8] [--what im2col,gemm,corr,lincomb,nms,roi]
"""
import argparse
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stmask_amd import ops  # noqa: E402

DEV = "cuda"
R50_DCN = [  # name, C, Hin, Win, stride
    ("L1.0", 128, 96, 160, 2), ("L1.2", 128, 48, 80, 1), ("L2.0", 256, 48, 80, 2), ("L2.2", 256, 24, 40, 1),
    ("L2.4", 256, 24, 40, 1), ("L3.0", 512, 24, 40, 2), ("L3.2", 512, 12, 20, 1)]


def timeit(fn, rounds=5, iters=10):
    """median over rounds of the mean over iters (ms)"""
    fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters)
    res.sort()
    return res[len(res) // 2]


def bench_im2col(B, variants=(1, 2, 3), layers=R50_DCN, label=""):
    tot = {v: 0.0 for v in variants}
    tot_bytes = 0
    for name, C, H, W, s in layers:
        Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
        x = torch.randn(B, C, H, W, device=DEV)
        om = torch.randn(B, 27, Ho, Wo, device=DEV)
        om[:, :18] = torch.rand(B, 18, Ho, Wo, device=DEV) * 4 - 2   # U(-2,2) like the synthetic DCN offset biases
        cols = torch.empty(B, C * 9, Ho * Wo, device=DEV)
        nbytes = 4 * B * (C * H * W + 27 * Ho * Wo + C * 9 * Ho * Wo)
        tot_bytes += nbytes
        line = f"im2col{label} {name} B={B} C={C} {H}x{W} s{s}: {nbytes / 1e6:8.1f} MB "
        for v in variants:
            ms = timeit(lambda: ops.deform_im2col(x, None, None, 3, s, 1, 1, 1, variant=v, fused_om=om, out=cols))
            tot[v] += ms
            line += f"| v{v} {ms * 1e3:8.1f} us {nbytes / ms / 1e6:7.0f} GB/s "
        print(line, flush=True)
    for v in variants:
        print(f"im2col{label} TOTAL v{v}: {tot[v] * 1e3:.1f} us/batch, {tot_bytes / tot[v] / 1e6:.0f} GB/s "
              f"({tot_bytes / tot[v] / 1e6 / 8000 * 100:.1f}% of 8 TB/s)", flush=True)
    return tot, tot_bytes


def bench_gemm(B):
    for name, C, H, W, s in R50_DCN:
        Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
        M, K, N = C, C * 9, Ho * Wo
        A = torch.randn(M, K, device=DEV) * K ** -0.5
        Bm = torch.randn(B, K, N, device=DEV)
        bias = torch.randn(M, device=DEV)
        fl = 2.0 * M * N * K * B
        line = f"gemm {name} M={M} N={N} K={K} batch={B}: "
        for tile in ("64", "128", "256"):
            os.environ["STM_GEMM_TILE"] = tile
            ms = timeit(lambda: ops.gemm_bias(A, Bm, bias))
            line += f"| tile{tile} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF "
        os.environ.pop("STM_GEMM_TILE", None)
        ms = timeit(lambda: torch.matmul(A, Bm))
        line += f"| torch.matmul {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF"
        print(line, flush=True)


def bench_deform_conv(B):
    tot = 0.0
    for name, C, H, W, s in R50_DCN:
        Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
        x = torch.randn(B, C, H, W, device=DEV)
        om = torch.randn(B, 27, Ho, Wo, device=DEV)
        w = torch.randn(C, C, 3, 3, device=DEV) * 0.02
        bias = torch.zeros(C, device=DEV)
        ms = timeit(lambda: ops.deform_conv(x, None, None, w, bias, s, 1, 1, 1, fused_om=om))
        dense = timeit(lambda: torch.nn.functional.conv2d(x, w, bias, s, 1))
        tot += ms
        print(f"deform_conv {name} B={B}: {ms * 1e3:8.1f} us  (dense MIOpen conv same shape {dense * 1e3:8.1f} us)", flush=True)
    print(f"deform_conv TOTAL: {tot * 1e3:.1f} us per batch of {B}", flush=True)


def bench_corr(B):
    f1, f2 = torch.randn(B, 256, 24, 40, device=DEV), torch.randn(B, 256, 24, 40, device=DEV)
    nbytes = 4 * B * (2 * 256 * 960 + 121 * 960)
    for var in ("0", "1"):
        os.environ["STM_CORR_VARIANT"] = var
        ms = timeit(lambda: ops.corr_patch(f1, f2, 11, 1, 1 / 256, 0.1))
        print(f"corr B={B} variant={'tiled' if var == '0' else 'generic'}: {ms * 1e3:8.1f} us {nbytes / ms / 1e6:7.0f} GB/s "
              f"{2 * 121 * 256 * 960 * B / ms / 1e9:6.2f} TF", flush=True)
    os.environ.pop("STM_CORR_VARIANT", None)


def bench_lincomb(ns=(10, 50, 100, 200)):
    proto = torch.relu(torch.randn(96, 160, 32, device=DEV))
    for n in ns:
        coeff = torch.randn(n, 32, device=DEV)
        c = torch.rand(n, 2, device=DEV)
        wh = torch.rand(n, 2, device=DEV) * 0.5
        box = torch.cat([c - wh / 2, c + wh / 2], 1)
        nbytes = 4 * (96 * 160 * 32 + n * 36 + n * 96 * 160)
        ms = timeit(lambda: ops.lincomb_sigmoid_crop(proto, coeff, box))
        ms2 = timeit(lambda: (torch.sigmoid(proto @ torch.tanh(coeff).t())).permute(2, 0, 1).contiguous())
        print(f"lincomb n={n}: {ms * 1e3:8.1f} us {nbytes / ms / 1e6:7.0f} GB/s   (torch matmul+sigmoid+permute, no crop: {ms2 * 1e3:.1f} us)",
              flush=True)


def bench_nms(Ks=(100, 1000, 4000, 15345), B=1):
    pri = torch.rand(15345, 4, device=DEV)
    for K in Ks:
        conf = torch.softmax(torch.randn(B, K, 41, device=DEV) * 2, -1)
        c = torch.rand(B, K, 2, device=DEV)
        wh = torch.rand(B, K, 2, device=DEV) * 0.2 + 0.01
        boxes = torch.cat([c - wh / 2, c + wh / 2], 2)
        cen = torch.rand(B, K, device=DEV)
        ms = timeit(lambda: ops.cc_fast_nms(conf, boxes, cen, 0.5, 200))
        print(f"cc_fast_nms K={K} batch={B}: {ms * 1e3:8.1f} us", flush=True)
    for Bt in (1, 8):
        loc = torch.randn(Bt, 15345, 4, device=DEV)
        logits = torch.randn(Bt, 15345, 41, device=DEV)
        logits[..., 0] += 5
        conf = torch.softmax(logits, -1)
        cen = torch.rand(Bt, 15345, device=DEV)
        ms = timeit(lambda: ops.detect_cc(loc, pri, conf, cen))
        ms2 = timeit(lambda: ops.generate_candidates(loc, pri, conf))
        print(f"detect_cc fused (N=15345) batch={Bt}: {ms * 1e3:8.1f} us ; generate_candidates: {ms2 * 1e3:8.1f} us", flush=True)


def bench_roi(ns=(10, 50, 150)):
    feat = torch.randn(1, 633, 24, 40, device=DEV)
    for n in ns:
        xy = torch.rand(n, 2, device=DEV) * torch.tensor([30.0, 16.0], device=DEV)
        wh = torch.rand(n, 2, device=DEV) * 8 + 1
        rois = torch.cat([torch.zeros(n, 1, device=DEV), xy, xy + wh], 1)
        ms = timeit(lambda: ops.roi_align(feat, rois, 7))
        print(f"roi_align n={n}: {ms * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--what", default="im2col,gemm,dconv,corr,lincomb,nms,roi")
    a = ap.parse_args()
    what = a.what.split(",")
    print(torch.cuda.get_device_name(0), flush=True)
    if "im2col" in what:
        bench_im2col(a.batch)
        bench_im2col(1, label="(B=1)")
        bench_im2col(32, variants=(2, 3), label="(B=32)")
    if "gemm" in what:
        bench_gemm(a.batch)
    if "dconv" in what:
        bench_deform_conv(a.batch)
    if "corr" in what:
        bench_corr(1)
        bench_corr(a.batch)
    if "lincomb" in what:
        bench_lincomb()
    if "nms" in what:
        bench_nms()
    if "roi" in what:
        bench_roi()
