#!/usr/bin/env python3
"""Steady-state per-step kernel table from a rocprofv3 --kernel-trace CSV of bench.py.
Steps are delimited by the once-per-step `cc_nms_kernel` launch; the first SKIP steps (warm-up: library kernel
selection runs naive reference kernels there) are dropped.  usage: summarize_trace.py <kernel_trace.csv> [skip_steps]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "cc_nms_kernel" in r["Kernel_Name"]]
if len(marks) <= skip + 1:
    sys.exit("not enough steps in the trace")
t0, t1 = marks[skip], marks[-1]
steps = len(marks) - 1 - skip
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]


def cat(name):
    n = name
    if "conv_planar_kx3_kernel" in n: return "ours: conv_planar_kx3_kernel (plane-split dense conv, kx-reuse staging)"
    if "conv_planar_kernel" in n: return "ours: conv_planar_kernel (plane-split dense conv)"
    if "conv_kxr_kernel" in n: return "ours: conv_kxr_kernel (narrow layers, kx-reuse)"
    if "conv_bf16x" in n: return "ours: conv_bf16x (register-staged variant)"
    if "split_planes" in n: return "ours: split_planes_kernel"
    if "gemm128" in n or "gemm_bias_f32" in n or "splitk_reduce" in n: return "ours: fp32 MFMA GEMM (DCN)"
    if "deform_im2col" in n: return "ours: deform_im2col"
    if "softmax_warp" in n: return "other torch"
    if "(anonymous namespace)::" in n and "at::native" not in n: return "ours: " + n.split("(anonymous namespace)::")[1].split("(")[0].split("<")[0]
    if any(k in n for k in ("igemm", "miopen", "Conv", "conv", "Cijk", "ck::", "gemm", "SubTensorOp", "batched_transpose")): return "dense conv / GEMM libraries (MIOpen, CK, rocBLAS)"
    if "copy" in n.lower() or "CatArray" in n: return "copies / cat"
    return "other torch"


tot = defaultdict(float)
cnt = defaultdict(int)
ktot = defaultdict(float)
kcnt = defaultdict(int)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    c = cat(r["Kernel_Name"])
    tot[c] += d
    cnt[c] += 1
    ktot[r["Kernel_Name"]] += d
    kcnt[r["Kernel_Name"]] += 1
busy = sum(tot.values())
wall = (t1 - t0) / 1e6
print(f"steady-state window: {steps} steps, {wall / steps:.3f} ms wall per step, {busy / steps:.3f} ms GPU-busy per step "
      f"({100 * busy / wall:.1f} % busy), {len(sel) / steps:.0f} kernel launches per step\n")
print("| category | launches/step | ms per step | % of busy |\n|---|---|---|---|")
for c, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"| {c} | {cnt[c] / steps:.1f} | {v / steps:.3f} | {100 * v / busy:.1f} |")
print("\n| kernel | launches/step | ms per step | avg us |\n|---|---|---|---|")
for k, v in sorted(ktot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"| `{k[:110]}` | {kcnt[k] / steps:.1f} | {v / steps:.3f} | {1e3 * v / kcnt[k]:.1f} |")
