#!/usr/bin/env python3
"""profiles/r0N_pmc_traffic.json from the two PMC passes of `scripts/measure_round.sh` (step 6) (gpurun_out/pmc_fetch, pmc_write).
usage: python scripts/make_pmc_json.py gpurun_out/r06 profiles/r06_pmc_traffic.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import load  # noqa: E402


def main():
    out, dst = sys.argv[1], sys.argv[2]
    fetch = load(os.path.join(out, "pmc_fetch"), "FETCH_SIZE")
    write = load(os.path.join(out, "pmc_write"), "WRITE_SIZE")
    try:
        doc = json.load(open(dst))
    except (OSError, ValueError):
        doc = {}
    keep = {k: doc[k] for k in ("kernel", "fetch_bytes_per_launch", "write_bytes_per_launch", "traffic_bytes_per_launch") if k in doc}

    def group(prefix):
        fb = sum(v[0] for k, v in fetch.items() if prefix in k) * 1024 * 2     # KiB -> bytes, gfx950 x2 correction
        nf = sum(v[1] for k, v in fetch.items() if prefix in k)
        wb = sum(v[0] for k, v in write.items() if prefix in k) * 1024
        nw = sum(v[1] for k, v in write.items() if prefix in k)
        if not nf or not nw:
            return None
        return {"launches": nf, "fetch_bytes_per_launch": fb / nf, "write_bytes_per_launch": wb / nw, "traffic_bytes_per_launch": fb / nf + wb / nw}

    doc = dict(keep)
    g = group("dcn_sample_planar_kernel")
    if g:
        doc["dcn_sample_planar"] = {"kernel": "dcn_sample_planar_kernel (7 DCN layers of R50 at batch 32)", **g}
    g = group("dcn_fused_kernel")
    if g:
        doc["dcn_fused"] = {"kernel": "dcn_fused_kernel (the 7 DCN layers of R50 at batch 32 as one kernel each: algorithmic bytes per launch 4 (C H W + 27 Ho Wo) "
                                      "+ planes (Cout Ho Wo + 9 C Cout) = 128 MB on average, no column buffer)", **g}
    g = group("corr_patch_tiled")
    if g:
        doc["corr_patch"] = {"kernel": "corr_patch_tiled<11> (P4 24x40, 256 channels, batch 32: inputs 62.9 MB, output 14.9 MB)", **g}
    def group_any(prefixes):
        fb = sum(v[0] for k, v in fetch.items() if any(p in k for p in prefixes)) * 1024 * 2
        nf = sum(v[1] for k, v in fetch.items() if any(p in k for p in prefixes))
        wb = sum(v[0] for k, v in write.items() if any(p in k for p in prefixes)) * 1024
        nw = sum(v[1] for k, v in write.items() if any(p in k for p in prefixes))
        if not nf or not nw:
            return None
        return {"launches": nf, "fetch_bytes_per_launch": fb / nf, "write_bytes_per_launch": wb / nw, "traffic_bytes_per_launch": fb / nf + wb / nw}

    g = group_any(("conv_planar_kernel", "conv_planar_kx3_kernel", "conv_kxr_kernel", "conv_chain_kernel"))   # (the launches of bench.py's roofline object: pmc_traffic("conv_planar"))
    if g:
        doc["conv_planar"] = {"kernel": "conv_planar_kernel<*> + conv_planar_kx3_kernel + conv_kxr_kernel<*> + conv_chain_kernel<*> (all launches of bench.py at batch 32, fp16x2 plane format; the split-K finishing launches ride inside their layer's events and are not counted as launches)", **g}
    g = group("conv_planar_kx3_kernel")
    if g:
        doc["conv_planar_kx3"] = {"kernel": "conv_planar_kx3_kernel (stride-1 kw = 3 layers on 256-pixel tiles: head towers, proto-net, FPN 3x3; kx-reuse staging)", **g}
    g = group("conv_chain_kernel")
    if g:
        doc["conv_chain"] = {"kernel": "conv_chain_kernel<*> (layer1's three bottlenecks at batch 32: conv2 3x3 -> conv3 + shortcut -> the next block's conv1; "
                                       "algorithmic bytes per launch 4 * M * (64 + 256 + 256 + 64) = 1258 MB, projection form 881 MB)", **g}
    g = group("stem_fused_kernel")
    if g:
        doc["stem_fused"] = {"kernel": "stem_fused_kernel (32 frames 384x640: 94 MB in, 126 MB of planes out)", **g}
    # launches per step of the profiled run, from that run's own result line (the last JSON line of pmc_fetch.log): bench.py refuses the file when a
    # later run's counts disagree (benchlib/roofline.py pmc_traffic)
    line = None
    try:
        for ln in open(os.path.join(out, "pmc_fetch.log"), errors="replace"):
            if ln.lstrip().startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
    except (OSError, ValueError):
        line = None
    if line is not None:
        for grp, key in (("conv_planar", "roofline"), ("dcn_fused", "roofline_dcn_fused"), ("dcn_sample_planar", "roofline_im2col")):
            lps = (line.get(key) or {}).get("launches_per_step")
            if grp in doc and lps is not None:
                doc[grp]["launches_per_step"] = lps
        doc["profiled_run"] = {"steps": line.get("steps"), "warmup": line.get("warmup"), "workload": line.get("config", {}).get("workload"),
                               "value": line.get("value")}
    doc["method"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 4 --warmup 2` (scripts/measure_round.sh, step 6); "
                     "counters are KiB; FETCH_SIZE doubled per the gfx950 correction (MI355X guide, HBM section); WRITE_SIZE exact for 16-byte streaming stores")
    per = {}
    for k in sorted(set(fetch) | set(write)):
        if not any(t in k for t in ("anonymous namespace", "stm_", "conv_planar", "conv_kxr", "conv_chain", "stem_fused", "dcn_", "corr_", "lincomb", "nms", "head_assemble", "roi_align", "mask_", "gather_", "match_", "pack_", "keep_", "shift_")):
            continue
        nf, nw = max(fetch[k][1], 1), max(write[k][1], 1)
        name = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        per[name] = {"launches": fetch[k][1], "fetch_mb": round(fetch[k][0] / nf * 2048 / 1e6, 2), "write_mb": round(write[k][0] / nw * 1024 / 1e6, 2)}
    doc["per_kernel"] = per
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in doc.items() if k != "per_kernel"}, indent=1)[:1500])


if __name__ == "__main__":
    main()
