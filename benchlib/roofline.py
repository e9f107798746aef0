"""Roofline accounting of bench.py: peaks, the committed PMC traffic figures, the objects built from per-launch HIP events."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BF16_MFMA_PEAK_TF = 2500.0  # dense bf16 / fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
PMC_FILE = "r06_pmc_traffic.json"


PMC_LAUNCH_TOLERANCE = 0.06     # relative difference of launches per step (the profiled command runs fewer steps: TemporalNet's launches are absent on a clip's first frame)


def pmc_traffic(kernel, launches_per_step=None, pmc_file=None):
    """HBM bytes per launch of the named kernel group from the committed rocprofv3 PMC passes (bench.py cannot collect PMC counters about itself):
    profiles/r06_pmc_traffic.json, produced by `scripts/measure_round.sh pmc` + scripts/make_pmc_json.py on this command and batch.  The file carries, per
    group, the launches per step of the run it was profiled on (from that run's own result line); a file whose count disagrees with THIS run's --
    another tile rule, another fusion threshold, a kernel that was renamed -- describes other launches and is refused: (None, file, reason).
    Returns (bytes per launch or None, file name or None, reason or None)."""
    name = pmc_file or PMC_FILE
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            d = json.load(fh)
        g = d[kernel]
        traffic = int(g["traffic_bytes_per_launch"])
    except (OSError, KeyError, ValueError, TypeError):
        return None, None, f"profiles/{name} has no group '{kernel}'"
    want = g.get("launches_per_step")
    if want is None:
        return None, name, "the file records no launches_per_step for this group (made before round 6): not checkable against this run, refused"
    if launches_per_step is None or abs(launches_per_step - want) > PMC_LAUNCH_TOLERANCE * max(want, 1e-9):
        return None, name, f"launches per step differ: this run {launches_per_step}, profiled run {want}: the counters describe other launches, refused"
    return traffic, name, None


def conv_roofline(conv_t, steps, planes, traffic=None, traffic_src=None, timed_in="the timed region", traffic_refused=None):
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
                "launches_per_step": round(len(conv_t) / steps, 2),
                "traffic": traffic, "traffic_refused": traffic_refused,
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



def traffic_source(src):
    return f"profiles/{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)" if src else None


def im2col_roofline(timing, planar_graph, steps, traffic=None, src=None, refused=None, timed_in=None):
    """The deformable sampler (the kernel north_star names), HBM bound: algorithmic bytes of SURVEY 8(d) per launch / HIP-event duration."""
    ker_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in timing)
    ker_bytes = sum(b for _, _, b in timing)
    n_launch = max(len(timing), 1)
    achieved = ker_bytes / (ker_ms * 1e-3) / 1e9 if ker_ms > 0 else 0.0
    obj = {"bound": "hbm", "kernel": ("dcn_sample_planar_kernel (deformable im2col of the DCN layers, NHWC in, plane columns out)"
                                      if planar_graph else "deform_im2col_lds (DCN layers)") + ", all launches of the timed region",
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
           "traffic": traffic, "traffic_refused": refused, "traffic_source": traffic_source(src) if traffic is not None else None,
           "launches": len(timing), "launches_per_step": round(len(timing) / max(steps, 1), 2), "avg_launch_us": round(ker_ms * 1e3 / n_launch, 2),
           "algorithmic_bytes_per_launch": int(ker_bytes / n_launch)}
    if timed_in:
        obj["timed_in"] = timed_in
    return obj


def dcn_fused_roofline(fused_t, steps, traffic=None, src=None, refused=None):
    """The deformable layers as ONE kernel (sampler + plane split + MFMA product, no column buffer): priced both ways -- against the HBM peak with SURVEY
    8(d)'s FUSED byte formula (input + offsets + output + weights; the 78 % of a DCN layer's bytes that were columns are gone, so this kernel is nowhere
    near HBM-bound) and against the matrix peak with the layer's reference flops.  Returns (object, total ms, total flops)."""
    f_ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in fused_t)
    f_by, f_fl = sum(t[2] for t in fused_t), sum(t[3] for t in fused_t)
    f_mf = sum(t[3] * t[4] for t in fused_t)
    obj = {"kernel": "dcn_fused_kernel (deformable convolution of the DCN layers: corner gathers, blend, fp16 plane split and the three plane products "
                     "in one kernel; all launches of the pass the other roofline objects come from)",
           "launches": len(fused_t), "launches_per_step": round(len(fused_t) / steps, 2), "avg_launch_us": round(f_ms * 1e3 / len(fused_t), 2),
           "ms_per_step": round(f_ms / steps, 3),
           "bound": "mfma", "achieved": round(f_fl / (f_ms * 1e-3) / 1e12, 1), "peak": round(BF16_MFMA_PEAK_TF * f_fl / f_mf, 1), "unit": "TFLOP/s",
           "frac": round(f_mf / (f_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF, 4),
           "hbm": {"bound": "hbm", "achieved": round(f_by / (f_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(f_by / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(f_by / len(fused_t)),
                   "formula": "4 B (C H W + 27 Ho Wo) + planes (Cout Ho Wo + 9 C Cout): SURVEY 8(d), fused im2col + GEMM"},
           "traffic": traffic, "traffic_refused": refused, "traffic_source": traffic_source(src) if traffic is not None else None,
           "replaces": "dcn_sample_planar_kernel (roofline_im2col: 412 MB of columns per launch at 0.48-0.50 of the HBM peak) + the 1x1 product over 9C "
                       "channels; what bounds the fused kernel instead: profiles/r05_dcn_fused_forms.txt"}
    return obj, f_ms, f_fl


def print_layer_table(conv_t, steps):
    """Per-layer-shape table of the dominant kernel family (stderr; the JSON line stays alone on stdout).  tile 0 = conv_kxr_kernel, -1 =
    conv_chain_kernel (conv2 3x3 + conv3 + shortcut + the next conv1 of a 64-channel bottleneck), -2 = the nine border-class windows of a TemporalNet
    layer in one conv_planar_kernel grid; TF = reference flops / time."""
    import sys
    agg = {}
    for t in conv_t:
        a = agg.setdefault(t[3], [0, 0.0, 0.0])
        a[0] += 1; a[1] += t[0].elapsed_time(t[1]); a[2] += t[2]
    print("%9s %5s %5s %2s %2s %2s %4s %6s %9s %8s %7s" % ("M", "C", "O", "k", "s", "g", "tile", "calls", "us/call", "TF", "ms/step"), file=sys.stderr)
    for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%9d %5d %5d %2d %2d %2d %4d %6d %9.1f %8.1f %7.3f" % (*key, n, ms * 1e3 / n, fl / (ms * 1e-3) / 1e12, ms / steps), file=sys.stderr)
