"""GPU end-to-end tests of the MI355X model: inference-graph variants against the module path and the reference goldens,
the clip pipelines against each other, the drop-in import names.  The tight, unconditional comparisons with the
reference's goldens (head outputs, every matched instance of every clip frame, full-size frames, row a18) live in
tests/test_gpu_parity.py.
"""
import pytest
import torch

from conftest import load_golden
from stmask_amd import synthetic
from stmask_amd.config import get_cfg
from stmask_amd.model import STMask
from test_host_model_cpu import CASES, run_clip

pytestmark = pytest.mark.gpu


def build(name, dev="cuda"):
    net = STMask(get_cfg(name))
    net.eval()
    synthetic.fill_state_dict(net, seed=0)
    return net.to(dev)


# head outputs and clip detections against the reference goldens: tests/test_gpu_parity.py (tight, unconditional)


def test_reference_layer_api_dropins():
    """The import names the reference uses resolve to the MI355X implementations and run."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "stmask_amd", "shims"))
    try:
        for m in ("dcn_v2", "mmcv", "mmcv.ops", "spatial_correlation_sampler"):
            sys.modules.pop(m, None)
        from dcn_v2 import DCN
        from mmcv.ops import DeformConv2d, roi_align
        from spatial_correlation_sampler import spatial_correlation_sample
        import oracle
        dcn = DCN(16, 16, kernel_size=3, stride=2, padding=1, dilation=1, deformable_groups=1).cuda()
        assert set(dict(dcn.named_parameters())) == {"weight", "bias", "conv_offset_mask.weight", "conv_offset_mask.bias"}
        x = torch.randn(2, 16, 12, 20, device="cuda")
        with torch.no_grad():
            dcn.conv_offset_mask.weight.normal_(0, 0.05)
            y = dcn(x)
            om = dcn.conv_offset_mask(x).cpu()
        o1, o2, m = torch.chunk(om, 3, 1)
        ref = oracle.deform_conv(x.cpu(), torch.cat((o1, o2), 1), torch.sigmoid(m), dcn.weight.cpu(), dcn.bias.cpu(), 2, 1)
        assert (y.cpu() - ref).abs().max() < 1e-4
        dc = DeformConv2d(16, 16, kernel_size=(3, 5), padding=(1, 2), deform_groups=1).cuda()
        off = torch.randn(2, 30, 12, 20, device="cuda")
        with torch.no_grad():
            y = dc(x, off)
        assert y.shape == x.shape
        assert (y.cpu() - oracle.deform_conv(x.cpu(), off.cpu(), None, dc.weight.cpu(), None, 1, (1, 2))).abs().max() < 1e-4
        c = spatial_correlation_sample(x, x.roll(1, 3), kernel_size=1, patch_size=11, stride=1, padding=0, dilation_patch=1)
        assert c.shape == (2, 11, 11, 12, 20)
        r = roi_align(x, torch.tensor([[0, 1.0, 1.0, 9.0, 7.0]], device="cuda"), 7)
        assert r.shape == (1, 16, 7, 7)
    finally:
        sys.path.pop(0)


def test_batched_pipeline_equals_per_clip_driver_on_gpu():
    """BatchedClipPipeline vs the reference-shaped per-clip driver, 3 clips x 4 frames at 128x192 on the MI355X."""
    from stmask_amd.dist import unpack_detections
    from stmask_amd.pipeline import BatchedClipPipeline, ClipPipeline
    net = build("STMask_plus_resnet50_config")
    clips = torch.stack([synthetic.synthetic_clip(4, 128, 192, seed=s) for s in (0, 5, 9)]).cuda()
    fast, ref = BatchedClipPipeline(net, 3), ClipPipeline(net, 3)
    for t in range(4):
        packed = fast.step(clips[:, t].contiguous())
        r = ref.step(clips[:, t].contiguous())
        d = fast.detections()
        for b in range(3):
            assert torch.equal(d[b]["box_ids"], r[b]["box_ids"]), (t, b)
            assert torch.equal(d[b]["class"], r[b]["class"])
            assert (d[b]["box"] - r[b]["box"]).abs().max() < 1e-4
            assert (d[b]["mask"] - r[b]["mask"]).abs().max() < 2e-5      # same kernels on the same boxes / coefficients (measured <= 2e-6)
            un = unpack_detections(packed[b])
            assert torch.equal(un["box_ids"], r[b]["box_ids"][:200])


def test_batched_pipeline_per_class_nms_equals_per_clip_driver():
    """Row a13 through the batched pipeline: Detect_TF.use_cross_class_nms = False (detection_TF.py:136-204, the README's mAP* column) -- one
    stm_fast_nms_batched_f32 launch pair for all clips -- against the reference-shaped per-clip driver with the same switch, frame by frame."""
    from stmask_amd.dist import unpack_detections
    from stmask_amd.pipeline import BatchedClipPipeline, ClipPipeline
    net = build("STMask_plus_resnet50_config")
    net.Detect_TF.use_cross_class_nms = False
    clips = torch.stack([synthetic.synthetic_clip(4, 128, 192, seed=s) for s in (0, 5, 9)]).cuda()
    fast, ref = BatchedClipPipeline(net, 3), ClipPipeline(net, 3)
    seen = 0
    for t in range(4):
        packed = fast.step(clips[:, t].contiguous())
        r = ref.step(clips[:, t].contiguous())
        d = fast.detections()
        for b in range(3):
            assert torch.equal(d[b]["box_ids"], r[b]["box_ids"]), (t, b)
            assert torch.equal(d[b]["class"], r[b]["class"])
            assert (d[b]["box"] - r[b]["box"]).abs().max() < 1e-4
            assert (d[b]["mask"] - r[b]["mask"]).abs().max() < 2e-5
            un = unpack_detections(packed[b])
            assert torch.equal(un["box_ids"], r[b]["box_ids"][:200])
            seen += d[b]["box"].shape[0]
    assert seen > 20


def test_batched_pipeline_non_tf_equals_model_forward_and_reference():
    """Row a18 through the batched pipeline: a config without the temporal-fusion module runs Detect + Track (detection.py:98-137,
    track.py:56-179) -- binary masks, the (mask_ious > 0.3).sum() < 2 update gate, the frame's own detections with their object ids -- for all clips
    per launch.  Against the module path (STMask.forward, one clip at a time) on three clips, and against the golden the reference's own
    Detect.detect / Track.track produced (model_r50_fca_nontf.npz) on the golden's clip."""
    from stmask_amd.dist import unpack_detections
    from stmask_amd.pipeline import BatchedClipPipeline
    from test_gpu_parity import build as build_p, match_instances
    net = build_p("STMask_plus_resnet50_config", temporal_fusion=False)
    g = load_golden("model_r50_fca_nontf.npz")
    h, w = [int(v) for v in g["frames_hw"]]
    T = int(g["n_frames"])
    clips = torch.stack([synthetic.synthetic_clip(T, h, w, seed=s) for s in (0, 5, 9)]).cuda()
    refs = [run_clip(net, clips[b], "cuda") for b in range(3)]                      # module path, clip by clip (fresh tracker state per clip: is_first)
    pipe = BatchedClipPipeline(net, 3)
    n_ids = 0
    for t in range(T):
        packed = pipe.step(clips[:, t].contiguous())
        d = pipe.detections()
        for b in range(3):
            r = refs[b][t]
            if r["box"].shape[0] == 0:
                assert not d[b] or d[b]["box"].shape[0] == 0
                continue
            assert torch.equal(d[b]["box_ids"], r["box_ids"]), (t, b)
            assert torch.equal(d[b]["class"], r["class"])
            assert (d[b]["box"] - r["box"]).abs().max() < 1e-5 and (d[b]["score"] - r["score"]).abs().max() < 1e-6
            assert torch.equal(d[b]["mask"], r["mask"])                              # binary masks of the same kernels on the same inputs
            un = unpack_detections(packed[b])
            assert torch.equal(un["box_ids"], r["box_ids"][:200]) and (un["box"] - r["box"][:200]).abs().max() < 1e-5
            n_ids += r["box"].shape[0]
        # clip 0 is the golden's clip: the reference's own after-NMS sets and ids
        ref_box, ref_cls, ref_ids = g[f"t{t}_box"], g[f"t{t}_class"], g[f"t{t}_box_ids"]
        gb, gc = d[0]["box"].cpu(), d[0]["class"].cpu()
        gi, ri = match_instances(gb, gc, ref_box, ref_cls)
        assert len(gi) >= 0.9 * ref_box.shape[0]
        if gb.shape[0] == ref_box.shape[0] == len(gi):
            assert d[0]["box_ids"].cpu()[gi].tolist() == ref_ids[ri].tolist(), t
    assert n_ids > 20


@pytest.mark.parametrize("channels_last", [False, True])
def test_optimized_inference_graph_on_gpu(channels_last):
    """fuse.optimize_for_inference (BN folded, one-pass conv epilogues, ReLU in the DCN GEMM) changes head outputs only
    by fp32 rounding, in NCHW and channels_last."""
    from stmask_amd.fuse import optimize_for_inference
    ref_net = build("STMask_plus_resnet50_config")
    opt_net = build("STMask_plus_resnet50_config")
    optimize_for_inference(opt_net)
    x = synthetic.synthetic_clip(2, 128, 192, seed=3).cuda()
    if channels_last:
        opt_net = opt_net.to(memory_format=torch.channels_last)
        opt_net.TemporalNet = opt_net.TemporalNet.to(memory_format=torch.contiguous_format)
    with torch.no_grad():
        _, a = ref_net.forward_single(x)
        _, b = opt_net.forward_single(x.contiguous(memory_format=torch.channels_last) if channels_last else x)
    for k in ("loc", "conf", "mask_coeff", "centerness", "proto", "track"):
        scale = max(1.0, a[k].abs().max().item())
        assert (a[k] - b[k]).abs().max().item() < 2e-4 * scale, k


@pytest.mark.parametrize("planes", ["fp16x2", "bf16x3"])
@pytest.mark.parametrize("name,tag", CASES[:3])
def test_planar_graph_matches_module_path_and_reference(name, tag, planes):
    """fuse.optimize_for_inference(planar=True): FPN pred/downsample layers, proto-net and (FCA-only configs) the whole
    shared head on the bf16-split matrix-core convolution, all five levels per launch.  Same tensors as the module
    path to fp32 rounding (5e-5 relative, the convolutions themselves are at 2e-6), the reference goldens to the
    tolerance the module path is held to, and the clip's detections as before."""
    from stmask_amd.fuse import optimize_for_inference
    g = load_golden(f"model_{tag}.npz")
    h, w = [int(v) for v in g["frames_hw"]]
    ref_net = build(name)
    opt_net = build(name)
    optimize_for_inference(opt_net, planar=True, planes=planes)
    opt_net = opt_net.to(memory_format=torch.channels_last)
    opt_net.TemporalNet = opt_net.TemporalNet.to(memory_format=torch.contiguous_format)
    assert opt_net._planar.fmt == (1 if planes == "fp16x2" else 0)
    assert opt_net._planar.head_planar and opt_net._planar.fcb == (tag != "r50_fca")
    frames = synthetic.synthetic_clip(int(g["n_frames"]), h, w, seed=0)
    x = frames[:2].cuda()
    with torch.no_grad():
        fa, a = ref_net.forward_single(x)
        fb, b = opt_net.forward_single(x.contiguous(memory_format=torch.channels_last))
    assert torch.equal(a["priors"], b["priors"])
    for k in ("loc", "conf", "mask_coeff", "centerness", "proto", "track"):
        assert a[k].shape == b[k].shape, k
        scale = max(1.0, a[k].abs().max().item())
        assert (a[k] - b[k]).abs().max().item() < 5e-5 * scale, (k, (a[k] - b[k]).abs().max().item())
    ci = opt_net.correlation_selected_layer
    assert (fa[ci] - fb[ci]).abs().max().item() < 5e-5 * max(1.0, fa[ci].abs().max().item())
    assert (a["T2S_feat"][ci] - b["T2S_feat"][ci]).abs().max().item() < 5e-5 * max(1.0, a["T2S_feat"][ci].abs().max().item())
    for k, gk in [("loc", "f0_loc"), ("conf", "f0_conf_logits"), ("mask_coeff", "f0_mask_coeff"),
                  ("centerness", "f0_centerness"), ("proto", "f0_proto")]:
        ref = g[gk]
        assert (b[k][0].cpu() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item()), k
    from test_gpu_parity import check_frame
    outs = run_clip(opt_net, frames.contiguous(memory_format=torch.channels_last), "cuda")
    for t, det in enumerate(outs):
        check_frame((tag, planes, t), det, g[f"t{t}_box"], g[f"t{t}_class"], g[f"t{t}_mask"], min_frac=0.98, tol_box=3e-6,
                    tol_rms=1e-4, tol_abs=2e-4)


@pytest.mark.parametrize("name,tag", CASES[:3])
def test_planar_graph_with_fused_deformable_layers_matches_reference(name, tag, monkeypatch):
    """The same graph with every deformable convolution on the fused kernel (csrc/dcn_fused.hip: the DCN layers of the backbone and -- FCB
    configs -- FeatureAlign's DeformConv2d on every FPN level; at the test's image size the small-grid rule would keep the sampler + product
    pair, so the rule is switched off): module path, reference goldens and the clip's detections to the tolerances of the test above."""
    from stmask_amd import planar, _lib
    from stmask_amd.fuse import optimize_for_inference
    monkeypatch.setattr(planar, "DCN_FUSED", True)
    monkeypatch.setattr(planar, "FCB_FUSED", True)
    monkeypatch.setattr(planar, "DCN_FUSED_MIN_TILES", 1)
    g = load_golden(f"model_{tag}.npz")
    h, w = [int(v) for v in g["frames_hw"]]
    ref_net = build(name)
    opt_net = build(name)
    optimize_for_inference(opt_net, planar=True, planes="fp16x2")
    opt_net = opt_net.to(memory_format=torch.channels_last)
    opt_net.TemporalNet = opt_net.TemporalNet.to(memory_format=torch.contiguous_format)
    frames = synthetic.synthetic_clip(int(g["n_frames"]), h, w, seed=0)
    x = frames[:2].cuda()
    n0 = _lib.lib().stm_debug_launch_count(1)
    with torch.no_grad():
        fa, a = ref_net.forward_single(x)
        fb, b = opt_net.forward_single(x.contiguous(memory_format=torch.channels_last))
    launched = _lib.lib().stm_debug_launch_count(1) - n0
    assert launched >= 7 + (0 if tag == "r50_fca" else 15), launched          # 7 DCN layers (+ 3 kernel shapes x 5 levels of the class branch)
    for k in ("loc", "conf", "mask_coeff", "centerness", "proto", "track"):
        scale = max(1.0, a[k].abs().max().item())
        assert (a[k] - b[k]).abs().max().item() < 5e-5 * scale, (k, (a[k] - b[k]).abs().max().item())
    for k, gk in [("loc", "f0_loc"), ("conf", "f0_conf_logits"), ("mask_coeff", "f0_mask_coeff"),
                  ("centerness", "f0_centerness"), ("proto", "f0_proto")]:
        ref = g[gk]
        assert (b[k][0].cpu() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item()), k
    from test_gpu_parity import check_frame
    outs = run_clip(opt_net, frames.contiguous(memory_format=torch.channels_last), "cuda")
    for t, det in enumerate(outs):
        check_frame((tag, "fused", t), det, g[f"t{t}_box"], g[f"t{t}_class"], g[f"t{t}_mask"], min_frac=0.98, tol_box=3e-6,
                    tol_rms=1e-4, tol_abs=2e-4)


@pytest.mark.parametrize("fmt", [1, 0])
def test_planar_temporalnet_matches_module(fmt):
    """PlanarTemporalNet (633 -> 640 zero-padded channels, any RoI count per launch) == the nn.Module TemporalNet."""
    from stmask_amd import planar
    from stmask_amd.planar import PlanarTemporalNet
    net = build("STMask_plus_resnet50_config")
    planar.set_format(fmt)
    ptn = PlanarTemporalNet(net.TemporalNet)
    for n in (1, 37, 200):
        x = torch.relu(torch.randn(n, 633, 7, 7, generator=torch.Generator().manual_seed(n))).cuda()
        with torch.no_grad():
            la, ca = net.TemporalNet(x)
            lb, cb = ptn(x)
        assert (la - lb).abs().max().item() < 2e-5 * max(1.0, la.abs().max().item())
        assert (ca - cb).abs().max().item() < 2e-5 * max(1.0, ca.abs().max().item())


def test_video_demo_end_to_end_flow():
    """uint8 frames -> device pre-processing (f3) -> hot path -> postprocess + RLE (f1) -> YouTube-VIS records (f2): the
    stages compose, every record is well-formed and its RLEs decode to masks of the original frame size."""
    import importlib.util
    import os
    import oracle
    spec = importlib.util.spec_from_file_location("run_video_demo", os.path.join(os.path.dirname(__file__), "..", "scripts",
                                                                                 "run_video_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    recs = demo.run(n_clips=2, n_frames=3)
    assert len(recs) > 0 and {r["video_id"] for r in recs} <= {0, 1}
    n_seg = 0
    for r in recs:
        assert 0.0 < r["score"] <= 1.0 and 1 <= r["category_id"] <= 40 and len(r["segmentations"]) == 3
        for sgm in r["segmentations"]:
            if sgm is None:
                continue
            assert sgm["size"] == [720, 1280] and isinstance(sgm["counts"], str)
            counts = oracle.rle_from_string(sgm["counts"].encode())
            assert int(counts.sum()) == 720 * 1280
            n_seg += 1
    assert n_seg > 0


def test_batched_pipeline_trunk_overlap_is_transparent():
    """step(..., next_frames=) runs the next frame's trunk on a second stream during the tracker stage; the packed
    detections of every step equal the serial schedule's to the run-to-run noise of the pipeline itself (two serial runs
    differ by ~2e-6: the dense-conv library's lateral 1x1 convolutions use a split-K kernel with atomic adds)."""
    from stmask_amd.pipeline import BatchedClipPipeline
    from stmask_amd.fuse import optimize_for_inference
    net = build("STMask_plus_resnet50_config")
    optimize_for_inference(net, planar=True)
    net = net.to(memory_format=torch.channels_last)
    net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    T = 5
    clips = torch.stack([synthetic.synthetic_clip(T, 128, 192, seed=s) for s in (0, 5, 9)]).cuda()
    frames = [clips[:, t].contiguous(memory_format=torch.channels_last) for t in range(T)]
    a, b = BatchedClipPipeline(net, 3), BatchedClipPipeline(net, 3)
    for t in range(T):
        ya = a.step(frames[t], is_first=(t == 0)).clone()
        yb = b.step(frames[t], is_first=(t == 0), next_frames=frames[t + 1] if t + 1 < T else None).clone()
        torch.cuda.synchronize()
        assert ya.shape == yb.shape and (ya - yb).abs().max().item() < 1e-4, t
    # a caller that changes its mind about the next frame still gets the right answer
    c = BatchedClipPipeline(net, 3)
    y0 = c.step(frames[0], is_first=True, next_frames=frames[3])
    y1 = c.step(frames[1], is_first=False)
    d = BatchedClipPipeline(net, 3)
    d.step(frames[0], is_first=True)
    assert (y1 - d.step(frames[1], is_first=False)).abs().max().item() < 1e-4


def test_fp16_plane_graph_out_of_range_falls_back_to_bf16x3():
    """The fp16x2 graph cannot carry |activation| > 65504.  Such a step must never return detections computed from inf / nan (a ReLU epilogue
    would turn them into zeros: plausible-looking wrong results) -- and it must not kill the run either: the pipeline puts back the tracker rows the
    step had already shifted, rebuilds the graph with bf16x3 planes in-process, repeats the step and stays there.  Checked on a clip whose THIRD
    frame leaves the range (so the repeated step has a tracked set to restore) against a pipeline that ran bf16x3 from the start, and on a net with
    one layer's weights scaled to overflow; range_fallback = False keeps the loud failure."""
    from stmask_amd.pipeline import BatchedClipPipeline
    from stmask_amd.fuse import optimize_for_inference
    from stmask_amd.ops import StmError

    def make(planes, scale_layer=False):
        net = build("STMask_plus_resnet50_config")
        if scale_layer:
            with torch.no_grad():
                net.fpn.pred_layers[0].weight.mul_(3e5)              # P3's prediction conv: its outputs leave fp16's range, nothing upstream does
        optimize_for_inference(net, planar=True, planes=planes)
        net = net.to(memory_format=torch.channels_last)
        net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
        return net, BatchedClipPipeline(net, 2)

    clip = torch.stack([synthetic.synthetic_clip(4, 128, 192, seed=s) for s in (2, 7)]).cuda()
    clip[:, 2] *= 1e5                                                 # frame 2 of both clips: |stem output| >> 65504
    frames = [clip[:, t].contiguous(memory_format=torch.channels_last) for t in range(4)]
    (net_a, a), (net_b, b) = make("fp16x2"), make("bf16x3")
    seen = 0
    for t in range(4):
        pa, pb = a.step(frames[t], is_first=(t == 0)), b.step(frames[t], is_first=(t == 0))
        assert a.fell_back == (t >= 2) and not b.fell_back
        da, db = a.detections(), b.detections()
        for c in range(2):
            assert torch.equal(da[c]["box_ids"], db[c]["box_ids"]) and torch.equal(da[c]["class"], db[c]["class"]), (t, c)
            if da[c]["box"].numel():
                assert (da[c]["box"] - db[c]["box"]).abs().max() < 1e-4 and (da[c]["mask"] - db[c]["mask"]).abs().max() < 1e-3
                seen += da[c]["box"].shape[0]
        if t >= 2:
            assert torch.isfinite(pa).all()
    assert seen > 10 and net_a._planar.fmt == 0 and net_a._planar_planes == "bf16x3"
    # one layer's weights out of range: same fallback on the first step, same results as the bf16x3 graph of the same weights
    (_, a2), (_, b2) = make("fp16x2", True), make("bf16x3", True)
    pa, pb = a2.step(frames[0], is_first=True), b2.step(frames[0], is_first=True)
    assert a2.fell_back and torch.equal(pa, pb)
    # the loud form
    _, a3 = make("fp16x2")
    a3.range_fallback = False
    with pytest.raises(StmError, match="range of the fp16x2 planar format"):
        a3.step(frames[2], is_first=True)


@pytest.mark.parametrize("name,tag", [CASES[0], CASES[3]])
def test_fp16x1_backbone_config5_flavour(name, tag):
    """BASELINE config 5, "fp16 MFMA backbone convs": optimize_for_inference(planes="fp16x1") runs every convolution of the
    ResNet backbone (bottleneck 1x1 / 3x3, projections, DCN offset convs, DCN GEMMs) on ONE fp16 plane -- hand-written
    v_mfma_f32_16x16x32_f16, fp32 accumulation / bias / residual -- while FPN, proto-net, heads and TemporalNet stay in the
    fp32-equivalent fp16x2 format.  Stated tolerance of the trunk outputs against the fp32 graph: 2e-2 of their range (fp16
    activations through 50 / 101 layers; measured ~3e-3); detections: >= 85 % of the reference's instances found (IoU > 0.5,
    same class) on every frame, their soft masks within 2e-2 RMS."""
    from stmask_amd.fuse import optimize_for_inference
    from test_gpu_parity import match_instances, soft_mask_delta
    g = load_golden(f"model_{tag}.npz")
    h, w = [int(v) for v in g["frames_hw"]]
    ref_net = build(name)
    opt_net = build(name)
    optimize_for_inference(opt_net, planar=True, planes="fp16x1")
    opt_net = opt_net.to(memory_format=torch.channels_last)
    opt_net.TemporalNet = opt_net.TemporalNet.to(memory_format=torch.contiguous_format)
    assert opt_net._planar_backbone.fmt == 2 and opt_net._planar.fmt == 1 and opt_net._planar_temporal.fmt == 1
    assert all(e["c1"].fmt == 2 and e["c3"].fmt == 2 for blks in opt_net._planar_backbone.blocks for e in blks)
    frames = synthetic.synthetic_clip(int(g["n_frames"]), h, w, seed=0)
    x = frames[:1].cuda()
    with torch.no_grad():
        _, a = ref_net.forward_single(x)
        _, b = opt_net.forward_single(x.contiguous(memory_format=torch.channels_last))
    errs = {}
    for k in ("proto", "loc", "mask_coeff", "conf"):
        errs[k] = (a[k] - b[k]).abs().max().item() / max(1e-6, a[k].abs().max().item())
        assert 1e-6 < errs[k] < 2e-2, (k, errs[k])       # fp16-level, and really not the fp32-equivalent path
    outs = run_clip(opt_net, frames.contiguous(memory_format=torch.channels_last), "cuda")
    for t, det in enumerate(outs):
        ref_box, ref_cls = g[f"t{t}_box"], g[f"t{t}_class"]
        gi, ri = match_instances(det["box"].cpu(), det["class"].cpu(), ref_box, ref_cls)
        assert len(gi) >= 0.85 * ref_box.shape[0], (t, len(gi), ref_box.shape[0])
        rms, _, _ = soft_mask_delta(det["mask"].cpu()[gi], g[f"t{t}_mask"][ri])
        assert rms.max().item() < 2e-2, (t, rms.max().item())


def test_batched_pipeline_trunk_from_hip_graphs_equals_eager():
    """use_graph: the trunk is replayed from round-robin HIP graphs (static input, static outputs, a memory pool and workspaces per slot).  Twelve
    steps of three clips -- without look-ahead, with the next frame's trunk on a side stream, with the next TWO frames' trunks on two side streams
    (two graph replays running beside each other and beside the tracker tail) -- tracker resets in the middle, and a caller that changes its mind
    about the frames it announced: same packed detections as the eager pipeline bit for bit (same kernels on the same data), and the graphs
    really are replayed."""
    from stmask_amd.pipeline import BatchedClipPipeline
    from stmask_amd.fuse import optimize_for_inference
    net = build("STMask_plus_resnet50_config")
    optimize_for_inference(net, planar=True)
    net = net.to(memory_format=torch.channels_last)
    net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    T = 14
    clips = torch.stack([synthetic.synthetic_clip(T, 128, 192, seed=s) for s in (0, 5, 9)]).cuda()
    frames = [clips[:, t].contiguous(memory_format=torch.channels_last) for t in range(T)]
    for depth in (0, 1, 2, 3):
        eager, graphed = BatchedClipPipeline(net, 3), BatchedClipPipeline(net, 3)
        graphed.use_graph = True
        for t in range(T):
            first = t in (0, 5)
            nxt = [frames[u] for u in range(t + 1, min(t + 1 + depth, T))] or None
            if depth >= 2 and t == 8:
                nxt = [frames[2], frames[3], frames[4]][:depth]          # announced, never passed: the trunks are dropped by the next step
            ya = eager.step(frames[t], is_first=first, next_frames=nxt[0] if nxt else None).clone()
            yb = graphed.step(frames[t], is_first=first, next_frames=nxt).clone()
            torch.cuda.synchronize()
            assert torch.equal(ya, yb), (depth, t, (ya - yb).abs().max().item())
        assert graphed.graph_active and len(graphed._graphs) == BatchedClipPipeline.N_GRAPH_SLOTS
        assert len(graphed._sides) == (0 if depth == 0 else max(2, BatchedClipPipeline.PREFETCH_DEPTH))          # the look-ahead trunks rotate over the side streams
        assert not eager.graph_active


def test_detection_gather_stream_is_not_a_trunk_stream():
    """The all-gather's communication stream must be none of the streams the pipeline rotates its prefetched trunks over (at the default
    look-ahead of three trunks the gather used to share a stream with every third trunk), nor the main stream."""
    from stmask_amd.dist import DetectionGatherer
    from stmask_amd.pipeline import concurrent_side_streams, trunk_stream_count
    dev = torch.device("cuda", torch.cuda.current_device())
    sides = concurrent_side_streams(dev, trunk_stream_count())
    comm = DetectionGatherer(dev).comm_stream(dev)
    assert len(sides) == trunk_stream_count() >= 2
    assert all(comm != s for s in sides) and comm != torch.cuda.current_stream(dev)
    # asking for the trunk streams again (a second pipeline in the process) returns the same ones, still without the gather's
    assert concurrent_side_streams(dev, trunk_stream_count()) == sides


def test_trunk_graph_ring_is_not_built_when_it_would_not_fit(monkeypatch, capsys):
    """Every trunk-graph slot keeps a private pool of a trunk's activations; when the 2 D + 1 further slots would not fit in the free HBM the pipeline says so once
    and keeps the eager trunk -- same results."""
    from stmask_amd.pipeline import BatchedClipPipeline
    from stmask_amd.fuse import optimize_for_inference
    net = build("STMask_plus_resnet50_config")
    optimize_for_inference(net, planar=True)
    net = net.to(memory_format=torch.channels_last)
    net.TemporalNet = net.TemporalNet.to(memory_format=torch.contiguous_format)
    T = 6
    clips = torch.stack([synthetic.synthetic_clip(T, 128, 192, seed=s) for s in (0, 5)]).cuda()
    frames = [clips[:, t].contiguous(memory_format=torch.channels_last) for t in range(T)]
    eager, tight = BatchedClipPipeline(net, 2), BatchedClipPipeline(net, 2)
    tight.use_graph = True
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (-(1 << 50), 1 << 38))          # "no free memory at all"
    for t in range(T):
        nxt = [frames[u] for u in range(t + 1, min(t + 3, T))] or None
        ya = eager.step(frames[t], is_first=(t == 0), next_frames=nxt[0] if nxt else None).clone()
        yb = tight.step(frames[t], is_first=(t == 0), next_frames=nxt).clone()
        torch.cuda.synchronize()
        assert torch.equal(ya, yb), t
    assert not tight.use_graph and not tight.graph_active and tight._graphs == [] and tight._graph_ws == []
    assert "do not fit" in capsys.readouterr().err
