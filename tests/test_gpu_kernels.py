"""GPU parity tests: every hand-written gfx950 kernel (called through the C ABI) against the CPU oracle on the same
seeded inputs, against the reference-generated golden fixtures, and -- at BASELINE sizes -- through size-independent
properties.  Tolerances: bit-exact for decode / IoU / NMS indices / mask IoU; 1e-4 abs (north star) for fp32
deformable conv, correlation, RoIAlign and masks, with the tighter figure each kernel actually achieves asserted too.
"""
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import ulp_diff
from stmask_amd import ops
from stmask_amd._lib import StmError

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def make_deform_case(B, C, H, W, kh, kw, sh, ph, pw, dg, seed, off_scale=2.0, with_mask=True):
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, sh, sh, ph, pw, 1, 1)
    x = rnd(B, C, H, W, seed=seed)
    off = rnd(B, dg * 2 * kh * kw, Ho, Wo, seed=seed + 1, scale=off_scale)
    mask = torch.sigmoid(rnd(B, dg * kh * kw, Ho, Wo, seed=seed + 2)) if with_mask else None
    return x, off, mask


DEFORM_CASES = [
    # B, C,  H,  W, kh, kw, s, ph, pw, dg, off_scale, mask   (reduced versions of the R50 layer shapes + FCB shapes)
    (2, 16, 24, 40, 3, 3, 2, 1, 1, 1, 2.0, True),    # L1.0-like stride 2
    (1, 32, 12, 20, 3, 3, 1, 1, 1, 1, 2.0, True),    # stride 1
    (2, 8, 12, 20, 3, 5, 1, 1, 2, 1, 1.5, False),    # FCB 3x5, v1
    (1, 8, 6, 10, 5, 3, 1, 2, 1, 1, 1.5, False),     # FCB 5x3 (HWo % 4 == 0)
    (1, 8, 3, 5, 5, 3, 1, 2, 1, 1, 1.0, False),      # P7: HWo = 15 -> scalar store path
    (1, 16, 9, 13, 3, 3, 1, 1, 1, 2, 3.0, True),     # odd sizes, 2 deformable groups
    (1, 8, 16, 16, 3, 3, 1, 1, 1, 1, 12.0, True),    # offsets far beyond the halo -> global fallback
    (1, 6, 10, 12, 3, 3, 2, 1, 1, 1, 2.0, True),     # C % 4 != 0 -> direct kernel
]


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("case", DEFORM_CASES)
def test_deform_im2col_vs_oracle(case, variant):
    B, C, H, W, kh, kw, s, ph, pw, dg, osc, wm = case
    x, off, mask = make_deform_case(B, C, H, W, kh, kw, s, ph, pw, dg, seed=7, off_scale=osc, with_mask=wm)
    ref = oracle.deform_im2col(x, off, mask, (kh, kw), s, (ph, pw), 1, dg)
    got = ops.deform_im2col(x.to(DEV), off.to(DEV), mask.to(DEV) if wm else None, (kh, kw), s, (ph, pw), 1, dg,
                            variant=variant).cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    assert err < 2e-5, err  # fp32 bilinear vs double oracle on O(1) data


@pytest.mark.parametrize("case", DEFORM_CASES[:7])
def test_deform_im2col_variants_agree_bitwise(case):
    B, C, H, W, kh, kw, s, ph, pw, dg, osc, wm = case
    x, off, mask = make_deform_case(B, C, H, W, kh, kw, s, ph, pw, dg, seed=9, off_scale=osc, with_mask=wm)
    a = ops.deform_im2col(x.to(DEV), off.to(DEV), mask.to(DEV) if wm else None, (kh, kw), s, (ph, pw), 1, dg, variant=1)
    for v in (2, 3):
        b = ops.deform_im2col(x.to(DEV), off.to(DEV), mask.to(DEV) if wm else None, (kh, kw), s, (ph, pw), 1, dg, variant=v)
        assert torch.equal(a, b), v


def test_deform_im2col_full_size_layers_variants_agree():
    """BASELINE sizes (R50 @384x640): all 7 DCN layer shapes, LDS-tiled kernel == direct kernel bit for bit."""
    shapes = [(128, 96, 160, 2), (128, 48, 80, 1), (256, 48, 80, 2), (256, 24, 40, 1), (512, 24, 40, 2), (512, 12, 20, 1)]
    for C, H, W, s in shapes:
        x, off, mask = make_deform_case(2, C, H, W, 3, 3, s, 1, 1, 1, seed=C + s)
        xd, od, md = x.to(DEV), off.to(DEV), mask.to(DEV)
        a = ops.deform_im2col(xd, od, md, 3, s, 1, 1, 1, variant=1)
        for v in (2, 3):
            b = ops.deform_im2col(xd, od, md, 3, s, 1, 1, 1, variant=v)
            assert torch.equal(a, b), (C, H, W, s, v)
        # spot-check one image row block against the oracle (full oracle im2col at this size takes seconds)
        ref = oracle.deform_im2col(x[:1, :8], off[:1], mask[:1], 3, s, 1, 1, 1)
        assert (b[:1, :72].cpu() - ref).abs().max() < 2e-5


def test_deform_im2col_fused_offset_mask_logits():
    """DCN path: the kernel reads the raw conv_offset_mask output (chunk/cat/sigmoid fused)."""
    B, C, H, W = 2, 16, 12, 20
    x = rnd(B, C, H, W, seed=1)
    om = rnd(B, 27, H, W, seed=2, scale=1.5)
    o1, o2, m = torch.chunk(om, 3, dim=1)
    ref = oracle.deform_im2col(x, torch.cat((o1, o2), 1), torch.sigmoid(m), 3, 1, 1, 1, 1)
    for v in (1, 2, 3):
        got = ops.deform_im2col(x.to(DEV), None, None, 3, 1, 1, 1, 1, variant=v, fused_om=om.to(DEV)).cpu()
        assert (got - ref).abs().max() < 2e-5


@pytest.mark.parametrize("case", DEFORM_CASES)
def test_deform_conv_vs_oracle(case):
    B, C, H, W, kh, kw, s, ph, pw, dg, osc, wm = case
    O = 24
    x, off, mask = make_deform_case(B, C, H, W, kh, kw, s, ph, pw, dg, seed=3, off_scale=osc, with_mask=wm)
    w = rnd(O, C, kh, kw, seed=11, scale=(C * kh * kw) ** -0.5)
    bias = rnd(O, seed=12) if wm else None
    ref = oracle.deform_conv(x, off, mask, w, bias, s, (ph, pw), 1, dg)
    got = ops.deform_conv(x.to(DEV), off.to(DEV), mask.to(DEV) if wm else None, w.to(DEV),
                          bias.to(DEV) if wm else None, s, (ph, pw), 1, dg).cpu()
    assert (got - ref).abs().max() < 1e-4       # north-star tolerance
    assert (got - ref).abs().max() < 2e-5       # what fp32 MFMA accumulation actually achieves at this K


def test_deform_conv_known_answers():
    """Zero offsets + unit mask == dense conv; mask 0.5 == half of it (the reference's zero-init state,
    backbone.py:24-26); integer offsets == conv of the shifted image (interior); 3x5 / 5x3 keep HxW."""
    x = rnd(2, 16, 20, 28, seed=5).to(DEV)
    for (kh, kw), (ph, pw), s in [((3, 3), (1, 1), 1), ((3, 3), (1, 1), 2), ((3, 5), (1, 2), 1), ((5, 3), (2, 1), 1)]:
        w = rnd(12, 16, kh, kw, seed=6, scale=0.1).to(DEV)
        Ho, Wo = ops.conv_out_hw(20, 28, kh, kw, s, s, ph, pw, 1, 1)
        off = torch.zeros(2, 2 * kh * kw, Ho, Wo, device=DEV)
        ones = torch.ones(2, kh * kw, Ho, Wo, device=DEV)
        dense = F.conv2d(x.double().cpu(), w.double().cpu(), None, s, (ph, pw)).float()
        got = ops.deform_conv(x, off, ones, w, None, s, (ph, pw)).cpu()
        assert got.shape == dense.shape
        assert (got - dense).abs().max() < 2e-5
        half = ops.deform_conv(x, off, ones * 0.5, w, None, s, (ph, pw)).cpu()
        assert (half - 0.5 * dense).abs().max() < 2e-5
        v1 = ops.deform_conv(x, off, None, w, None, s, (ph, pw)).cpu()
        assert (v1 - dense).abs().max() < 2e-5
    # integer shift (+1 row, -2 cols) == conv of the rolled image away from the border
    w = rnd(12, 16, 3, 3, seed=8, scale=0.1).to(DEV)
    off = torch.zeros(2, 18, 20, 28, device=DEV)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    got = ops.deform_conv(x, off, None, w, None, 1, 1).cpu()
    shifted = torch.roll(x.cpu(), shifts=(-1, 2), dims=(2, 3))
    dense = F.conv2d(shifted.double(), w.double().cpu(), None, 1, 1).float()
    assert (got[:, :, 3:-3, 4:-4] - dense[:, :, 3:-3, 4:-4]).abs().max() < 2e-5


def test_deform_conv_linearity_full_size():
    """Property at BASELINE size (L2.2: 256ch 24x40, batch 4): conv(a*x1 + x2) == a*conv(x1) + conv(x2)."""
    B, C, H, W = 4, 256, 24, 40
    x1, off, mask = make_deform_case(B, C, H, W, 3, 3, 1, 1, 1, 1, seed=21)
    x2 = rnd(B, C, H, W, seed=22)
    w = rnd(256, C, 3, 3, seed=23, scale=(C * 9) ** -0.5).to(DEV)
    od, md = off.to(DEV), mask.to(DEV)
    f = lambda t: ops.deform_conv(t.to(DEV), od, md, w, None, 1, 1)
    lhs = f(0.5 * x1 + x2)
    rhs = 0.5 * f(x1) + f(x2)
    assert (lhs - rhs).abs().max() < 1e-4


@pytest.mark.parametrize("M,N,K,batch", [(64, 64, 16, 1), (128, 3840, 1152, 2), (256, 960, 2304, 1), (512, 240, 4608, 3),
                                         (41, 15, 37, 2), (256, 60, 3840, 1), (130, 257, 50, 1)])
def test_gemm_bias_f32(M, N, K, batch):
    A = rnd(M, K, seed=1, scale=K ** -0.5)
    Bm = rnd(batch, K, N, seed=2)
    bias = rnd(M, seed=3)
    ref = (A.double() @ Bm.double() + bias.double()[None, :, None])
    got = ops.gemm_bias(A.to(DEV), Bm.to(DEV), bias.to(DEV)).cpu()
    assert (got.double() - ref).abs().max() < 2e-5
    got_relu = ops.gemm_bias(A.to(DEV), Bm.to(DEV), None, relu=True).cpu()
    assert (got_relu.double() - (A.double() @ Bm.double()).clamp(min=0)).abs().max() < 2e-5


def test_gemm_is_exact_fmaf_chain_on_integers():
    """A = I with an ASYMMETRIC B catches a transposed C/D layout; small integers are exact in fp32."""
    M = K = 128
    A = torch.eye(M)
    Bm = (torch.arange(K * 192).view(K, 192) % 251).float()
    assert torch.equal(ops.gemm_bias(A.to(DEV), Bm.to(DEV)).cpu(), Bm)
    A2 = ((torch.arange(M * K).view(M, K) % 7) - 3).float()
    assert torch.equal(ops.gemm_bias(A2.to(DEV), Bm.to(DEV)).cpu(), A2 @ Bm)


# ------------------------------------------------------------------------------------------ temporal
@pytest.mark.parametrize("B,C,H,W", [(1, 256, 24, 40), (2, 32, 12, 20), (1, 20, 7, 9), (1, 8, 3, 5), (3, 20, 6, 8), (2, 40, 30, 44),
                                     (2, 256, 46, 80), (1, 24, 13, 40), (2, 12, 9, 16), (1, 9, 5, 72)])
def test_correlation_vs_oracle(B, C, H, W):
    f1, f2 = rnd(B, C, H, W, seed=1), rnd(B, C, H, W, seed=2)
    ref = oracle.corr_patch(f1, f2, 11, 1)
    got = ops.corr_patch(f1.to(DEV), f2.to(DEV), 11, 1).cpu()
    assert got.shape == (B, 11, 11, H, W)
    assert (got - ref).abs().max() < 1e-4 * max(1.0, C / 64)
    # fused epilogue of correlate(): / C and leaky_relu(0.1)
    ref2 = oracle.correlate(f1, f2, 11)
    got2 = ops.corr_patch(f1.to(DEV), f2.to(DEV), 11, 1, scale=1.0 / C, leaky_slope=0.1).view(B, 121, H, W).cpu()
    assert (got2 - ref2).abs().max() < 2e-6
    # channels-last form (what the temporal fusion's RoI kernel gathers from): the same values, bit for bit, in [B, H, W, 128];
    # the 7 spare channels are not written (both the tiled and the generic kernel: W % 4 != 0 takes the latter)
    nhwc = ops.corr_patch_nhwc(f1.to(DEV), f2.to(DEV), 11, scale=1.0 / C, leaky_slope=0.1).cpu()
    assert nhwc.shape == (B, H, W, 128) and torch.equal(nhwc[..., :121], got2.permute(0, 2, 3, 1))
    # channels_last inputs (the trunk's fp32 outputs) are read in place: a staging unit is (row, x, 4 channels) instead of
    # (channel, row, 4 x), the LDS image and the arithmetic are the same -> the same bits (C % 4 != 0 takes the generic kernel)
    cl1, cl2 = (t.to(DEV).contiguous(memory_format=torch.channels_last) for t in (f1, f2))
    assert not cl1.is_contiguous() or C == 1
    nhwc_cl = ops.corr_patch_nhwc(cl1, cl2, 11, scale=1.0 / C, leaky_slope=0.1).cpu()
    if C % 4 == 0 or W % 4 != 0:
        assert torch.equal(nhwc_cl[..., :121], nhwc[..., :121])
    else:       # NCHW inputs on the tiled kernel, channels-last ones on the generic kernel: another summation order
        assert (nhwc_cl[..., :121] - nhwc[..., :121]).abs().max() < 2e-6


def test_correlation_known_answers_and_generic_path(tunables):
    f1 = rnd(1, 16, 12, 20, seed=4)
    f2 = torch.roll(f1, shifts=(2, -3), dims=(2, 3))
    out = ops.corr_patch(f1.to(DEV), f2.to(DEV), 11, 1).cpu()
    centre = (f1 * f2).sum(1)
    assert (out[0, 5, 5] - centre[0]).abs().max() < 1e-5            # centre tap == channel dot product
    energy = (f1 * f1).sum(1)[0]
    assert (out[0, 7, 2, 2:-2, 3:-3] - energy[2:-2, 3:-3]).abs().max() < 1e-5   # peak at the true displacement
    # other patch sizes / dilation take the generic kernel
    for P, dil in [(5, 1), (7, 2)]:
        ref = oracle.corr_patch(f1, f2, P, dil)
        got = ops.corr_patch(f1.to(DEV), f2.to(DEV), P, dil).cpu()
        assert (got - ref).abs().max() < 1e-5
    tunables.set(STM_CORR_VARIANT="1")
    got = ops.corr_patch(f1.to(DEV), f2.to(DEV), 11, 1).cpu()
    assert (got - out).abs().max() < 1e-5


def test_correlation_symmetry_full_size():
    """corr(f1,f2)[i,j,y,x] == corr(f2,f1)[10-i,10-j,y+i-5,x+j-5] (BASELINE P4 size, batch 8)."""
    f1, f2 = rnd(8, 256, 24, 40, seed=5).to(DEV), rnd(8, 256, 24, 40, seed=6).to(DEV)
    a, b = ops.corr_patch(f1, f2, 11, 1), ops.corr_patch(f2, f1, 11, 1)
    for i, j in [(0, 0), (3, 9), (5, 5), (10, 2)]:
        dy, dx = i - 5, j - 5
        ys = slice(max(0, -dy), 24 - max(0, dy))
        xs = slice(max(0, -dx), 40 - max(0, dx))
        ys2 = slice(max(0, -dy) + dy, 24 - max(0, dy) + dy)
        xs2 = slice(max(0, -dx) + dx, 40 - max(0, dx) + dx)
        assert (a[:, i, j, ys, xs] - b[:, 10 - i, 10 - j, ys2, xs2]).abs().max() < 1e-4


def test_roi_align_vs_oracle_and_known_answers():
    feat = rnd(1, 40, 24, 40, seed=3)
    g = torch.Generator().manual_seed(4)
    xy = torch.rand(30, 2, generator=g) * torch.tensor([30.0, 16.0])
    wh = torch.rand(30, 2, generator=g) * torch.tensor([20.0, 14.0]) + 0.3
    rois = torch.cat([torch.zeros(30, 1), xy, xy + wh], 1)
    rois[0] = torch.tensor([0, -3.0, -2.0, 50.0, 30.0])   # overhangs the map
    rois[1] = torch.tensor([0, 5.0, 5.0, 5.0, 5.0])       # empty roi
    for aligned, sr in [(True, 0), (False, 0), (True, 2)]:
        ref = oracle.roi_align(feat, rois, 7, 1.0, sr, "avg", aligned)
        got = ops.roi_align(feat.to(DEV), rois.to(DEV), 7, 1.0, sr, aligned).cpu()
        assert (got - ref).abs().max() < 1e-5
    const = torch.full((1, 3, 24, 40), 2.5)
    inside = rois[2:].clone()                               # RoIs overhanging the map legitimately average in zeros
    inside[:, 3] = inside[:, 3].clamp(max=39.0)
    inside[:, 4] = inside[:, 4].clamp(max=23.0)
    out = ops.roi_align(const.to(DEV), inside.to(DEV), 7).cpu()
    assert (out - 2.5).abs().max() < 1e-6                  # constant map -> constant
    ramp = torch.arange(40.0).view(1, 1, 1, 40).expand(1, 1, 24, 40).contiguous()
    r = torch.tensor([[0, 4.0, 3.0, 18.0, 17.0]])
    out = ops.roi_align(ramp.to(DEV), r.to(DEV), 7).cpu()[0, 0, 0]
    centres = 4.0 - 0.5 + (torch.arange(7.0) + 0.5) * 2.0   # aligned=True shifts by -0.5; linear ramp -> bin centre
    assert (out - centres).abs().max() < 1e-5
    assert ops.roi_align(feat.to(DEV), torch.zeros(0, 5, device=DEV), 7).shape == (0, 40, 7, 7)


# ------------------------------------------------------------------------------------------ post-processing
CASES = ["c0_", "c1_", "c2_"]


@pytest.mark.parametrize("p", CASES)
def test_decode_bit_exact(golden_postproc, p):
    g = golden_postproc
    got = ops.decode(g[p + "loc"].to(DEV), g["priors"].to(DEV)).cpu()
    assert torch.equal(got, oracle.decode(g[p + "loc"], g["priors"]))          # bit-exact vs oracle
    ref = g[p + "boxes"]                                                        # reference (MKL exp): <= 1 ULP of w/h
    wh = (ref[:, 2:] - ref[:, :2]).abs().repeat(1, 2)
    assert ((got - ref).abs() <= 1.2e-7 * (wh + ref.abs()) + 1e-9).all()


def test_decode_bit_exact_full_size_and_extremes():
    n = 15345 * 8
    g = torch.Generator().manual_seed(1)
    loc = torch.randn(n, 4, generator=g) * torch.tensor([2.0, 2.0, 8.0, 8.0])
    loc[:8, 2:] = torch.tensor([[0.0, -0.0], [500.0, -500.0], [440.0, 443.5], [-430.0, -520.0], [1e-8, -1e-8],
                                [88.0, 89.0], [3.0, 4.0], [-3.0, -4.0]])
    pri = torch.rand(n, 4, generator=g)
    got, ref = ops.decode(loc.to(DEV), pri.to(DEV)).cpu(), oracle.decode(loc, pri)
    assert torch.equal(torch.isnan(got), torch.isnan(ref)) and torch.isnan(ref).sum() <= 8   # exp overflow rows: inf - inf
    assert torch.equal(torch.nan_to_num(got, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    assert ops.decode(torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, device=DEV)).shape == (0, 4)


def test_fcb_ali_offsets_bit_exact(golden_fcb_ali):
    g = golden_fcb_ali
    for kh, kw in [(3, 3), (3, 5), (5, 3)]:
        got = ops.fcb_ali_offsets(g["loc"].to(DEV), kh, kw).cpu()
        assert torch.equal(got, oracle.fcb_ali_offsets(g["loc"], kh, kw))
        ref = g[f"off_{kh}x{kw}"]
        assert (got - ref).abs().max() <= 4e-7 * ref.abs().max()


@pytest.mark.parametrize("p", CASES)
def test_generate_candidates_bit_exact(golden_postproc, p):
    g = golden_postproc
    loc, conf, pri = g[p + "loc"], g[p + "conf"], g["priors"]
    keep, box, cnt = ops.generate_candidates(loc[None].to(DEV), pri.to(DEV), conf[None].to(DEV), 0.05)
    k = int(cnt[0])
    assert torch.equal(keep[0, :k].cpu(), g[p + "keep_idx"])                   # reference's own keep set
    assert torch.equal(keep[0, :k].cpu(), oracle.candidate_filter(conf, 0.05))
    assert torch.equal(box[0, :k].cpu(), oracle.decode(loc, pri)[g[p + "keep_idx"]])
    assert (keep[0, k:] == 0).all() and (box[0, k:] == 0).all()


def test_generate_candidates_batched_and_empty():
    g = torch.Generator().manual_seed(2)
    N = 3000
    loc, pri = torch.randn(3, N, 4, generator=g), torch.rand(N, 4, generator=g)
    logits = torch.randn(3, N, 41, generator=g)
    logits[..., 0] += 4.0
    logits[1] -= 100.0  # frame 1: nothing but background
    logits[1, :, 0] += 200.0
    conf = torch.softmax(logits, -1)
    keep, box, cnt = ops.generate_candidates(loc.to(DEV), pri.to(DEV), conf.to(DEV), 0.05)
    for b in range(3):
        ref = oracle.candidate_filter(conf[b], 0.05)
        assert int(cnt[b]) == len(ref)
        assert torch.equal(keep[b, :len(ref)].cpu(), ref)
        assert torch.equal(box[b, :len(ref)].cpu(), oracle.decode(loc[b], pri)[ref])
    assert int(cnt[1]) == 0 and int(cnt[0]) > 50


@pytest.mark.parametrize("p", CASES)
def test_cc_fast_nms_bit_exact(golden_postproc, p):
    g = golden_postproc
    conf, box, cen = g[p + "cand_conf"], g[p + "cand_box"], g[p + "cand_centerness"]
    idx, cls, sc, bx, cnt = ops.cc_fast_nms(conf.to(DEV), box.to(DEV), cen.to(DEV), 0.5, 200)
    n = int(cnt)
    o_idx, o_cls, o_sc = oracle.cc_fast_nms(conf, box, cen, 0.5, 200)
    assert n == len(o_idx)
    assert torch.equal(idx[:n].cpu(), o_idx) and torch.equal(cls[:n].cpu(), o_cls) and torch.equal(sc[:n].cpu(), o_sc)
    # and against the reference's own outputs
    assert torch.equal(bx[:n].cpu(), g[p + "cc_box"]) and torch.equal(cls[:n].cpu(), g[p + "cc_class"])
    assert torch.equal(sc[:n].cpu(), g[p + "cc_score"])
    assert (idx[n:] == 0).all() and (sc[n:] == 0).all()


def test_cc_fast_nms_edge_cases():
    dev = DEV
    # ties -> lower row first; chain suppression; NaN IoU (zero-area duplicates) dropped
    boxes = torch.tensor([[0.1, 0.1, 0.3, 0.3], [0.6, 0.6, 0.9, 0.9], [0.1, 0.1, 0.3, 0.3], [0.4, 0.1, 0.5, 0.2]])
    conf = torch.zeros(4, 41)
    conf[:, 3] = 0.5
    idx, cls, sc, _, cnt = ops.cc_fast_nms(conf.to(dev), boxes.to(dev), None, 0.5, 200)
    assert idx[:int(cnt)].tolist() == [0, 1, 3] and cls[:int(cnt)].tolist() == [3, 3, 3]
    A, Bx, C = [0.10, 0.10, 0.50, 0.50], [0.22, 0.10, 0.62, 0.50], [0.34, 0.10, 0.74, 0.50]
    conf = torch.zeros(3, 41)
    conf[:, 1] = torch.tensor([0.9, 0.8, 0.7])
    idx, _, _, _, cnt = ops.cc_fast_nms(conf.to(dev), torch.tensor([A, Bx, C]).to(dev), None, 0.5, 200)
    assert idx[:int(cnt)].tolist() == [0]
    z = torch.tensor([[0.5, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 0.5]])
    c2 = torch.zeros(2, 41)
    c2[:, 1] = torch.tensor([0.9, 0.8])
    idx, _, _, _, cnt = ops.cc_fast_nms(c2.to(dev), z.to(dev), None, 0.5, 200)
    assert idx[:int(cnt)].tolist() == [0]
    # K = 1, K = 0, device-side K
    one = ops.cc_fast_nms(conf[:1].to(dev), torch.tensor([A]).to(dev), torch.tensor([0.5]).to(dev), 0.5, 200)
    assert int(one[4]) == 1 and one[1][0].item() == 1 and abs(one[2][0].item() - 0.45) < 1e-7
    none = ops.cc_fast_nms(torch.zeros(0, 41, device=dev), torch.zeros(0, 4, device=dev), None, 0.5, 200)
    assert int(none[4]) == 0
    kd = torch.tensor([2], dtype=torch.int32, device=dev)
    part = ops.cc_fast_nms(conf.to(dev), torch.tensor([A, Bx, C]).to(dev), None, 0.5, 200, k_dev=kd)
    full2 = ops.cc_fast_nms(conf[:2].to(dev), torch.tensor([A, Bx]).to(dev), None, 0.5, 200)
    assert int(part[4]) == int(full2[4]) and torch.equal(part[0][:1], full2[0][:1])


@pytest.mark.parametrize("K", [5000, 15345])
def test_cc_fast_nms_large_k_vs_oracle(K):
    """Maximum sizes: every prior a candidate (sorts 8192 / 16384 keys in LDS)."""
    g = torch.Generator().manual_seed(K)
    conf = torch.softmax(torch.randn(K, 41, generator=g) * 2, -1)
    c = torch.rand(K, 2, generator=g)
    wh = torch.rand(K, 2, generator=g) * 0.2 + 0.01
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    cen = torch.tanh(torch.randn(K, generator=g) + 1)
    idx, cls, sc, _, cnt = ops.cc_fast_nms(conf.to(DEV), boxes.to(DEV), cen.to(DEV), 0.5, 200)
    o_idx, o_cls, o_sc = oracle.cc_fast_nms(conf, boxes, cen, 0.5, 200)
    n = int(cnt)
    assert n == len(o_idx) and torch.equal(idx[:n].cpu(), o_idx) and torch.equal(cls[:n].cpu(), o_cls)
    assert torch.equal(sc[:n].cpu(), o_sc)


@pytest.mark.parametrize("quantum", [0.0, 0.02])
def test_fused_detect_more_candidates_than_the_lds_sort_holds(quantum):
    """BASELINE config 5 size: N_p = 58 860 priors (736x1280) with ~40 000 of them over the threshold -- more than the 16 384
    keys the one-workgroup sort holds.  The kernel must still return exactly the reference's result (sort ALL candidates,
    top 200): radix select of the 200th best key, ties by lower row.  quantum > 0 quantises the scores so that the 200th
    best score is shared by hundreds of rows (the tie rule decides which of them enter)."""
    from oracle.cpu_path import _detect_cc
    N, B = 58860, 2
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(B, N, 41, generator=g) * 2.0
    logits[..., 0] -= 1.5
    logits[1, ::3, 0] += 30.0                                   # frame 1: a third of the rows are background
    conf = torch.softmax(logits, -1)
    if quantum:
        conf = (conf / quantum).floor() * quantum + 0.001
    cen = torch.ones(B, N) if quantum else torch.tanh(torch.randn(B, N, generator=g) + 1.5)
    c = torch.rand(N, 2, generator=g)
    wh = torch.rand(N, 2, generator=g) * 0.2 + 0.02
    pri = torch.cat([c, wh], 1)
    loc = torch.randn(B, N, 4, generator=g) * 0.5
    n_cand = [(conf[b, :, 1:].max(1).values > 0.05).sum().item() for b in range(B)]
    assert n_cand[0] > 16384 and n_cand[1] > 16384, n_cand
    idx, cls, sc, bx, cnt = ops.detect_cc(loc.to(DEV), pri.to(DEV), conf.to(DEV), cen.to(DEV), 0.05, 0.5, 200)
    o_idx, o_cls, o_sc, o_bx, o_cnt = _detect_cc(loc, pri, conf, cen, 0.05, 0.5, 200)
    for b in range(B):
        n = int(cnt[b])
        assert n == int(o_cnt[b]) and n > 0
        assert torch.equal(idx[b, :n].cpu(), o_idx[b, :n]) and torch.equal(cls[b, :n].cpu(), o_cls[b, :n])
        assert torch.equal(sc[b, :n].cpu(), o_sc[b, :n]) and torch.equal(bx[b, :n].cpu(), o_bx[b, :n])
    # determinism (the insertion order of the candidates is an atomic race; the result must not depend on it)
    again = ops.detect_cc(loc.to(DEV), pri.to(DEV), conf.to(DEV), cen.to(DEV), 0.05, 0.5, 200)
    assert torch.equal(again[0], idx) and torch.equal(again[2], sc)


def test_cc_fast_nms_more_rows_than_the_lds_sort_holds():
    """The reference-shaped call (candidate rows in, Detect_TF.cc_fast_nms) with K = 57 000 rows -- what a 736x1280 frame of the
    FCB(ali) config produces with synthetic weights -- goes through stm_cc_fast_nms_ws_f32 and equals the oracle."""
    K = 57000
    g = torch.Generator().manual_seed(21)
    conf = torch.softmax(torch.randn(K, 41, generator=g) * 2, -1)
    c = torch.rand(K, 2, generator=g)
    wh = torch.rand(K, 2, generator=g) * 0.2 + 0.01
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    cen = torch.tanh(torch.randn(K, generator=g) + 1)
    idx, cls, sc, bx, cnt = ops.cc_fast_nms(conf.to(DEV), boxes.to(DEV), cen.to(DEV), 0.5, 200)
    o_idx, o_cls, o_sc = oracle.cc_fast_nms(conf, boxes, cen, 0.5, 200)
    n = int(cnt)
    assert n == len(o_idx) and torch.equal(idx[:n].cpu(), o_idx) and torch.equal(cls[:n].cpu(), o_cls)
    assert torch.equal(sc[:n].cpu(), o_sc) and torch.equal(bx[:n].cpu(), boxes[o_idx])
    with pytest.raises(StmError, match="use stm_cc_fast_nms_ws_f32"):      # the LDS-only entry point stays loud
        ops.cc_fast_nms(conf.to(DEV), boxes.to(DEV), cen.to(DEV), 0.5, 200, k_dev=torch.tensor([K], dtype=torch.int32, device=DEV))


@pytest.mark.parametrize("p", CASES)
def test_fused_detect_equals_chain(golden_postproc, p):
    """stm_detect_cc (decode + threshold + NMS, no host sync) == generate_candidate -> cc_fast_nms of the reference."""
    g = golden_postproc
    loc, conf, cen, pri = g[p + "loc"], g[p + "conf"], g[p + "centerness"], g["priors"]
    B = 3
    locb = torch.stack([loc, loc.flip(0), loc])
    confb = torch.stack([conf, conf.flip(0), conf])
    cenb = torch.stack([cen, cen.flip(0), cen])
    idx, cls, sc, bx, cnt = ops.detect_cc(locb.to(DEV), pri.to(DEV), confb.to(DEV), cenb.to(DEV), 0.05, 0.5, 200)
    n = int(cnt[0])
    assert n == len(g[p + "cc_class"]) and int(cnt[2]) == n
    assert torch.equal(bx[0, :n].cpu(), oracle.decode(loc, pri)[idx[0, :n].cpu()])
    assert torch.equal(idx[0, :n].cpu(), g[p + "keep_idx"][oracle.cc_fast_nms(g[p + "cand_conf"], g[p + "cand_box"],
                                                                              g[p + "cand_centerness"], 0.5, 200)[0]])
    assert torch.equal(cls[0, :n].cpu(), g[p + "cc_class"]) and torch.equal(sc[0, :n].cpu(), g[p + "cc_score"])
    assert torch.equal(idx[2], idx[0]) and torch.equal(sc[2], sc[0])
    # frame 1 (rows flipped) against the oracle chain
    keep = oracle.candidate_filter(confb[1], 0.05)
    bxs = oracle.decode(locb[1], pri)
    o_idx, o_cls, o_sc = oracle.cc_fast_nms(confb[1][keep], bxs[keep], cenb[1][keep].view(-1), 0.5, 200)
    n1 = int(cnt[1])
    assert n1 == len(o_idx) and torch.equal(idx[1, :n1].cpu(), keep[o_idx]) and torch.equal(sc[1, :n1].cpu(), o_sc)


@pytest.mark.parametrize("p", CASES)
def test_per_class_fast_nms_bit_exact(golden_postproc, p):
    g = golden_postproc
    conf, box, cen = g[p + "cand_conf"], g[p + "cand_box"], g[p + "cand_centerness"]
    idx, cls, sc, bx, cnt = ops.fast_nms(conf.to(DEV), box.to(DEV), cen.to(DEV), 0.5, 200, 0.05, 100)
    n = int(cnt)
    o_idx, o_cls, o_sc = oracle.fast_nms(conf, box, cen, 0.5, 200, 0.05, 100)
    assert n == len(o_idx) and torch.equal(idx[:n].cpu(), o_idx) and torch.equal(cls[:n].cpu(), o_cls)
    assert torch.equal(sc[:n].cpu(), g[p + "pc_score"]) and torch.equal(bx[:n].cpu(), g[p + "pc_box"])
    assert torch.equal(cls[:n].cpu(), g[p + "pc_class"])


@pytest.mark.parametrize("p", CASES)
def test_per_class_fast_nms_non_tf_golden(golden_postproc, p):
    """Row a18: Detect.fast_nms (detection.py:211-261) ranks WITHOUT centerness -- the layer API's non-TF detector and the batched op against the
    reference's own outputs (tests/golden/postproc_nontf.npz)."""
    from conftest import load_golden
    from stmask_amd.config import get_cfg
    from stmask_amd.layers.functions import Detect
    g, gn = golden_postproc, load_golden("postproc_nontf.npz")
    cfg = get_cfg("STMask_plus_resnet50_config")
    det = Detect(cfg.num_classes, 0, cfg.nms_top_k, cfg.nms_conf_thresh, cfg.nms_thresh, cfg=cfg)
    det.use_cross_class_nms = False
    preds = {"loc": g[p + "loc"][None].to(DEV), "conf": g[p + "conf"][None].to(DEV), "priors": g["priors"][None].to(DEV),
             "mask_coeff": g[p + "mask_coeff"][None].to(DEV), "track": torch.zeros(1, g[p + "loc"].shape[0], 8, device=DEV),
             "centerness": g[p + "centerness"][None].to(DEV), "proto": g[p + "proto"][None].to(DEV)}
    out = det(preds, None)[0]["detection"]
    assert torch.equal(out["score"].cpu(), gn[p + "pcn_score"]) and torch.equal(out["class"].cpu(), gn[p + "pcn_class"])
    assert torch.equal(out["mask_coeff"].cpu(), gn[p + "pcn_mask_coeff"])
    assert (out["box"].cpu() - gn[p + "pcn_box"]).abs().max().item() <= 2.4e-7        # (decode: <= 4 ULP of w / h from the reference's MKL exp)
    # the batched op with centerness = None: the same rows
    idx, cls, sc, bx, cnt = ops.detect_pc(preds["loc"], preds["priors"][0], preds["conf"], None, 0.05, 0.5, 200, 100)
    n = int(cnt[0])
    assert n == len(gn[p + "pcn_class"]) and torch.equal(sc[0, :n].cpu(), gn[p + "pcn_score"]) and torch.equal(cls[0, :n].cpu(), gn[p + "pcn_class"])


@pytest.mark.parametrize("p", CASES)
def test_jaccard_bit_exact(golden_postproc, p):
    cb = golden_postproc[p + "cand_box"]
    got = ops.jaccard(cb[:64].to(DEV), cb[:96].to(DEV)).cpu()
    assert torch.equal(got, golden_postproc[p + "jaccard"])


@pytest.mark.parametrize("p", CASES)
def test_generate_mask_within_tolerance(golden_postproc, p):
    g = golden_postproc
    proto, coeff, box = g[p + "proto"], g[p + "cc_mask_coeff"], g[p + "cc_box"]
    got = ops.lincomb_sigmoid_crop(proto.to(DEV), coeff.to(DEV), box.to(DEV)).cpu()
    ref_o, ref_g = oracle.generate_mask(proto, coeff, box), g[p + "masks"]
    assert got.shape == ref_g.shape
    for ref in (ref_o, ref_g):
        d = got - ref
        assert d.abs().max() < 1e-4 and (d.pow(2).sum(dim=(1, 2)).sqrt().max() < 1e-4)   # max-abs and per-mask L2
    assert (got - ref_o).abs().max() < 5e-6
    assert torch.equal(got == 0, ref_g == 0)
    nocrop = ops.lincomb_sigmoid_crop(proto.to(DEV), coeff[:5].to(DEV), None).cpu()
    assert (nocrop - g[p + "masks_nocrop"]).abs().max() < 1e-5
    # device-side count: rows >= n_dev are zero
    nd = torch.tensor([3], dtype=torch.int32, device=DEV)
    part = ops.lincomb_sigmoid_crop(proto.to(DEV), coeff.to(DEV), box.to(DEV), n_dev=nd).cpu()
    assert torch.equal(part[:3], got[:3]) and (part[3:] == 0).all()


def test_generate_mask_full_size_and_crop_boundaries(golden_postproc):
    g = golden_postproc
    ones = torch.ones(24, 40, 32)
    coeff = torch.full((5, 32), 10.0)
    got = ops.lincomb_sigmoid_crop(ones.to(DEV), coeff.to(DEV), g["crop_boxes"].to(DEV)).cpu()
    assert torch.equal(got, g["crop_mask"])
    gen = torch.Generator().manual_seed(3)
    proto = torch.relu(torch.randn(96, 160, 32, generator=gen))
    coeff = torch.randn(100, 32, generator=gen)
    c = torch.rand(100, 2, generator=gen)
    wh = torch.rand(100, 2, generator=gen) * 0.5
    box = torch.cat([c - wh / 2, c + wh / 2], 1)
    got = ops.lincomb_sigmoid_crop(proto.to(DEV), coeff.to(DEV), box.to(DEV)).cpu()
    ref = oracle.generate_mask(proto, coeff, box)
    assert (got - ref).abs().max() < 1e-5 and torch.equal(got == 0, ref == 0)
    assert ops.lincomb_sigmoid_crop(proto.to(DEV), coeff[:0].to(DEV), box[:0].to(DEV)).shape == (0, 96, 160)


def test_generate_mask_rows_of_several_clips_in_one_launch():
    gen = torch.Generator().manual_seed(8)
    protos = torch.relu(torch.randn(3, 24, 40, 32, generator=gen))
    coeff = torch.randn(37, 32, generator=gen)
    c = torch.rand(37, 2, generator=gen)
    wh = torch.rand(37, 2, generator=gen) * 0.6
    box = torch.cat([c - wh / 2, c + wh / 2], 1)
    rp = torch.tensor([0] * 5 + [1] * 20 + [2] * 12, dtype=torch.int32)   # chunk of 8 rows straddles clips
    got = ops.lincomb_sigmoid_crop(protos.to(DEV), coeff.to(DEV), box.to(DEV), row_proto=rp.to(DEV)).cpu()
    for pi in range(3):
        sel = torch.nonzero(rp == pi).view(-1)
        ref = oracle.generate_mask(protos[pi], coeff[sel], box[sel])
        assert (got[sel] - ref).abs().max() < 5e-6


def test_generate_mask_big_tracked_set_takes_32_row_chunks():
    """Past ~1 100 rows at 96x160 the launch walks 32 rows per workgroup instead of 8 (the prototypes of a pixel block are read
    once per chunk): same values bit for bit as the 8-row launches of the same rows, rows of several clips in one chunk, the
    bit words included; spot-checked against the oracle."""
    gen = torch.Generator().manual_seed(9)
    n, P = 1150, 5
    protos = torch.relu(torch.randn(P, 96, 160, 32, generator=gen))
    coeff = torch.randn(n, 32, generator=gen)
    c = torch.rand(n, 2, generator=gen)
    wh = torch.rand(n, 2, generator=gen) * 0.5
    box = torch.cat([c - wh / 2, c + wh / 2], 1)
    rp = torch.sort(torch.randint(0, P, (n,), generator=gen)).values.to(torch.int32)
    big, big_bits = ops.lincomb_sigmoid_crop_bits(protos.to(DEV), coeff.to(DEV), box.to(DEV), rp.to(DEV))
    for lo in range(0, n, 500):                       # 500 rows per launch: 8-row chunks
        hi = min(n, lo + 500)
        part, part_bits = ops.lincomb_sigmoid_crop_bits(protos.to(DEV), coeff[lo:hi].to(DEV), box[lo:hi].to(DEV), rp[lo:hi].to(DEV))
        assert torch.equal(part, big[lo:hi]) and torch.equal(part_bits, big_bits[lo:hi])
    for r in (0, 31, 32, 577, n - 1):
        ref = oracle.generate_mask(protos[int(rp[r])], coeff[r:r + 1], box[r:r + 1])
        assert (big[r].cpu() - ref[0]).abs().max() < 5e-6


@pytest.mark.parametrize("hw", [(96, 160), (23, 41), (24, 40)])
def test_generate_mask_rows_that_miss_a_pixel_block_are_zero_filled(hw):
    """lincomb_kernel zero-fills the rows whose crop rectangle misses a workgroup's 256 pixels with 16-byte stores instead of walking them (h*w a
    multiple of 4; any other size keeps the pixel loop): small boxes, the output buffers poisoned first (torch.empty re-uses the freed block), soft
    masks against the oracle and the bit words against the binarised soft masks."""
    H, W = hw
    gen = torch.Generator().manual_seed(21)
    n, P = 90, 3
    protos = torch.relu(torch.randn(P, H, W, 32, generator=gen))
    coeff = torch.randn(n, 32, generator=gen)
    c = torch.rand(n, 2, generator=gen)
    wh = torch.rand(n, 2, generator=gen) * 0.25
    box = torch.cat([c - wh / 2, c + wh / 2], 1)
    box[7] = torch.tensor([0.0, 0.0, 1.0, 1.0])          # one row covering everything
    box[8] = torch.tensor([0.3, 0.3, 0.3, 0.3])          # one degenerate box
    rp = torch.sort(torch.randint(0, P, (n,), generator=gen)).values.to(torch.int32)
    words = (H * W + 63) // 64
    poison_m = torch.full((n, H, W), float("nan"), device=DEV)
    poison_b = torch.full((n, words), -1, dtype=torch.int64, device=DEV)
    del poison_m, poison_b
    got, bits = ops.lincomb_sigmoid_crop_bits(protos.to(DEV), coeff.to(DEV), box.to(DEV), rp.to(DEV))
    got, bits = got.cpu(), bits.cpu()
    assert not torch.isnan(got).any()
    for pi in range(P):
        sel = torch.nonzero(rp == pi).view(-1)
        ref = oracle.generate_mask(protos[pi], coeff[sel], box[sel])
        assert (got[sel] - ref).abs().max() < 5e-6 and torch.equal(got[sel] == 0, ref == 0)
    flat = (got.view(n, -1) > 0.5)
    pad = torch.zeros(n, words * 64, dtype=torch.bool)
    pad[:, :H * W] = flat
    weights = (torch.ones(64, dtype=torch.int64) << torch.arange(64, dtype=torch.int64))      # bit i of a word = pixel 64 word + i
    want = (pad.view(n, words, 64).to(torch.int64) * weights).sum(-1)
    assert torch.equal(bits, want)


@pytest.mark.parametrize("p", CASES)
def test_mask_iou_bit_exact(golden_postproc, p):
    m = golden_postproc[p + "masks"]
    got = ops.mask_iou(m[: min(20, len(m))].to(DEV), m.to(DEV), 0.5).cpu()
    assert torch.equal(got, golden_postproc[p + "mask_iou"])
    # hw not a multiple of 64, empty masks -> union 0 -> 0
    a = (torch.rand(3, 7, 11) > 0.5).float()
    a[1] = 0
    assert torch.equal(ops.mask_iou(a.to(DEV), a.to(DEV)).cpu(), oracle.mask_iou(a, a))


@pytest.mark.parametrize("fmt", ["nchw", "nhwc"])
def test_bias_act_epilogue(fmt):
    y = rnd(3, 64, 12, 20, seed=1)
    bias, res = rnd(64, seed=2), rnd(3, 64, 12, 20, seed=3)
    mf = torch.channels_last if fmt == "nhwc" else torch.contiguous_format
    for use_res in (False, True):
        for relu in (False, True):
            ref = y + bias.view(1, -1, 1, 1) + (res if use_res else 0)
            ref = ref.clamp(min=0) if relu else ref
            yd = y.to(DEV).contiguous(memory_format=mf).clone()
            rd = res.to(DEV).contiguous(memory_format=mf) if use_res else None
            out = ops.bias_act_(yd, bias.to(DEV), rd, relu)
            assert out.data_ptr() == yd.data_ptr()
            assert torch.equal(out.cpu(), ref), (fmt, use_res, relu)   # same IEEE ops, same order: bit-exact
    # residual in the other memory format is converted, not misread
    yd = y.to(DEV).contiguous(memory_format=mf).clone()
    other = torch.contiguous_format if fmt == "nhwc" else torch.channels_last
    out = ops.bias_act_(yd, bias.to(DEV), res.to(DEV).contiguous(memory_format=other), True)
    assert torch.equal(out.cpu(), (y + bias.view(1, -1, 1, 1) + res).clamp(min=0))


def test_bad_arguments_raise():
    x = torch.zeros(1, 8, 8, 8, device=DEV)
    with pytest.raises(StmError):
        ops.deform_im2col(x, torch.zeros(1, 17, 8, 8, device=DEV), None, 3, 1, 1)      # wrong offset channels
    with pytest.raises(StmError):
        ops.corr_patch(x, torch.zeros(1, 8, 8, 9, device=DEV), 11)                      # shape mismatch
    with pytest.raises(StmError):
        ops.corr_patch(x, x, 4)                                                         # even patch size
    with pytest.raises(StmError):
        ops.cc_fast_nms(torch.zeros(4, 41, device=DEV), torch.zeros(4, 4, device=DEV), None, 0.5, 4096)


@pytest.mark.parametrize("B,C,H,W,stride", [(2, 128, 12, 20, 1), (1, 256, 9, 13, 2), (2, 512, 6, 10, 1), (8, 128, 48, 80, 1)])
def test_dcn_sample_planar_equals_im2col(B, C, H, W, stride):
    """The planar deformable sampler (NHWC in, pixel-major offsets, bf16-plane columns with K = tap*C + channel) holds
    exactly the values of stm_deform_im2col_f32 on the same inputs, and through them the oracle's."""
    x = rnd(B, C, H, W, seed=C)
    Ho, Wo = ops.conv_out_hw(H, W, 3, 3, stride, stride, 1, 1, 1, 1)
    om = rnd(B, 27, Ho, Wo, seed=C + 1, scale=1.5)                   # raw conv_offset_mask output (mask logits last)
    cols = ops.deform_im2col(x.to(DEV), None, None, 3, stride, 1, 1, 1, fused_om=om.to(DEV))       # [B, C*9, Ho*Wo], row c*9 + k
    ref = cols.view(B, C, 9, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, 9 * C)           # -> [pixel, k*C + c]
    pl = ops.dcn_sample_planar(x.permute(0, 2, 3, 1).contiguous().to(DEV), om.permute(0, 2, 3, 1).reshape(-1, 27).to(DEV), stride, 1, 1)
    assert pl.shape == (3, 9 * C // 32, B * Ho * Wo, 32)
    assert torch.equal(ops.planes_to_f32(pl), ref)
    # fp16 two-plane output: the same planes stm_split_planes_fmt_f32 makes of those values
    ph = ops.dcn_sample_planar(x.permute(0, 2, 3, 1).contiguous().to(DEV), om.permute(0, 2, 3, 1).reshape(-1, 27).to(DEV), stride, 1, 1,
                               fmt=1)
    assert ph.shape == (2, 9 * C // 32, B * Ho * Wo, 32) and ph.dtype == torch.float16
    assert torch.equal(ph, ops.split_planes(ref.contiguous(), fmt=1).view_as(ph))
    if B * H * W <= 1000:
        o_cols = oracle.deform_im2col(x, om[:, 0:18].contiguous(), torch.sigmoid(om[:, 18:27]), (3, 3), stride, 1, 1, 1)
        assert (cols.cpu() - o_cols).abs().max().item() < 2e-5


@pytest.mark.parametrize("kh,kw", [(3, 3), (3, 5), (5, 3)])
def test_deform_sample_planar_mask_free_equals_im2col(kh, kw):
    """stm_deform_sample_planar_f32 without mask (mmcv DeformConv2d of FeatureAlign: 3x3 / 3x5 / 5x3 taps, C = 256) on a
    channel slice of a wider pixel-major tensor, written at a pixel offset of a shared column buffer: the values of
    stm_deform_im2col_f32 (v1, no mask) on the same inputs, bit for bit."""
    B, C, H, W, wide = 2, 256, 7, 9, 384
    K = kh * kw
    xw = rnd(B * H * W, wide, seed=kh)                                # pixel-major, 384 channels; the layer reads [64, 320)
    x = xw[:, 64:64 + C]
    off = rnd(B * H * W, 2 * K, seed=kw + 7, scale=1.5)               # pixel-major offsets (dy, dx per tap)
    x_nchw = x.reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous()
    off_nchw = off.reshape(B, H, W, 2 * K).permute(0, 3, 1, 2).contiguous()
    cols = ops.deform_im2col(x_nchw.to(DEV), off_nchw.to(DEV), None, (kh, kw), 1, ((kh - 1) // 2, (kw - 1) // 2), 1, 1)   # [B, C*K, H*W]
    ref = cols.view(B, C, K, H * W).permute(0, 3, 2, 1).reshape(B * H * W, K * C)                  # [pixel, k*C + c]
    for fmt in (0, 1):
        ntot, pix0 = B * H * W + 37, 21
        out = torch.zeros(2 if fmt == 1 else 3, K * C // 32, ntot, 32, device=DEV, dtype=torch.float16 if fmt == 1 else torch.bfloat16)
        xd = xw.to(DEV)
        ops.deform_sample_planar(xd[:, 64:64 + C], B, H, W, C, off.to(DEV), (kh, kw), ((kh - 1) // 2, (kw - 1) // 2), out, pix0, fmt)
        got = out[:, :, pix0:pix0 + B * H * W].contiguous()
        assert torch.equal(got, ops.split_planes(ref.contiguous(), fmt=fmt).view_as(got)), (kh, kw, fmt)
        assert torch.count_nonzero(out[:, :, :pix0]) == 0 and torch.count_nonzero(out[:, :, pix0 + B * H * W:]) == 0


def test_mask_iou_grouped_equals_full_masked_by_group():
    """stm_mask_iou_grouped_f32 (the batched pipeline's form: only pairs of the same clip) == stm_mask_iou_f32 with the
    pairs of different groups set to 0; group sizes that straddle the 256-column blocks, empty groups included."""
    g = torch.Generator().manual_seed(3)
    n1, n2, h, w = 37, 700, 24, 40
    m1, m2 = torch.rand(n1, h, w, generator=g), torch.rand(n2, h, w, generator=g)
    g2 = torch.sort(torch.randint(0, 9, (n2,), generator=g)).values.to(torch.int32)
    g2[g2 == 4] = 5                                   # an empty group in the middle
    g1 = torch.randint(0, 9, (n1,), generator=g).to(torch.int32)
    full = ops.mask_iou(m1.to(DEV), m2.to(DEV)).cpu()
    got = ops.mask_iou(m1.to(DEV), m2.to(DEV), group1=g1.to(DEV), group2=g2.to(DEV)).cpu()
    same = g1[:, None] == g2[None, :]
    assert torch.equal(got, torch.where(same, full, torch.zeros_like(full)))
    assert same.any() and (~same).any()
    # the C entry promises nothing about the order of the groups: shuffled columns give the same pairs
    perm = torch.randperm(n2, generator=g)
    got_p = ops.mask_iou(m1.to(DEV), m2[perm].to(DEV), group1=g1.to(DEV), group2=g2[perm].to(DEV)).cpu()
    assert torch.equal(got_p, got[:, perm])


@pytest.mark.parametrize("scale", [0.5, 6.0])
@pytest.mark.parametrize("B,C,H,W,stride", [(2, 128, 21, 37, 1), (2, 128, 30, 45, 2), (1, 256, 17, 16, 1)])
def test_dcn_sample_planar_lds_form_equals_register_gather(B, C, H, W, stride, scale, tunables):
    """The LDS-staged sampler (STM_DCN_LDS=1: tiles of 4 x 16 / 2 x 16 output pixels, 32-channel chunks, corners gathered from LDS; pairs whose
    corners leave the staged rectangle gathered from global memory) against the register-gather kernel: the same planes bit for bit,
    with offsets inside the halo (scale 0.5) and far outside it (scale 6: most pairs take the global path), partial tiles on both
    axes, both strides."""
    x = rnd(B, H, W, C, seed=C + H)
    Ho, Wo = ops.conv_out_hw(H, W, 3, 3, stride, stride, 1, 1, 1, 1)
    om = rnd(B * Ho * Wo, 27, seed=W, scale=scale)
    outs = {}
    for lds in ("1", "0"):
        tunables.set(STM_DCN_LDS=lds)
        outs[lds] = ops.dcn_sample_planar(x.to(DEV), om.to(DEV), stride, 1, 1, fmt=1).cpu()
    tunables.clear("STM_DCN_LDS")
    assert torch.equal(outs["1"], outs["0"])
    assert outs["1"].abs().sum().item() > 0


def test_detect_cc_on_logits_equals_softmax_then_detect():
    """stm_detect_cc_logits_f32 (softmax folded into the candidate pass) against F.softmax + stm_detect_cc_f32: the same prior indices,
    classes and boxes; scores within 2 ulp of a probability (the row sum is taken in another order than torch's softmax kernel)."""
    g = torch.Generator().manual_seed(5)
    B, N, ncls = 3, 15345, 41
    logits = torch.randn(B, N, ncls, generator=g) * 1.5
    logits[..., 0] += 5.0                                      # background-dominated, a few hundred candidates per frame
    loc = torch.randn(B, N, 4, generator=g) * 0.3
    pri = torch.rand(N, 4, generator=g) * 0.5 + 0.25
    cen = torch.rand(B, N, generator=g)
    a = ops.detect_cc(loc.to(DEV), pri.to(DEV), torch.softmax(logits.to(DEV), -1), cen.to(DEV), 0.05, 0.5, 200)
    b = ops.detect_cc(loc.to(DEV), pri.to(DEV), logits.to(DEV), cen.to(DEV), 0.05, 0.5, 200, logits=True)
    cnt = a[4].cpu()
    assert int(cnt.min()) > 5
    agree = 0
    for f in range(B):
        n = int(cnt[f])
        if int(b[4][f]) == n and torch.equal(a[0][f, :n], b[0][f, :n]):
            agree += 1
            assert torch.equal(a[1][f, :n], b[1][f, :n]) and torch.equal(a[3][f, :n], b[3][f, :n])
            assert (a[2][f, :n] - b[2][f, :n]).abs().max().item() < 3e-7
    assert agree == B      # (a score within 1 ulp of the 0.05 threshold or of a neighbour in the sort could differ; not with this seed)


def test_detect_pc_batched_equals_per_frame_per_class_nms():
    """stm_fast_nms_batched_f32 through the candidate pass's keep lists (ops.detect_pc: one launch pair for all frames, each frame's sort sized by its
    own candidate count) == stm_generate_candidates_f32 + stm_fast_nms_f32 frame by frame on gathered rows: indices, classes, scores, boxes, counts."""
    g = torch.Generator().manual_seed(11)
    B, N, ncls = 5, 3000, 41
    priors = torch.rand(N, 4, generator=g) * 0.5 + 0.1
    priors[:, 2:] = priors[:, 2:] * 0.3 + 0.02
    loc = torch.randn(B, N, 4, generator=g) * 0.5
    logits = torch.randn(B, N, ncls, generator=g)
    logits[:, :, 0] += 9.0                                   # background wins everywhere ...
    hot = torch.rand(B, N, generator=g) < torch.tensor([0.02, 0.2, 0.0, 0.6, 0.05]).view(B, 1)   # very different candidate counts, one empty frame
    cls_hot = torch.randint(1, ncls, (B, N), generator=g)
    logits[hot, cls_hot[hot]] += 8.0                         # ... but on the hot rows
    conf = torch.softmax(logits, -1)
    cen = torch.rand(B, N, generator=g)
    d = lambda t: t.to(DEV)
    idx, cls, sc, bx, cnt = ops.detect_pc(d(loc), d(priors), d(conf), d(cen), 0.05, 0.5, 200, 100)
    keep_idx, cand_box, count = ops.generate_candidates(d(loc), d(priors), d(conf), 0.05)
    counts = count.tolist()
    assert counts[2] == 0 and max(counts) > 1000 and cnt.tolist()[2] == 0
    for b in range(B):
        k = counts[b]
        if k == 0:
            continue
        rows = keep_idx[b, :k]
        i1, c1, s1, b1, n1 = ops.fast_nms(d(conf)[b].index_select(0, rows), cand_box[b, :k], d(cen)[b].index_select(0, rows), 0.5, 200, 0.05, 100)
        n = int(n1)
        assert int(cnt[b]) == n and n > 0
        assert torch.equal(idx[b, :n], rows[i1[:n]]) and torch.equal(cls[b, :n], c1[:n])
        assert torch.equal(sc[b, :n], s1[:n]) and torch.equal(bx[b, :n], b1[:n])


def test_mask_iou_row_block_kernel_equals_small_kernel(tunables):
    """mask_iou_pairs_kernel (the row-block form the library takes from 4 096 rows: eight staged rows per workgroup, matched-column list, DPP sums)
    forced onto small problems (STM_MIOU_BIG_ROWS=1) against the one-thread-per-column kernel: same bits -- with and without groups, row counts that
    leave a partial last block (n1 % 8 != 0), a group with more than 2 048 columns (the list's overflow path), empty groups, long rows (920 words:
    six staged rows per workgroup)."""
    g = torch.Generator().manual_seed(9)
    cases = [(37, 700, 24, 40, 9), (13, 2600, 12, 20, 2), (5, 300, 184, 320, 3)]          # n1, n2, h, w, groups
    for n1, n2, h, w, ng in cases:
        m1, m2 = torch.rand(n1, h, w, generator=g).to(DEV), torch.rand(n2, h, w, generator=g).to(DEV)
        g2 = torch.sort(torch.randint(0, ng, (n2,), generator=g)).values.to(torch.int32)
        if ng > 4:
            g2[g2 == 4] = 5                                    # an empty group
        if ng == 2:
            g2[:2300] = 0                                      # > 2 048 columns of one group against every block of group-0 rows
        g1 = torch.randint(0, ng, (n1,), generator=g).to(torch.int32)
        outs = {}
        for big in ("100000", "1"):
            tunables.set(STM_MIOU_BIG_ROWS=big)
            outs[big] = (ops.mask_iou(m1, m2).cpu(), ops.mask_iou(m1, m2, group1=g1.to(DEV), group2=g2.to(DEV)).cpu())
            tunables.clear("STM_MIOU_BIG_ROWS")
        assert torch.equal(outs["1"][0], outs["100000"][0]), (n1, n2, "all pairs")
        assert torch.equal(outs["1"][1], outs["100000"][1]), (n1, n2, "grouped")
        same = g1[:, None] == g2[None, :]
        assert torch.equal(outs["1"][1], torch.where(same, outs["1"][0], torch.zeros_like(outs["1"][0])))
