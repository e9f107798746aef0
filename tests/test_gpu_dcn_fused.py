"""GPU parity of the fused deformable convolution (csrc/dcn_fused.hip, stm_deform_conv_fused_planar_f32: sampler -> fp16 plane
split -> MFMA product in one kernel, no column buffer) -- rows a1 / a7 of SURVEY.md section 8:
  * against the fp64 CPU oracle of dcn_v2.DCN / mmcv DeformConv2d (oracle.deform_conv) on seeded inputs, tolerance stated as for
    the planar convolutions: |y - y_fp64| <= 2e-6 * sum |x w| (+ the bilinear blend's own fp32 rounding, 2e-6 * the same sum);
  * against the kernel pair it replaces (stm_deform_sample_planar_f32 + stm_conv2d_planar_f32 over taps * C channels) on the same
    inputs: same sampled values, same plane products -- within fp32 summation order (2e-6 of the magnitude sum), at the 7 layer
    shapes of R50 @384x640 too;
  * known answers: zero offsets + zero mask logits = half the dense convolution (the reference's zero-init state,
    backbone.py:24-26), border pixels included.
"""
import pytest
import torch
import torch.nn.functional as F

import oracle
from stmask_amd import ops, _lib
from stmask_amd._lib import StmError

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def fused(x_nchw, om_nchw, w, bias, stride, pad, has_mask, relu, fmt=1, out_fmt=None):
    """x [B, C, H, W], om [B, 2K (+K), Ho, Wo] (raw offsets, then mask LOGITS) -> fp32 [B*Ho*Wo, O] through the fused kernel."""
    B, C, H, W = x_nchw.shape
    O, _, kh, kw = w.shape
    x_pix = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    om = om_nchw.permute(0, 2, 3, 1).reshape(-1, om_nchw.shape[1]).contiguous().to(DEV)
    packed, out_scale = ops.conv_pack_weights(w.to(DEV), tile_n=128, fmt=fmt)
    pl = ops.deform_conv_fused_planar(x_pix, B, H, W, C, om, packed, out_scale, None if bias is None else bias.to(DEV), O, (kh, kw), stride, pad, 1,
                                      has_mask=has_mask, relu=relu, fmt=fmt, out_fmt=out_fmt)
    return ops.planes_to_f32(pl).cpu()


def pair(x_nchw, om_nchw, w, bias, stride, pad, has_mask, relu, fmt=1):
    """The same layer as the kernel pair the fused kernel replaces: planar sampler (columns as planes) + planar 1x1 product."""
    B, C, H, W = x_nchw.shape
    O, _, kh, kw = w.shape
    K = kh * kw
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, stride, pad[0], pad[1], 1, 1)
    x_pix = x_nchw.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    om = om_nchw.permute(0, 2, 3, 1).reshape(-1, om_nchw.shape[1]).contiguous().to(DEV)
    if has_mask:
        cols = ops.dcn_sample_planar(x_pix.view(B, H, W, C), om, stride, pad, 1, fmt=fmt)
    else:
        cols = torch.zeros(ops.plane_layout(fmt)[0], K * C // 32, B * H * W, 32, device=DEV, dtype=ops.plane_layout(fmt)[1])
        ops.deform_sample_planar(x_pix, B, H, W, C, om, (kh, kw), pad, cols, 0, fmt)
    wk = w.permute(0, 2, 3, 1).reshape(O, K * C, 1, 1).contiguous().to(DEV)
    packed, out_scale = ops.conv_pack_weights(wk, tile_n=128, fmt=fmt)
    y = ops.conv2d_planar(cols, packed, (O, K * C, 1, 1), (B, Ho, Wo), None if bias is None else bias.to(DEV), None, stride=1, padding=0, relu=relu,
                          out="f32", tile_n=128, fmt=fmt, out_scale=out_scale)
    return y.cpu()


FUSED_CASES = [
    # B, C,  H,  W, O, kh, kw, s, (ph, pw), off_scale, mask, relu
    (2, 128, 12, 20, 128, 3, 3, 1, (1, 1), 1.5, True, True),       # stride 1, one tile row of patches
    (1, 128, 24, 40, 128, 3, 3, 2, (1, 1), 2.0, True, True),       # stride 2 (L2.0-like), odd patch grid
    (3, 64, 9, 13, 256, 3, 3, 1, (1, 1), 3.0, True, False),        # odd sizes, two channel tiles, no ReLU
    (1, 64, 16, 16, 128, 3, 3, 1, (1, 1), 12.0, True, True),       # offsets far outside the image (zero corners, clamped addresses)
    (2, 256, 7, 9, 256, 3, 5, 1, (1, 2), 1.5, False, True),        # FCB 3x5, no mask, no bias (the class branch's own shape)
    (1, 64, 6, 10, 128, 5, 3, 1, (2, 1), 1.5, False, False),       # FCB 5x3
    (1, 64, 3, 5, 128, 3, 3, 1, (1, 1), 1.0, False, True),         # P7: 15 pixels, one partial patch
]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_deform_conv_vs_oracle_and_pair(case):
    B, C, H, W, O, kh, kw, s, pad, osc, wm, relu = case
    K = kh * kw
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, s, s, pad[0], pad[1], 1, 1)
    x = rnd(B, C, H, W, seed=3)
    off = rnd(B, 2 * K, Ho, Wo, seed=4, scale=osc)
    logit = rnd(B, K, Ho, Wo, seed=5) if wm else None
    om = torch.cat([off, logit], 1) if wm else off
    w = rnd(O, C, kh, kw, seed=11, scale=(C * K) ** -0.5)
    bias = rnd(O, seed=12) if wm else None
    mask = torch.sigmoid(logit) if wm else None
    ref = oracle.deform_conv(x, off, mask, w, bias, s, pad, 1, 1)                                   # [B, O, Ho, Wo], fp64 accumulation
    mag = oracle.deform_conv(x.abs(), off, mask, w.abs(), None if bias is None else bias.abs(), s, pad, 1, 1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, O)
    mag = mag.permute(0, 2, 3, 1).reshape(-1, O)
    if relu:
        ref = ref.clamp_min(0)
    n0 = _lib.lib().stm_debug_launch_count(1)
    got = fused(x, om, w, bias, s, pad, wm, relu)
    assert _lib.lib().stm_debug_launch_count(1) == n0 + 1
    assert got.shape == ref.shape
    err = ((got - ref).abs() / (mag + 1e-3)).max().item()
    assert err < 4e-6, err                     # stated: 2e-6 (plane products) + 2e-6 (fp32 bilinear blend) of sum |x w|
    assert (got - ref).abs().max().item() < 1e-4   # north-star tolerance, absolute
    if C in (128, 256, 512) and (wm or C == 256):                   # (the shapes the unfused sampler is built for)
        two = pair(x, om, w, bias, s, pad, wm, relu)
        assert ((got - two).abs() / (mag + 1e-3)).max().item() < 2e-6   # same values and products, another summation order


@pytest.mark.parametrize("shape", [(2, 128, 14, 18, 256, 1), (1, 256, 24, 40, 512, 2), (2, 64, 5, 7, 256, 1)])
def test_fused_deform_conv_wide_tiles_equal_narrow_tiles(shape, tunables):
    """Layers with Cout a multiple of 256 run on 64-pixel x 256-channel tiles (every pixel sampled once); STM_DCN_FUSED_WIDE=0 keeps the 128 x 128 tiles.
    Same sampled values, same products in the same order per accumulator: the two must agree bit for bit."""
    B, C, H, W, O, s = shape
    Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
    x = rnd(B, C, H, W, seed=31)
    om = torch.cat([rnd(B, 18, Ho, Wo, seed=32, scale=2.0), rnd(B, 9, Ho, Wo, seed=33)], 1)
    w = rnd(O, C, 3, 3, seed=34, scale=(C * 9) ** -0.5)
    bias = rnd(O, seed=35)
    wide = fused(x, om, w, bias, s, (1, 1), True, True)
    tunables.set(STM_DCN_FUSED_WIDE=0)
    narrow = fused(x, om, w, bias, s, (1, 1), True, True)
    assert torch.equal(wide, narrow)


def test_fused_deform_conv_dilation_two():
    """dilation 2 (the geometry struct's dh / dw; the reference's DCN keeps dilation 1, mmcv's DeformConv2d takes any): fused kernel vs the fp64 oracle."""
    B, C, H, W, O, s, pad, dil = 2, 64, 11, 14, 128, 1, (2, 2), 2
    Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, pad[0], pad[1], dil, dil)
    x = rnd(B, C, H, W, seed=3)
    off = rnd(B, 18, Ho, Wo, seed=4, scale=1.5)
    logit = rnd(B, 9, Ho, Wo, seed=5)
    w = rnd(O, C, 3, 3, seed=11, scale=(C * 9) ** -0.5)
    bias = rnd(O, seed=12)
    ref = oracle.deform_conv(x, off, torch.sigmoid(logit), w, bias, s, pad, dil, 1).permute(0, 2, 3, 1).reshape(-1, O)
    mag = oracle.deform_conv(x.abs(), off, torch.sigmoid(logit), w.abs(), bias.abs(), s, pad, dil, 1).permute(0, 2, 3, 1).reshape(-1, O)
    x_pix = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)
    om = torch.cat([off, logit], 1).permute(0, 2, 3, 1).reshape(-1, 27).contiguous().to(DEV)
    packed, out_scale = ops.conv_pack_weights(w.to(DEV), tile_n=128, fmt=1)
    pl = ops.deform_conv_fused_planar(x_pix, B, H, W, C, om, packed, out_scale, bias.to(DEV), O, (3, 3), s, pad, dil, has_mask=True, relu=False, fmt=1)
    got = ops.planes_to_f32(pl).cpu()
    assert got.shape == ref.shape
    assert ((got - ref).abs() / (mag + 1e-3)).max().item() < 4e-6


@pytest.mark.parametrize("fmt", [1, 2])
def test_fused_deform_conv_planes_formats(fmt):
    """fp16 x 1 (BASELINE config 5's backbone format): one plane, one product, stated tolerance 1e-3 of sum |x w|; a format-2 layer may hand
    both planes to a format-1 consumer (out_fmt 1).  fmt 1 runs the same case at its own tolerance."""
    B, C, H, W, O, s = 2, 128, 12, 20, 128, 1
    x = rnd(B, C, H, W, seed=1)
    om = torch.cat([rnd(B, 18, H, W, seed=2, scale=2.0), rnd(B, 9, H, W, seed=3)], 1)
    w = rnd(O, C, 3, 3, seed=4, scale=(9 * C) ** -0.5)
    bias = rnd(O, seed=5)
    ref = oracle.deform_conv(x, om[:, :18].contiguous(), torch.sigmoid(om[:, 18:]), w, bias, s, (1, 1), 1, 1).permute(0, 2, 3, 1).reshape(-1, O).clamp_min(0)
    mag = oracle.deform_conv(x.abs(), om[:, :18].contiguous(), torch.sigmoid(om[:, 18:]), w.abs(), bias.abs(), s, (1, 1), 1, 1).permute(0, 2, 3, 1).reshape(-1, O)
    tol = 4e-6 if fmt == 1 else 1e-3
    got = fused(x, om, w, bias, s, (1, 1), True, True, fmt=fmt)
    assert ((got - ref).abs() / (mag + 1e-3)).max().item() < tol
    if fmt == 2:
        got1 = fused(x, om, w, bias, s, (1, 1), True, True, fmt=2, out_fmt=1)      # both planes of the fp32 result: closer to it than one plane can be
        assert ((got1 - ref).abs() / (mag + 1e-3)).max().item() < tol
        assert (got1 - got).abs().max().item() <= 2.0 ** -10 * got1.abs().max().item()


def test_fused_deform_conv_known_answers():
    """Zero offsets and zero mask logits (the reference's zero-initialised conv_offset_mask, backbone.py:24-26): the deformable convolution is
    exactly half the dense one, borders included; offsets of one whole pixel = the dense convolution of the shifted image (interior)."""
    B, C, H, W, O = 2, 64, 10, 14, 128
    x = rnd(B, C, H, W, seed=21)
    w = rnd(O, C, 3, 3, seed=22, scale=0.05)
    for s in (1, 2):
        Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
        om = torch.zeros(B, 27, Ho, Wo)
        dense = F.conv2d(x.double(), w.double(), None, s, 1).permute(0, 2, 3, 1).reshape(-1, O)
        got = fused(x, om, w, None, s, (1, 1), True, False)
        assert (got.double() - 0.5 * dense).abs().max().item() < 2e-5
    om = torch.zeros(B, 27, H, W)
    om[:, 1:18:2] = 1.0                                   # dx = +1 on every tap
    om[:, 18:] = 30.0                                     # sigmoid -> 1.0f
    shifted = torch.zeros_like(x)
    shifted[..., :-1] = x[..., 1:]
    dense = F.conv2d(shifted.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(B, H, W, O)
    got = fused(x, om, w, None, 1, (1, 1), True, False).reshape(B, H, W, O)
    # (column 0 differs by construction: the dense convolution pads where the shifted sample x[-1 + 1] exists)
    assert (got[:, :, 1:-2].double() - dense[:, :, 1:-2]).abs().max().item() < 2e-5


def test_fused_deform_conv_channel_slice_and_pixel_offset():
    """x as a channel slice of a wider pixel-major tensor (the FCB class branch reads the towers' concatenated output) and the result written
    at a pixel offset of a larger plane buffer (the FPN levels of a shared head share one): only that window changes."""
    B, C, H, W, O, wide = 2, 64, 7, 9, 128, 192
    kh, kw, pad = 3, 5, (1, 2)
    K = kh * kw
    xw = rnd(B * H * W, wide, seed=31).to(DEV)
    x_nchw = xw[:, 64:64 + C].reshape(B, H, W, C).permute(0, 3, 1, 2).contiguous().cpu()
    off = rnd(B, 2 * K, H, W, seed=32, scale=1.5)
    w = rnd(O, C, kh, kw, seed=33, scale=(C * K) ** -0.5)
    ref = fused(x_nchw, off, w, None, 1, pad, False, True)
    packed, out_scale = ops.conv_pack_weights(w.to(DEV), tile_n=128, fmt=1)
    ntot, pix0 = B * H * W + 37, 21
    out = torch.full((2, O // 32, ntot, 32), 7.0, device=DEV, dtype=torch.float16)
    om = off.permute(0, 2, 3, 1).reshape(-1, 2 * K).contiguous().to(DEV)
    ops.deform_conv_fused_planar(xw[:, 64:64 + C], B, H, W, C, om, packed, out_scale, None, O, (kh, kw), 1, pad, 1, has_mask=False, relu=True, fmt=1,
                                 out=out, out_off=pix0)
    got = ops.planes_to_f32(out[:, :, pix0:pix0 + B * H * W].contiguous()).cpu()
    assert torch.equal(got, ref)
    assert bool((out[:, :, :pix0] == 7.0).all()) and bool((out[:, :, pix0 + B * H * W:] == 7.0).all())


def test_fused_deform_conv_full_size_layers_match_the_pair():
    """The 7 DCN layer shapes of R50 @384x640 at batch 2: fused kernel against the sampler + product pair."""
    shapes = [(128, 96, 160, 2), (128, 48, 80, 1), (256, 48, 80, 2), (256, 24, 40, 1), (512, 24, 40, 2), (512, 12, 20, 1)]
    for C, H, W, s in shapes:
        B = 2
        Ho, Wo = ops.conv_out_hw(H, W, 3, 3, s, s, 1, 1, 1, 1)
        x = rnd(B, C, H, W, seed=C + s).clamp_min(0)                 # (conv1's output is a ReLU's)
        om = torch.cat([rnd(B, 18, Ho, Wo, seed=C + 1, scale=2.0), rnd(B, 9, Ho, Wo, seed=C + 2)], 1)
        w = rnd(C, C, 3, 3, seed=C + 3, scale=(9 * C) ** -0.5)
        bias = rnd(C, seed=C + 4)
        got = fused(x, om, w, bias, s, (1, 1), True, True)
        two = pair(x, om, w, bias, s, (1, 1), True, True)
        scale = two.abs().max().item()
        assert (got - two).abs().max().item() < 4e-6 * max(scale, 1.0), (C, H, W, s)


def test_fused_deform_conv_rejects_bad_arguments():
    x = rnd(1 * 8 * 8, 64).to(DEV)
    om = rnd(64, 27).to(DEV)
    w = rnd(128, 64, 3, 3, scale=0.05).to(DEV)
    packed, sc = ops.conv_pack_weights(w, tile_n=128, fmt=1)
    with pytest.raises(StmError):
        ops.deform_conv_fused_planar(x.cpu(), 1, 8, 8, 64, om, packed, sc, None, 128)                      # CPU tensor: no fallback
    with pytest.raises(StmError):
        ops.deform_conv_fused_planar(x, 1, 8, 8, 64, om[:, :20].contiguous(), packed, sc, None, 128)        # too few offset / mask channels
    with pytest.raises(StmError):
        ops.deform_conv_fused_planar(x, 1, 8, 8, 64, om, packed, sc, None, 96)                              # Cout not a multiple of 128
    with pytest.raises(StmError):
        ops.deform_conv_fused_planar(x, 1, 8, 8, 64, om, packed, sc, None, 128, fmt=0)                      # bf16 x 3: the pair stays
    assert not ops.deform_conv_fused_supported(48, 128, 3, True, 1) and ops.deform_conv_fused_supported(256, 256, (3, 5), False, 1)


@pytest.mark.parametrize("O", [128, 256])                     # 128 x 128 and 64 x 256 tiles
def test_fused_deform_conv_raises_the_fp16_range_flag(O):
    """A sampled value beyond fp16's range (|v| > 65504, inf, nan) must raise the sticky range flag even though the layer's ReLU would turn a nan output
    into 0: since round 6 the fused kernel tests the PRE-activation outputs (a sample whose h plane is inf makes every product it enters inf or nan) and
    its producer waves carry no range bookkeeping.  Also with weights that are all zero (0 * inf = nan), and not for in-range inputs."""
    flag = ops.planar_range_flag()
    flag.zero_()

    def raised():
        torch.cuda.synchronize()
        v = int(flag.item())
        flag.zero_()
        return v

    B, C, H, W = 1, 64, 8, 8
    x = rnd(B, C, H, W, seed=3)
    om = torch.cat([torch.zeros(B, 18, H, W), torch.full((B, 9, H, W), 10.0)], 1)      # zero offsets, mask ~ 1: the nine taps sample x itself
    w = rnd(O, C, 3, 3, seed=6, scale=0.05)
    y = fused(x, om, w, None, 1, (1, 1), True, True)
    assert raised() == 0 and torch.isfinite(y).all()
    for bad in (1e5, -3e38, float("inf"), float("-inf"), float("nan")):
        xb = x.clone()
        xb[0, 7, 4, 4] = bad
        for wt in (w, torch.zeros_like(w), -w.abs()):             # (-|w|: the ReLU would hide a -inf / nan output)
            fused(xb, om, wt, None, 1, (1, 1), True, True)
            assert raised() == 1, (bad, float(wt.abs().max()))
    # a value fp16 can hold leaves the flag alone
    xb = x.clone()
    xb[0, 7, 4, 4] = 6.0e4
    fused(xb, om, w, None, 1, (1, 1), True, True)
    assert raised() == 0
