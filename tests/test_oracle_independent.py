"""Independent anchors for the four third-party restatements of the oracle (CPU, no GPU).

The reference binds its deformable convolutions, RoIAlign and correlation to un-vendored CUDA packages (dcn_v2, mmcv-full 1.1.2,
spatial-correlation-sampler: SURVEY.md section 8(c), "parity unpinned"), and the goldens of tests/golden/ were captured with THIS repository's
oracle plugged into the reference's Python -- self-consistent, not independent.  Their binaries cannot be had here, but
their published arithmetic can be written a second time on PyTorch's own samplers, which share no code with oracle/stm_oracle.c:

  * deformable convolution (DCNv2 / mmcv DeformConv2d): every tap is a bilinear sample of the ZERO-EXTENDED image -- exactly
    F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=True) on pixel coordinates -- times the mask, then an ordinary matrix product;
  * RoIAlign (mmcv 1.x, aligned=True, sampling_ratio=0): a bin is the mean of ceil(roi_h / ph) x ceil(roi_w / pw) bilinear samples; a sample
    outside [-1, H] x [-1, W] counts as 0, the others are taken on the CLAMPED coordinate = grid_sample(padding_mode="border");
  * spatial_correlation_sample(kernel_size=1, patch_size=P): out[b, i, j, y, x] = sum_c f1[b, c, y, x] * f2[b, c, y + i - P//2, x + j - P//2], zero outside
    -- written with F.pad and slices.

Tolerance 2e-5 relative to the largest value: grid_sample interpolates in fp32 with its own operation order, the oracle accumulates in double.
"""
import math

import pytest
import torch
import torch.nn.functional as F

import oracle


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _sample_zero_extended(img, ys, xs):
    """img [C, H, W], ys / xs [...] pixel coordinates -> [C, ...] bilinear samples of the image extended by zeros."""
    C, H, W = img.shape
    gx = 2.0 * xs / (W - 1) - 1.0
    gy = 2.0 * ys / (H - 1) - 1.0
    grid = torch.stack([gx, gy], -1).view(1, -1, 1, 2).double()
    out = F.grid_sample(img[None].double(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(C, *ys.shape)


def deform_conv_by_grid_sample(x, offset, mask, weight, bias, stride, padding, dilation):
    B, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    (sh, sw), (ph, pw), (dh, dw) = stride, padding, dilation
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    oy = torch.arange(Ho, dtype=torch.float64).view(Ho, 1) * sh - ph
    ox = torch.arange(Wo, dtype=torch.float64).view(1, Wo) * sw - pw
    y = torch.zeros(B, O, Ho, Wo, dtype=torch.float64)
    for b in range(B):
        cols = []
        for i in range(kh):
            for j in range(kw):
                k = i * kw + j
                ys = oy + i * dh + offset[b, 2 * k].double()          # channel 2k = dy, 2k + 1 = dx (Featurealign.py:46-69, dcn_v2)
                xs = ox + j * dw + offset[b, 2 * k + 1].double()
                v = _sample_zero_extended(x[b], ys, xs)               # [C, Ho, Wo]
                if mask is not None:
                    v = v * mask[b, k].double()
                cols.append(v)
        col = torch.stack(cols, 1)                                    # [C, K, Ho, Wo]
        y[b] = torch.einsum("ock,ckhw->ohw", weight.double().view(O, C, kh * kw), col)
        if bias is not None:
            y[b] += bias.double().view(O, 1, 1)
    return y


@pytest.mark.parametrize("case", [
    # B, C, H, W, O, kh, kw, stride, padding, with mask, offset scale
    (2, 8, 9, 11, 6, 3, 3, (1, 1), (1, 1), True, 2.0),       # dcn_v2.DCN as Bottleneck uses it
    (1, 8, 12, 10, 5, 3, 3, (2, 2), (1, 1), True, 2.0),      # the stride-2 blocks (backbone.py:21-22)
    (1, 4, 7, 9, 4, 3, 5, (1, 1), (1, 2), False, 1.5),       # FeatureAlign's DeformConv2d 3x5, no mask (Featurealign.py:27-31)
    (1, 4, 9, 7, 4, 5, 3, (1, 1), (2, 1), False, 4.0),       # 5x3, offsets that leave the image by several pixels
])
def test_deform_conv_oracle_equals_grid_sample_formulation(case):
    B, C, H, W, O, kh, kw, stride, padding, with_mask, oscale = case
    g = _gen(kh * 100 + kw + H)
    x = torch.randn(B, C, H, W, generator=g)
    Ho = (H + 2 * padding[0] - kh) // stride[0] + 1
    Wo = (W + 2 * padding[1] - kw) // stride[1] + 1
    offset = torch.randn(B, 2 * kh * kw, Ho, Wo, generator=g) * oscale
    mask = torch.rand(B, kh * kw, Ho, Wo, generator=g) if with_mask else None
    weight = torch.randn(O, C, kh, kw, generator=g) * 0.2
    bias = torch.randn(O, generator=g) if with_mask else None
    got = oracle.deform_conv(x, offset, mask, weight, bias, stride, padding, 1, 1).double()
    ref = deform_conv_by_grid_sample(x, offset, mask, weight, bias, stride, padding, (1, 1))
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def roi_align_by_grid_sample(feat, rois, ph, pw):
    B, C, H, W = feat.shape
    outs = []
    for r in rois:
        b = int(r[0])
        x1, y1, x2, y2 = [float(v) - 0.5 for v in r[1:]]              # aligned=True, spatial_scale 1
        rw, rh = x2 - x1, y2 - y1
        gh, gw = max(int(math.ceil(rh / ph)), 1), max(int(math.ceil(rw / pw)), 1)
        bh, bw = rh / ph, rw / pw
        iy = (torch.arange(ph, dtype=torch.float64).view(ph, 1) * bh + y1).view(ph, 1, 1, 1) + \
            ((torch.arange(gh, dtype=torch.float64) + 0.5) * bh / gh).view(1, 1, gh, 1)
        ix = (torch.arange(pw, dtype=torch.float64).view(pw, 1) * bw + x1).view(1, pw, 1, 1) + \
            ((torch.arange(gw, dtype=torch.float64) + 0.5) * bw / gw).view(1, 1, 1, gw)
        ys, xs = iy.expand(ph, pw, gh, gw), ix.expand(ph, pw, gh, gw)
        inside = (ys >= -1.0) & (ys <= H) & (xs >= -1.0) & (xs <= W)
        gx = 2.0 * xs.clamp(0, W - 1) / (W - 1) - 1.0
        gy = 2.0 * ys.clamp(0, H - 1) / (H - 1) - 1.0
        grid = torch.stack([gx, gy], -1).view(1, -1, 1, 2)
        v = F.grid_sample(feat[b:b + 1].double(), grid, mode="bilinear", padding_mode="border", align_corners=True).view(C, ph, pw, gh, gw)
        v = v * inside.view(1, ph, pw, gh, gw)
        outs.append(v.sum((3, 4)) / (gh * gw))
    return torch.stack(outs)


def test_roi_align_oracle_equals_grid_sample_formulation():
    g = _gen(7)
    feat = torch.randn(2, 5, 24, 40, generator=g)
    rois = torch.tensor([[0, 3.2, 4.1, 17.9, 15.3], [1, -2.0, -1.5, 9.0, 6.0],      # one that starts outside the map
                         [0, 30.0, 10.0, 44.0, 27.0],                                 # one that ends outside
                         [1, 5.0, 5.0, 5.8, 6.1], [0, 0.0, 0.0, 40.0, 24.0]])        # a tiny one, the whole map
    got = oracle.roi_align(feat, rois, 7).double()
    ref = roi_align_by_grid_sample(feat, rois, 7, 7)
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


def test_correlation_oracle_equals_shifted_products():
    g = _gen(11)
    B, C, H, W, P = 2, 12, 9, 13, 11
    f1, f2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    got = oracle.corr_patch(f1, f2, P).double()                        # [B, P, P, H, W]
    R = P // 2
    f2p = F.pad(f2.double(), (R, R, R, R))
    ref = torch.empty(B, P, P, H, W, dtype=torch.float64)
    for i in range(P):
        for j in range(P):
            ref[:, i, j] = (f1.double() * f2p[:, :, i:i + H, j:j + W]).sum(1)
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # the reference's follow-up (track_to_segment_head.py:60-62): / C, leaky_relu(0.1), [B, P*P, H, W]
    full = oracle.correlate(f1, f2, P).double()
    assert (full - F.leaky_relu(ref.view(B, P * P, H, W) / C, 0.1)).abs().max().item() <= 2e-5
