"""GPU parity of the planar split-operand convolution (stm_conv2d_planar_f32, include/stmask_hip.h) against the fp64 CPU
oracle (oracle.conv2d_nhwc) and torch's fp32 convolution on the same seeded inputs.

Tolerances, stated: with three bf16 planes per operand (six products) or two fp16 planes (three products) every fp32
product is reproduced to ~2^-22..2^-24 and the sum is accumulated in fp32, so the result must sit within a few fp32 ULPs
of the reduction's magnitude: |y - y_fp64| <= 2e-6 * sum_k |x_k w_k|  (observed ~2e-7..4e-7).  With ONE fp16 plane
(fp16x1, BASELINE config 5) each operand carries 11 bits: |y - y_fp64| <= 1e-3 * sum_k |x_k w_k| (2 * 2^-11 per product).
"""
import pytest
import torch
import torch.nn.functional as F

import oracle
from stmask_amd import ops
from ctypes import c_int

from stmask_amd._lib import StmError

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CONV_CASES = [
    # B, H,  W,  C, Cout, kh, kw, stride, pad(h,w), bias, residual, relu
    (2, 12, 20, 32, 64, 3, 3, 1, (1, 1), True, False, True),      # head-tower-like 3x3
    (1, 24, 40, 64, 256, 1, 1, 1, (0, 0), True, True, True),      # bottleneck conv3 + residual
    (2, 13, 17, 32, 27, 3, 3, 2, (1, 1), True, False, False),     # conv_offset_mask-like: stride 2, Cout 27, odd sizes
    (1, 6, 10, 64, 41, 3, 5, 1, (1, 2), True, False, False),      # FCB trailing conv 3x5 -> 41 classes
    (1, 3, 5, 32, 130, 5, 3, 1, (2, 1), False, False, True),      # P7: 15 pixels, 5x3, Cout just over one tile
    (3, 9, 11, 96, 4, 3, 3, 1, (1, 1), True, False, False),       # bbox layer: Cout 4
    (1, 48, 80, 256, 256, 3, 3, 1, (1, 1), True, False, True),    # the P3 tower layer at full size
]


def conv_nhwc(x, w, b=None, r=None, stride=1, padding=0, relu=False, fmt=0, tile_n=128):
    """fp32 NHWC in / out through the planar kernel: split -> stm_conv2d_planar_f32 (fp32 output)."""
    if not (x.is_cuda and w.is_cuda):
        raise StmError("conv_nhwc: tensors must be on the MI355X")
    B, H, W, _ = x.shape
    pk = ops.conv_pack_weights(w, tile_n=tile_n, fmt=fmt)
    scale = 1.0
    if fmt >= 1:
        pk, scale = pk
    y = ops.conv2d_planar(ops.split_planes(x, fmt), pk, tuple(w.shape), (B, H, W), b, r, stride=stride, padding=padding, relu=relu, out="f32",
                          tile_n=tile_n, fmt=fmt, out_scale=scale)
    Ho, Wo = ops.conv_out_hw(H, W, w.shape[2], w.shape[3], *([stride] * 2 if isinstance(stride, int) else stride),
                             *([padding] * 2 if isinstance(padding, int) else padding), 1, 1)
    return y.view(B, Ho, Wo, w.shape[0])


def test_conv_matches_torch_fp32_conv_and_is_no_worse():
    """Against torch's own fp32 convolution (MIOpen): both sit at fp32 rounding distance from the fp64 oracle."""
    B, H, W, C, O = 2, 24, 40, 128, 128
    x, w, b = rnd(B, H, W, C, seed=5), rnd(O, C, 3, 3, seed=6, scale=0.03), rnd(O, seed=7)
    ref = oracle.conv2d_nhwc(x, w, b, None, stride=1, padding=1)
    yt = F.conv2d(x.permute(0, 3, 1, 2).to(DEV), w.to(DEV), b.to(DEV), padding=1).permute(0, 2, 3, 1).cpu()
    e_torch = (yt - ref).abs().max().item()
    for fmt in (0, 1):
        y = conv_nhwc(x.to(DEV), w.to(DEV), b.to(DEV), padding=1, fmt=fmt).cpu()
        e_ours = (y - ref).abs().max().item()
        assert e_ours < 5e-6 and e_ours < 4 * e_torch + 1e-6, (fmt, e_ours, e_torch)


def test_conv_known_answers_and_linearity():
    # identity 1x1 kernel returns the input; a one-hot 3x3 tap shifts the image with zero padding
    C = 32
    x = rnd(1, 7, 9, C, seed=1)
    w = torch.eye(C).reshape(C, C, 1, 1)
    y = conv_nhwc(x.to(DEV), w.to(DEV)).cpu()
    assert torch.equal(y, x)                                   # x = p0 + p1 + p2 exactly, times 1.0
    w3 = torch.zeros(C, C, 3, 3)
    w3[:, :, 0, 2] = torch.eye(C)                              # tap (ky=0, kx=2): y[oy, ox] = x[oy - 1, ox + 1]
    y = conv_nhwc(x.to(DEV), w3.to(DEV), padding=1).cpu()
    exp = torch.zeros_like(x)
    exp[:, 1:, :-1] = x[:, :-1, 1:]
    assert torch.equal(y, exp)
    # linearity at a BASELINE-size layer: conv(a x1 + x2) == a conv(x1) + conv(x2) within fp32 rounding
    x1, x2 = rnd(8, 48, 80, 256, seed=2).to(DEV), rnd(8, 48, 80, 256, seed=3).to(DEV)
    w = rnd(256, 256, 3, 3, seed=4, scale=0.02).to(DEV)
    f = lambda t: conv_nhwc(t, w, padding=1, fmt=1)
    lhs, rhs = f(0.5 * x1 + x2), 0.5 * f(x1) + f(x2)
    assert (lhs - rhs).abs().max().item() < 2e-5


def test_conv_rejects_bad_arguments():
    w = rnd(8, 20, 3, 3).to(DEV)
    with pytest.raises(StmError):
        ops.conv_pack_weights(w)                               # Cin not a multiple of 32
    w = rnd(8, 32, 3, 3).to(DEV)
    pk = ops.conv_pack_weights(w)
    with pytest.raises(StmError):
        ops.conv2d_planar(ops.split_planes(rnd(1, 5, 5, 64).to(DEV)), pk, (8, 32, 3, 3), (1, 5, 5))   # channel mismatch
    with pytest.raises(StmError):
        ops.split_planes(rnd(1, 5, 5, 32))                     # CPU tensor: no fallback
    with pytest.raises(StmError):
        ops.conv2d_planar(ops.split_planes(rnd(1, 5, 5, 32).to(DEV), fmt=1), pk, (8, 32, 3, 3), (1, 5, 5))   # fp16 planes into a bf16 layer


def planes_to_f32(pl):
    return ops.planes_to_f32(pl)


def test_split_planes_is_exact():
    x = rnd(3, 5, 8, 64, seed=11) * torch.logspace(-6, 6, 64)
    pl = ops.split_planes(x.to(DEV)).cpu()
    assert pl.shape == (3, 2, 3 * 5 * 8, 32) and pl.dtype == torch.bfloat16       # [plane, channel slab, pixel, 32]
    assert torch.equal(planes_to_f32(pl), x.view(-1, 64))   # 8+8+8 significand bits: the fp32 value is recovered exactly


@pytest.mark.parametrize("mg", ["1", "2", "n64"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_planar_vs_oracle(case, mg, tunables):
    """Planar (pre-split, LDS-DMA staged) kernel: 128- and 256-pixel tiles of 128 channels and the 128 x 64 tile, fp32
    and planar outputs, both residual forms."""
    tile_n = 64 if mg == "n64" else 128
    if mg != "n64":
        tunables.set(STM_CONV_MG=mg)
    B, H, W, C, O, kh, kw, s, pad, has_bias, has_res, relu = case
    x = rnd(B, H, W, C, seed=0)
    w = rnd(O, C, kh, kw, seed=1, scale=(C * kh * kw) ** -0.5)
    b = rnd(O, seed=2) if has_bias else None
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, s, s, pad[0], pad[1], 1, 1)
    r = rnd(B, Ho, Wo, O, seed=3) if has_res else None
    ref = oracle.conv2d_nhwc(x, w, b, r, stride=s, padding=pad, relu=relu)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs() if has_bias else None, r.abs() if has_res else None, stride=s, padding=pad)
    pkt = ops.conv_pack_weights(w.to(DEV), tile_n=tile_n)
    xp = ops.split_planes(x.to(DEV))
    bd = b.to(DEV) if has_bias else None
    y32, ypl = ops.conv2d_planar(xp, pkt, tuple(w.shape), (B, H, W), bd, r.to(DEV) if has_res else None, stride=s, padding=pad,
                                 relu=relu, out="both", tile_n=tile_n)
    y32, ypl = y32.cpu().view(ref.shape), ypl.cpu()
    assert ((y32 - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
    # the planar output IS the fp32 output, split (channels past Cout in the last slab are never written)
    assert torch.equal(planes_to_f32(ypl)[:, :O], y32.view(-1, O))
    if has_res:                                        # residual handed over as planes gives the same result
        y2 = ops.conv2d_planar(xp, pkt, tuple(w.shape), (B, H, W), bd, ops.split_planes(r.to(DEV)), stride=s, padding=pad,
                               relu=relu, out="f32", tile_n=tile_n).cpu()
        assert torch.equal(y2.view(ref.shape), y32)


def test_conv_planar_chain_of_layers():
    """Two stacked layers handing planes to each other == the fp64 oracle applied twice (error stays at fp32 level)."""
    x = rnd(2, 24, 40, 64, seed=21)
    w1, w2 = rnd(64, 64, 3, 3, seed=22, scale=0.05), rnd(96, 64, 1, 1, seed=23, scale=0.1)
    b1 = rnd(64, seed=24)
    h_ref = oracle.conv2d_nhwc(x, w1, b1, None, padding=1, relu=True)
    y_ref = oracle.conv2d_nhwc(h_ref, w2, None, None)
    h = ops.conv2d_planar(ops.split_planes(x.to(DEV)), ops.conv_pack_weights(w1.to(DEV)), (64, 64, 3, 3), (2, 24, 40), b1.to(DEV),
                          padding=1, relu=True)
    y = ops.conv2d_planar(h, ops.conv_pack_weights(w2.to(DEV)), (96, 64, 1, 1), (2, 24, 40), out="f32").cpu()
    assert (y.view(y_ref.shape) - y_ref).abs().max().item() < 1e-5


def test_conv_planar_levels_groups_and_slices():
    """Launch-descriptor features of stm_conv2d_planar_f32 against the oracle: pixel axis = concatenated levels (one launch
    over several image sizes), grouped convolution, reading a channel range of a wider buffer, writing fp32 output at a
    column offset of a wider matrix, and writing planes into a pixel slice of a larger buffer."""
    from stmask_amd.planar import PlanarConv
    B, sizes, C = 2, [(12, 20), (6, 10), (3, 5)], 64
    G, Og = 2, 64
    xs = [rnd(B, h, w, G * C + 32, seed=10 + i) for i, (h, w) in enumerate(sizes)]       # 32 leading channels are skipped
    wts = rnd(G * Og, C, 3, 5, seed=20, scale=0.03)
    bias = rnd(G * Og, seed=21)
    starts = [0]
    for h, w in sizes:
        starts.append(starts[-1] + B * h * w)
    ntot = starts[-1]
    flat = torch.cat([x.reshape(-1, x.shape[-1]) for x in xs], 0)                     # [ntot, 160] fp32, levels concatenated
    xp = ops.split_planes(flat.to(DEV))                                                # [3, 5, ntot, 32]
    conv = PlanarConv(wts.to(DEV), bias.to(DEV), 1, (1, 2), relu=True, groups=G, tile_n=64)
    buf = torch.zeros(ntot, 3 * G * Og, device=DEV)
    y32, ypl = conv(xp, ("levels", B, sizes), out="both", out_f32=buf, x_ch_off=32, out_ch_off=G * Og)
    assert y32 is buf and torch.count_nonzero(buf[:, :G * Og]) == 0 and torch.count_nonzero(buf[:, 2 * G * Og:]) == 0
    got = buf[:, G * Og:2 * G * Og].cpu()
    for l, (h, w) in enumerate(sizes):
        for g in range(G):
            xg = xs[l][..., 32 + g * C:32 + (g + 1) * C].contiguous()
            ref = oracle.conv2d_nhwc(xg, wts[g * Og:(g + 1) * Og], bias[g * Og:(g + 1) * Og], None, padding=(1, 2), relu=True)
            mag = oracle.conv2d_nhwc(xg.abs(), wts[g * Og:(g + 1) * Og].abs(), bias[g * Og:(g + 1) * Og].abs(), None, padding=(1, 2))
            out = got[starts[l]:starts[l + 1], g * Og:(g + 1) * Og].view(B, h, w, Og)
            assert ((out - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6, (l, g)
    assert torch.equal(planes_to_f32(ypl.cpu()), got)
    # one level as an image batch read from / written into pixel slices of larger plane buffers
    l = 1
    h, w = sizes[l]
    w1 = rnd(32, 32, 3, 3, seed=30, scale=0.05)
    c1 = PlanarConv(w1.to(DEV), None, 1, 1, relu=False)
    dst = torch.zeros(3, 1, ntot, 32, device=DEV, dtype=torch.bfloat16)
    c1(xp, ("img", B, h, w), out="planes", x_off=starts[l], out_planes=dst, out_off=starts[l], x_ch_off=0)
    ref = oracle.conv2d_nhwc(xs[l][..., :32].contiguous(), w1, None, None, padding=1)
    full = planes_to_f32(dst.cpu())
    assert (full[starts[l]:starts[l + 1]].view(B, h, w, 32) - ref).abs().max().item() < 1e-5
    assert torch.count_nonzero(full[:starts[l]]) == 0 and torch.count_nonzero(full[starts[l + 1]:]) == 0


@pytest.mark.parametrize("splitk", ["2", "5"])
@pytest.mark.parametrize("tile_n", [64, 128])
def test_conv_planar_splitk(splitk, tile_n, tunables):
    """Split-K (partial sums through a workspace + finishing kernel, taken for grids that would idle most CUs) against the
    oracle and the unsplit launch, with residual, ReLU, both output forms, Cout not a multiple of 8 (scalar finish)."""
    from stmask_amd.planar import PlanarConv
    for (B, H, W, C, O, k, has_res) in [(2, 6, 10, 256, 128, 3, True), (1, 5, 7, 512, 41, 1, False)]:
        x = rnd(B, H, W, C, seed=3)
        w = rnd(O, C, k, k, seed=4, scale=(C * k * k) ** -0.5)
        b = rnd(O, seed=5)
        r = rnd(B * H * W, O, seed=6) if has_res else None
        ref = oracle.conv2d_nhwc(x, w, b, r.view(B, H, W, O) if has_res else None, padding=k // 2, relu=True)
        mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs(), r.abs().view(B, H, W, O) if has_res else None, padding=k // 2)
        conv = PlanarConv(w.to(DEV), b.to(DEV), 1, k // 2, relu=True, tile_n=tile_n)
        xp = ops.split_planes(x.to(DEV))
        tunables.set(STM_CONV_SPLITK="1")
        y1 = conv(xp, ("img", B, H, W), out="f32", residual=r.to(DEV) if has_res else None).cpu()
        tunables.set(STM_CONV_SPLITK=splitk)
        y32, ypl = conv(xp, ("img", B, H, W), out="both", residual=r.to(DEV) if has_res else None)
        tunables.clear("STM_CONV_SPLITK")
        y32, ypl = y32.cpu(), ypl.cpu()
        assert ((y32.view(ref.shape) - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
        assert ((y32 - y1).abs() / mag.view(-1, O).clamp_min(1e-6)).max().item() < 1e-6      # only the summation order differs
        assert torch.equal(planes_to_f32(ypl)[:, :O], y32)


@pytest.mark.parametrize("tile_n", [64, 128])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_planar_fp16_two_plane_format(case, tile_n):
    """fmt 1: two fp16 planes, three MFMA products (h0 g0 + h0 g1 + h1 g0), weights scaled by a power of two.  Same stated
    tolerance as the bf16 three-plane form: |y - y_fp64| <= 2e-6 * sum|x w| (observed ~4e-7); planar output = fp32 output
    to 2^-21 relative (22 significand bits)."""
    B, H, W, C, O, kh, kw, s, pad, has_bias, has_res, relu = case
    x = rnd(B, H, W, C, seed=0)
    w = rnd(O, C, kh, kw, seed=1, scale=(C * kh * kw) ** -0.5)
    b = rnd(O, seed=2) if has_bias else None
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, s, s, pad[0], pad[1], 1, 1)
    r = rnd(B, Ho, Wo, O, seed=3) if has_res else None
    ref = oracle.conv2d_nhwc(x, w, b, r, stride=s, padding=pad, relu=relu)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs() if has_bias else None, r.abs() if has_res else None, stride=s, padding=pad)
    pk, osc = ops.conv_pack_weights(w.to(DEV), tile_n=tile_n, fmt=1)
    xp = ops.split_planes(x.to(DEV), fmt=1)
    assert xp.dtype == torch.float16 and xp.shape[0] == 2
    assert (planes_to_f32(xp.cpu()) - x.view(-1, C)).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()
    y32, ypl = ops.conv2d_planar(xp, pk, tuple(w.shape), (B, H, W), b.to(DEV) if has_bias else None, r.to(DEV) if has_res else None,
                                 stride=s, padding=pad, relu=relu, out="both", tile_n=tile_n, fmt=1, out_scale=osc)
    y32, ypl = y32.cpu().view(ref.shape), ypl.cpu()
    assert ((y32 - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
    assert (planes_to_f32(ypl)[:, :O] - y32.view(-1, O)).abs().max().item() <= 2.0 ** -21 * max(1.0, y32.abs().max().item())
    if has_res:
        y2 = ops.conv2d_planar(xp, pk, tuple(w.shape), (B, H, W), b.to(DEV) if has_bias else None, ops.split_planes(r.to(DEV), fmt=1),
                               stride=s, padding=pad, relu=relu, out="f32", tile_n=tile_n, fmt=1, out_scale=osc).cpu()
        assert (y2.view(ref.shape) - y32).abs().max().item() < 2e-6 * max(1.0, y32.abs().max().item())


def test_fp16_planes_keep_22_bits_across_magnitudes():
    """x = h + l / 2048 with the low plane stored scaled: every element keeps 22 significand bits from 6.1e-5 to 65504 and
    an absolute error below 1.5e-11 under that (an unscaled low plane would lose bits from |x| < 0.12 down)."""
    g = torch.Generator().manual_seed(11)
    mant = 1.0 + torch.rand(4096, 32, generator=g)
    expo = torch.randint(-30, 16, (4096, 32), generator=g).float()
    sign = torch.where(torch.rand(4096, 32, generator=g) < 0.5, -1.0, 1.0)
    x = (sign * mant * 2.0 ** expo).clamp(-65504.0, 65504.0)
    xp = ops.split_planes(x.to(DEV), fmt=1)
    back = planes_to_f32(xp.cpu())
    err = (back.double() - x.double()).abs()
    assert (err <= torch.maximum(2.0 ** -22 * x.double().abs(), torch.tensor(1.5e-11, dtype=torch.float64))).all()
    torch.cuda.synchronize()
    assert int(ops.planar_range_flag().item()) == 0


@pytest.mark.parametrize("scale", [1e-3, 1e-6, 3e3])
def test_conv_planar_fp16_format_small_and_large_activations(scale):
    """The stated bound (2e-6 of sum|x w|) holds for activations far from 1: 1e-3 (low-plane scaling) and 3e3 (outputs
    near the top of the fp16 range); at 1e-6, under fp16's normal range, the documented absolute floor takes over."""
    B, H, W, C, O = 2, 9, 12, 64, 96
    x = rnd(B, H, W, C, seed=20) * scale
    w = rnd(O, C, 3, 3, seed=21, scale=(C * 9) ** -0.5)
    ref = oracle.conv2d_nhwc(x, w, None, None, padding=1)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), None, None, padding=1)
    pk, osc = ops.conv_pack_weights(w.to(DEV), fmt=1)
    y32, ypl = ops.conv2d_planar(ops.split_planes(x.to(DEV), fmt=1), pk, tuple(w.shape), (B, H, W), padding=1, out="both", fmt=1,
                                 out_scale=osc)
    y32 = y32.cpu().view(ref.shape)
    # elements under fp16's normal range (6.1e-5) carry an absolute error of up to 1.5e-11 instead of a relative one
    wsum = oracle.conv2d_nhwc(torch.ones_like(x), w.abs(), None, None, padding=1)
    assert ((y32 - ref).abs() <= 2e-6 * mag + 1.5e-11 * wsum).all()
    if scale >= 1e-3:
        assert ((y32 - ref).abs() / mag.clamp_min(1e-30)).max().item() < 2e-6
    yb = planes_to_f32(ypl.cpu())[:, :O]
    assert ((yb - y32.view(-1, O)).abs() <= 2.0 ** -21 * y32.view(-1, O).abs() + 1.5e-11).all()
    assert int(ops.planar_range_flag().item()) == 0


def test_fp16_range_flag_is_raised_by_every_plane_producer():
    """|x| > 65504 has no fp16 plane representation: stm_split_planes_fmt_f32, the planar conv epilogue (vector and scalar
    forms) and the planar deformable sampler raise the registered sticky flag; in-range runs leave it alone; a fp32-only
    output does not raise it."""
    flag = ops.planar_range_flag()
    flag.zero_()

    def raised():
        torch.cuda.synchronize()
        v = int(flag.item())
        flag.zero_()
        return v

    x = rnd(1, 4, 4, 32, seed=1)
    ops.split_planes(x.to(DEV), fmt=1)
    assert raised() == 0
    for bad in (7e4, -1e9, float("inf"), float("nan")):
        xb = x.clone()
        xb[0, 2, 1, 5] = bad
        ops.split_planes(xb.to(DEV), fmt=1)
        assert raised() == 1, bad
    ops.split_planes(xb.to(DEV), fmt=0)                  # the bf16 format has fp32's range
    assert raised() == 0
    # conv epilogue: inputs in range, outputs beyond it
    for O in (64, 41):                                   # vector / scalar epilogue
        w = torch.full((O, 32, 1, 1), 1.0)
        xin = torch.full((1, 4, 4, 32), 3000.0)          # y = 96000
        pk, osc = ops.conv_pack_weights(w.to(DEV), fmt=1)
        xp = ops.split_planes(xin.to(DEV), fmt=1)
        y = ops.conv2d_planar(xp, pk, tuple(w.shape), (1, 4, 4), out="f32", fmt=1, out_scale=osc)
        assert raised() == 0 and torch.equal(y.cpu(), torch.full((16, O), 96000.0))
        ops.conv2d_planar(xp, pk, tuple(w.shape), (1, 4, 4), out="both", fmt=1, out_scale=osc)
        assert raised() == 1
    # sampler
    xs = torch.full((1, 6, 6, 128), 1e5)
    om = torch.zeros(36, 27)
    om[:, 18:] = 10.0                                    # mask ~ 1
    ops.dcn_sample_planar(xs.to(DEV), om.to(DEV), 1, 1, 1, fmt=1)
    assert raised() == 1
    ops.dcn_sample_planar((xs * 1e-3).to(DEV), om.to(DEV), 1, 1, 1, fmt=1)
    assert raised() == 0


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("tile_n", [64, 128])
def test_conv_planar_fp16_loop_variants_agree_bitwise(tile_n, fmt, tunables):
    """The K-loop variants of the fp16-format kernels (two planes and one plane) -- three-buffer ring with fragment prefetch
    (default on the 128-wide tiles and the short loops of the 64-wide ones), the two-buffer loop -- add the same products in
    the same order: same bits.  Split-K (partial sums + finishing kernel) differs by the summation order only."""
    from stmask_amd.planar import PlanarConv
    tol = 2e-6 if fmt == 1 else 1e-3
    for (B, H, W, C, O, k) in [(2, 24, 40, 64, 128, 3), (8, 48, 80, 128, 256, 3), (1, 12, 20, 512, 64, 1), (2, 6, 10, 1024, 128, 3)]:
        x = rnd(B, H, W, C, seed=C + k)
        w = rnd(O, C, k, k, seed=O, scale=(C * k * k) ** -0.5)
        b = rnd(O, seed=7)
        conv = PlanarConv(w.to(DEV), b.to(DEV), 1, k // 2, relu=True, tile_n=tile_n, fmt=fmt)
        xp = ops.split_planes(x.to(DEV), fmt=fmt)
        outs = {}
        for name, env in [("ring", {}), ("two-buffer", {"STM_CONV_RING": "2", "STM_CONV_RING64": "2"}), ("ring64", {"STM_CONV_RING64": "4"}),
                          ("splitk", {"STM_CONV_SPLITK": "3"})]:
            tunables.set(**env)
            y32, ypl = conv(xp, ("img", B, H, W), out="both")
            outs[name] = (y32.cpu(), ypl.cpu())
            tunables.clear(*env)
        ref = oracle.conv2d_nhwc(x, w, b, None, padding=k // 2, relu=True)
        mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs(), None, padding=k // 2)
        assert ((outs["ring"][0].view(ref.shape) - ref).abs() / mag.clamp_min(1e-6)).max().item() < tol
        for name in ("two-buffer", "ring64"):
            assert torch.equal(outs[name][0], outs["ring"][0]) and torch.equal(outs[name][1], outs["ring"][1]), (name, C, k)
        assert ((outs["splitk"][0] - outs["ring"][0]).abs() / mag.view(-1, O).clamp_min(1e-6)).max().item() < 1e-6


@pytest.mark.parametrize("tile_n", [64, 128])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_planar_fp16_one_plane_format(case, tile_n):
    """fmt 2 (BASELINE config 5: genuine fp16 convolutions): one fp16 plane per operand, one MFMA product, fp32 accumulation,
    fp32 bias / residual / ReLU.  Stated tolerance: 1e-3 * sum |x w| against the fp64 oracle (each operand is rounded to 11
    bits: at most 2 * 2^-11 per product).  The planes output is RN16 of the fp32 output; with out_fmt 1 the layer writes both
    planes of the fp16x2 format (what the last backbone layer of a stage hands to the fp32-equivalent FPN)."""
    from stmask_amd.planar import PlanarConv
    B, H, W, C, O, kh, kw, s_, pad, has_bias, has_res, relu = case
    x = rnd(B, H, W, C, seed=0)
    w = rnd(O, C, kh, kw, seed=1, scale=(C * kh * kw) ** -0.5)
    b = rnd(O, seed=2) if has_bias else None
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, s_, s_, pad[0], pad[1], 1, 1)
    r = rnd(B, Ho, Wo, O, seed=3) if has_res else None
    ref = oracle.conv2d_nhwc(x, w, b, r, stride=s_, padding=pad, relu=relu)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs() if has_bias else None, r.abs() if has_res else None, stride=s_, padding=pad)
    xp = ops.split_planes(x.to(DEV), fmt=2)
    assert xp.shape[0] == 1 and xp.dtype == torch.float16
    assert torch.equal(ops.planes_to_f32(xp.cpu()), x.view(-1, C).half().float())
    conv = PlanarConv(w.to(DEV), b.to(DEV) if has_bias else None, s_, pad, relu=relu, tile_n=tile_n, fmt=2)
    rp = ops.split_planes(r.to(DEV), fmt=2) if has_res else None
    y32, ypl = conv(xp, ("img", B, H, W), out="both", residual=r.view(-1, O).to(DEV) if has_res else None)
    y32, ypl = y32.cpu(), ypl.cpu()
    err = ((y32.view(ref.shape) - ref).abs() / mag.clamp_min(1e-6)).max().item()
    assert err < 1e-3, err
    assert err > 1e-6                                              # it really is the one-plane arithmetic
    assert ypl.shape[0] == 1 and torch.equal(ops.planes_to_f32(ypl)[:, :O], y32.half().float())
    # both planes on request (out_fmt 1): the fp32 output to 22 bits
    conv2 = PlanarConv(w.to(DEV), b.to(DEV) if has_bias else None, s_, pad, relu=relu, tile_n=tile_n, fmt=2, out_fmt=1)
    z32, zpl = conv2(xp, ("img", B, H, W), out="both", residual=r.view(-1, O).to(DEV) if has_res else None)
    assert torch.equal(z32.cpu(), y32) and zpl.shape[0] == 2
    assert (ops.planes_to_f32(zpl.cpu())[:, :O] - y32).abs().max().item() <= 4e-7 * max(1.0, y32.abs().max().item())
    # plane 0 of the two-plane tensor IS the one-plane tensor (channels past Cout in the last slab are never written)
    assert torch.equal(ops.planes_to_f32(zpl[0:1].cpu())[:, :O], ops.planes_to_f32(ypl)[:, :O])
    if has_res:                                                    # residual as a (one-plane) planar tensor: rounded to fp16 first
        y3 = conv(xp, ("img", B, H, W), out="f32", residual=rp).cpu()
        assert (y3 - y32).abs().max().item() < 2e-3 * max(1.0, r.abs().max().item())
    # a two-plane fp16x2 tensor is accepted as input: plane 0 is read
    y4 = conv(ops.split_planes(x.to(DEV), fmt=1), ("img", B, H, W), out="f32").cpu() if not has_res else None
    if y4 is not None:
        assert torch.equal(y4, y32)


@pytest.mark.parametrize("fmt", [1, 0, 2])
@pytest.mark.parametrize("case", [(2, 12, 20, 64, 24, 40), (1, 9, 7, 32, 18, 14), (2, 5, 6, 96, 13, 11), (1, 8, 8, 32, 8, 8)])
def test_resize_bilinear_planes_equals_interpolate_then_split(case, fmt):
    """stm_resize_bilinear_planes_f32 == F.interpolate(mode="bilinear", align_corners=False) followed by
    stm_split_planes_fmt_f32: the same expression in the same order, so at most an ulp of fp32 apart before the split."""
    import torch.nn.functional as F
    B, H, W, C, Ho, Wo = case
    x = rnd(B, H, W, C, seed=H * W)
    got = planes_to_f32(ops.resize_bilinear_planes(x.to(DEV), (Ho, Wo), fmt=fmt).cpu())
    ref = F.interpolate(x.permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).reshape(-1, C)
    # (non-integer scales: the source coordinate itself may differ by an ulp from ATen's, hence 4e-7 and not 2^-23)
    tol = {0: 4e-7, 1: 2.0 ** -21, 2: 2.0 ** -10}[fmt]
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()) + 1e-7
    on_gpu = F.interpolate(x.to(DEV).permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).contiguous()
    assert (planes_to_f32(ops.split_planes(on_gpu, fmt=fmt).cpu()) - got).abs().max().item() <= (4e-7 if fmt != 2 else 2.0 ** -10) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("fmt", [1, 0, 2])
@pytest.mark.parametrize("case", [(2, 12, 20, 64), (1, 9, 7, 32), (2, 5, 6, 96), (1, 1, 1, 32)])
def test_bias_relu_maxpool_planes_equals_torch_chain(case, fmt):
    """stm_bias_relu_maxpool_planes_f32 == split(max_pool2d(relu(x + bias), 3, 2, 1)): bias add and ReLU are monotone per
    channel, so pooling first changes nothing -- equal bit for bit (odd sizes, 1x1 input, borders included)."""
    import torch.nn.functional as F
    B, H, W, C = case
    x = rnd(B, H, W, C, seed=H + W)
    bias = rnd(C, seed=3)
    pl, (Ho, Wo) = ops.bias_relu_maxpool_planes(x.to(DEV), bias.to(DEV), fmt=fmt)
    ref = F.max_pool2d(torch.relu(x + bias).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
    assert (Ho, Wo) == tuple(ref.shape[1:3])
    assert torch.equal(pl.cpu(), ops.split_planes(ref.to(DEV), fmt=fmt).cpu())
    pl0, _ = ops.bias_relu_maxpool_planes(x.to(DEV), None, fmt=fmt)
    ref0 = F.max_pool2d(torch.relu(x).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(pl0.cpu(), ops.split_planes(ref0.to(DEV), fmt=fmt).cpu())


@pytest.mark.parametrize("fmt", [1, 0, 2])
def test_roi_align_planes_equals_cat_relu_roialign_split(fmt):
    """stm_roi_align_planes_f32 == relu(cat(corr, T2S_prev, T2S)) -> stm_roi_align_avg_f32 -> channel reorder + zero pad ->
    stm_split_planes_fmt_f32, bit for bit (same arithmetic, operation for operation); boxes of all sizes, at the borders,
    degenerate, and on every image of the batch; the RoIAlign itself is pinned to the oracle in test_gpu_kernels.py."""
    B, H, W, C1, Cc = 3, 12, 20, 32, 9
    g = torch.Generator().manual_seed(5)
    prev, cur = rnd(B, H, W, C1, seed=1), rnd(B, H, W, C1, seed=2)
    corr = rnd(B, Cc, H, W, seed=3)
    n = 41
    x1 = torch.rand(n, generator=g) * W * 1.1 - 1.0
    y1 = torch.rand(n, generator=g) * H * 1.1 - 1.0
    bw = torch.rand(n, generator=g) ** 2 * W
    bh = torch.rand(n, generator=g) ** 2 * H
    rois = torch.stack([torch.randint(0, B, (n,), generator=g).float(), x1, y1, x1 + bw, y1 + bh], 1)
    rois[0, 1:] = torch.tensor([3.0, 4.0, 3.0, 4.0])            # empty box
    rois[1, 1:] = torch.tensor([0.0, 0.0, float(W), float(H)])  # whole map
    pl = ops.roi_align_planes(prev.to(DEV), cur.to(DEV), corr.to(DEV), rois.to(DEV), 7, fmt=fmt)
    feats = torch.relu(torch.cat([corr, prev.permute(0, 3, 1, 2), cur.permute(0, 3, 1, 2)], 1)).contiguous()
    ref = ops.roi_align(feats.to(DEV), rois.to(DEV), 7)                                   # [n, Cc + 2 C1, 7, 7]
    ref = torch.cat([ref[:, Cc:], ref[:, :Cc]], 1).permute(0, 2, 3, 1)                    # [T2S_prev | T2S | corr], NHWC
    cpad = pl.shape[1] * 32
    ref = torch.nn.functional.pad(ref, (0, cpad - ref.shape[-1])).contiguous()
    assert torch.equal(pl.cpu(), ops.split_planes(ref, fmt=fmt).cpu().view_as(pl.cpu()))
    # the correlation volume channels-last in a padded row (16 floats, the spare ones poisoned): same planes
    cl = torch.full((B, H, W, 16), float("nan"))
    cl[..., :Cc] = corr.permute(0, 2, 3, 1)
    pl2 = ops.roi_align_planes(prev.to(DEV), cur.to(DEV), cl.to(DEV), rois.to(DEV), 7, fmt=fmt, corr_nhwc=Cc)
    assert torch.equal(pl2.cpu(), pl.cpu())


def test_roi_align_planes_forms_agree_bitwise(tunables):
    """The three kernels behind stm_roi_align_planes_nhwc_f32 -- one pixel's channel groups per wave (STM_ROI_TILED=0), 16 pixels x all slabs per
    workgroup (1), the RoI's feature patch staged in LDS (2, the default when the channel counts are whole slabs) -- write the same planes: boxes
    inside, across the borders, outside the map, degenerate, a tenth of the map, and the WHOLE map (960 pixels: beyond the staged patch, the
    workgroup's direct-load path); correlation channels 41 of a 64-float row (the slab holds real channels, masked channels and padding)."""
    B, H, W, C1, Cc, ld = 3, 24, 40, 64, 41, 64
    g = torch.Generator().manual_seed(7)
    prev, cur = rnd(B, H, W, C1, seed=1).to(DEV), rnd(B, H, W, C1, seed=2).to(DEV)
    cl = torch.full((B, H, W, ld), float("nan"))
    cl[..., :Cc] = rnd(B, H, W, Cc, seed=3)
    n = 64
    x1 = torch.rand(n, generator=g) * W * 1.2 - 3.0
    y1 = torch.rand(n, generator=g) * H * 1.2 - 3.0
    bw = torch.rand(n, generator=g) ** 2 * W * 0.6
    bh = torch.rand(n, generator=g) ** 2 * H * 0.6
    rois = torch.stack([torch.randint(0, B, (n,), generator=g).float(), x1, y1, x1 + bw, y1 + bh], 1)
    rois[0, 1:] = torch.tensor([3.0, 4.0, 3.0, 4.0])                      # empty box
    rois[1, 1:] = torch.tensor([0.0, 0.0, float(W), float(H)])            # whole map
    rois[2, 1:] = torch.tensor([-9.0, -7.0, -2.5, -1.5])                  # outside
    rois[3, 1:] = torch.tensor([W - 0.5, H - 0.5, W + 6.0, H + 5.0])      # hanging over the far corner
    rois[4, 1:] = torch.tensor([5.0, 2.0, 31.0, 20.0])                    # 27 x 19 pixels: beyond the staged patch too
    outs = []
    for form in ("0", "1", "2"):
        tunables.set(STM_ROI_TILED=form)
        outs.append(ops.roi_align_planes(prev, cur, cl.to(DEV), rois.to(DEV), 7, fmt=1, corr_nhwc=Cc).clone())
        tunables.clear("STM_ROI_TILED")
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(ops.planes_to_f32(outs[2])).all()


@pytest.mark.parametrize("fmt", [1, 0])
def test_stem_row_patches_plus_planar_conv_equals_7x7_stride2(fmt):
    """stm_stem_rows_planes_f32 + a (7 x 1), stride (2, 1) planar convolution over the row-patch tensor == the stem's 7x7 /
    stride-2 / padding-3 convolution on the 3-channel frame (oracle, fp64 accumulation) to the planar kernel's stated bound;
    odd and even frame sizes, borders included."""
    from stmask_amd import planar
    from stmask_amd.planar import PlanarConv
    planar.set_format(fmt)
    try:
        for (B, H, W) in [(2, 24, 36), (1, 17, 23)]:
            x = rnd(B, H, W, 3, seed=H)
            w = rnd(64, 3, 7, 7, seed=5, scale=147 ** -0.5)
            ref = oracle.conv2d_nhwc(x, w, None, None, stride=2, padding=3)
            mag = oracle.conv2d_nhwc(x.abs(), w.abs(), None, None, stride=2, padding=3)
            rp, Wo = ops.stem_rows_planes(x.to(DEV), 7, 2, 3, fmt)
            assert Wo == ref.shape[2]
            # the row-patch tensor holds the 21 contiguous values of each kernel row
            R = planes_to_f32(rp.cpu()).view(B, H, Wo, 32)
            xp = torch.nn.functional.pad(x, (0, 0, 3, 3))
            for ox in (0, 1, Wo - 1):
                want = xp[:, :, 2 * ox:2 * ox + 7, :].reshape(B, H, 21)
                assert (R[:, :, ox, :21] - want).abs().max().item() <= 2.0 ** -21 * 4 and torch.count_nonzero(R[:, :, ox, 21:]) == 0
            wk = torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(64, 7, 21), (0, 11)).permute(0, 2, 1).reshape(64, 32, 7, 1).contiguous()
            conv = PlanarConv(wk.to(DEV), None, (2, 1), (3, 0), relu=False)
            y = conv(rp, ("img", B, H, Wo), out="f32").cpu().view(ref.shape)
            assert ((y - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
    finally:
        planar.set_format(1)


def test_f16_entry_points_are_plane_format_2():
    """The `_f16`-named C entries of BASELINE config 5 (stm_split_planes_f16 / stm_conv_pack_weights_f16 / stm_conv2d_planar_f16 /
    stm_dcn_sample_planar_f16) called through ctypes give the bits of the format-aware entries with fmt = 2."""
    import ctypes
    from stmask_amd import _lib
    from stmask_amd._lib import c_f, c_i, c_l, c_p, c_sz
    lib = _lib.lib()
    stream = c_p(torch.cuda.current_stream().cuda_stream)
    B, H, W, C, O = 2, 12, 20, 64, 96
    x, w, b = rnd(B, H, W, C, seed=1).to(DEV), (rnd(O, C, 3, 3, seed=2) * 0.04).to(DEV), rnd(O, seed=3).to(DEV)
    xp = torch.empty(1, C // 32, B * H * W, 32, device=DEV, dtype=torch.float16)
    _lib.check(lib.stm_split_planes_f16(c_p(x.data_ptr()), c_p(xp.data_ptr()), c_l(B * H * W), c_i(C), stream), "stm_split_planes_f16")
    assert torch.equal(xp, ops.split_planes(x, fmt=2))
    pk_ref, oscale = ops.conv_pack_weights(w, tile_n=128, fmt=2)
    pk = torch.empty_like(pk_ref)
    _lib.check(lib.stm_conv_pack_weights_f16(c_p(w.data_ptr()), c_p(pk.data_ptr()), c_i(O), c_i(C), c_i(3), c_i(3), c_i(128), c_f(1.0 / oscale), stream),
               "stm_conv_pack_weights_f16")
    assert torch.equal(pk, pk_ref)
    y_ref = ops.conv2d_planar(xp, pk, (O, C, 3, 3), (B, H, W), b, None, padding=1, relu=True, out="f32", fmt=2, out_scale=oscale)
    g = _lib.ConvGeom(B, H, W, C, H, W, O, 3, 3, 1, 1, 1, 1, 0, 0, 0, 3)      # planes / fmt left at other values: the entry forces them
    g.out_scale, g.tile_n = oscale, 128
    y = torch.empty(B * H * W, O, device=DEV)
    _lib.check(lib.stm_conv2d_planar_f16(c_p(xp.data_ptr()), c_p(pk.data_ptr()), c_p(b.data_ptr()), c_p(0), c_p(0), c_p(y.data_ptr()), c_p(0),
                                         ctypes.byref(g), c_i(1), c_p(0), c_sz(0), stream), "stm_conv2d_planar_f16")
    assert torch.equal(y, y_ref)
    xs, om = rnd(1, 8, 10, 128, seed=4).to(DEV), (rnd(80, 27, seed=5) * 0.8).to(DEV)
    cols_ref = ops.dcn_sample_planar(xs, om, 1, 1, 1, fmt=2)
    cols = torch.empty_like(cols_ref)
    dg = _lib.DeformGeom(1, 128, 8, 10, 3, 3, 1, 1, 1, 1, 1, 1, 1, 8, 10)
    _lib.check(lib.stm_dcn_sample_planar_f16(c_p(xs.data_ptr()), c_p(om.data_ptr()), c_i(27), c_p(cols.data_ptr()), c_i(80), c_l(0), ctypes.byref(dg),
                                             stream), "stm_dcn_sample_planar_f16")
    assert cols.shape[0] == 1 and torch.equal(cols, cols_ref)


# ---------------------------------------------------------------------------------------------- narrow layers: the kx-reuse kernel
def _planar_conv_vs_oracle(conv, xs, sizes, B, wts, bias, pad, relu, C, groups, cg, real, fmt, tol):
    """Run `conv` over the concatenated levels and compare every (level, group) with the fp64 oracle."""
    starts = [0]
    for h, w in sizes:
        starts.append(starts[-1] + B * h * w)
    flat = torch.cat([x.reshape(-1, x.shape[-1]) for x in xs], 0)
    xp = ops.split_planes(flat.to(DEV), fmt)
    shape = ("levels", B, sizes) if len(sizes) > 1 else ("img", B, *sizes[0])
    y32, ypl = conv(xp, shape, out="both")
    y32, ypl = y32.cpu(), ypl.cpu()
    worst = 0.0
    for l, (h, w) in enumerate(sizes):
        for g in range(groups):
            xg = xs[l][..., g * C:(g + 1) * C].contiguous()
            wg, bg = wts[g * cg:g * cg + real[g]], (bias[g * cg:g * cg + real[g]] if bias is not None else None)
            ref = oracle.conv2d_nhwc(xg, wg, bg, None, padding=pad, relu=relu)
            mag = oracle.conv2d_nhwc(xg.abs(), wg.abs(), bg.abs() if bg is not None else None, None, padding=pad)
            out = y32[starts[l]:starts[l + 1], g * cg:g * cg + real[g]].view(B, h, w, real[g])
            worst = max(worst, ((out - ref).abs() / mag.clamp_min(1e-6)).max().item())
    assert worst < tol, worst
    return y32, ypl, worst


KXR_CASES = [
    # kh, kw, groups, channels per group (row stride), real channels per group, sizes, B, relu
    (3, 3, 3, 64, (41, 5, 32), [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)], 3, False),     # head output layers, five levels
    (3, 5, 3, 64, (41, 5, 32), [(12, 20), (6, 10), (3, 5)], 2, False),
    (5, 3, 3, 64, (41, 5, 32), [(12, 20), (6, 10), (3, 5)], 2, False),
    (3, 3, 1, 64, (64,), [(24, 40)], 2, True),                                           # layer1's 64 -> 64 3x3 (+ ReLU)
    (3, 3, 1, 32, (27,), [(13, 17)], 3, False),                                          # DCN offset / mask convolution, odd sizes
    (5, 3, 2, 64, (64, 16), [(9, 7), (4, 5)], 1, True),
]


@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("case", KXR_CASES)
def test_conv_kxr_vs_oracle(case, fmt):
    """stm_conv2d_planar_kxr_f32 (csrc/conv_kxr.hip: kx-reuse staging, channel tiles of 16, transposed product, stores from the
    accumulators) against the fp64 oracle at the bound of the general planar kernel -- 2e-6 of sum |x w| with two fp16 planes, 1e-3
    with one -- over multi-level pixel axes whose tiles straddle image rows, images and (for the last tile) the level's end, grouped
    layers with different real channel counts, and both output forms."""
    from stmask_amd.planar import PlanarConv
    kh, kw, G, cg, real, sizes, B, relu = case
    C = 64
    xs = [rnd(B, h, w, G * C, seed=40 + i) for i, (h, w) in enumerate(sizes)]
    wts = rnd(G * cg, C, kh, kw, seed=50, scale=(C * kh * kw) ** -0.5)
    for g in range(G):
        wts[g * cg + real[g]:(g + 1) * cg] = 0.0                     # zero-padded groups, as planar.py builds them
    bias = rnd(G * cg, seed=51)
    pad = ((kh - 1) // 2, (kw - 1) // 2)
    conv = PlanarConv(wts.to(DEV), bias.to(DEV), 1, pad, relu=relu, groups=G, group_cout=list(real), tile_n=64, fmt=fmt)
    conv.kxr = ops.conv_kxr_supported(G * cg, C, kh, kw, 1, pad, G, list(real), fmt)     # (the graph leaves four-tile groups to the general kernel)
    conv.kxr_min_pixels = 0
    assert conv.kxr
    tol = 2e-6 if fmt == 1 else 1e-3
    y32, ypl, worst = _planar_conv_vs_oracle(conv, xs, sizes, B, wts, bias, pad, relu, C, G, cg, real, fmt, tol)
    # the planes ARE the fp32 output (channels of whole 16-tiles; the rest of a slab is never written)
    back = planes_to_f32(ypl)
    for g in range(G):
        sl = slice(g * cg, g * cg + real[g])
        assert (back[:, sl] - y32[:, sl]).abs().max().item() <= 2.0 ** (-21 if fmt == 1 else -10) * max(1.0, y32[:, sl].abs().max().item())
    # ... and the general kernel gives the same values to fp32 rounding (different MFMA operand roles, same products in the same order)
    ref_conv = PlanarConv(wts.to(DEV), bias.to(DEV), 1, pad, relu=relu, groups=G, group_cout=list(real), tile_n=64, fmt=fmt)
    ref_conv.kxr = False
    flat = torch.cat([x.reshape(-1, x.shape[-1]) for x in xs], 0)
    shape = ("levels", B, sizes) if len(sizes) > 1 else ("img", B, *sizes[0])
    y_old = ref_conv(ops.split_planes(flat.to(DEV), fmt), shape, out="f32").cpu()
    for g in range(G):
        sl = slice(g * cg, g * cg + real[g])
        assert (y_old[:, sl] - y32[:, sl]).abs().max().item() < (2e-5 if fmt == 1 else 2e-2)


def test_conv_kxr_known_answers():
    """One-hot taps shift the image with zero padding -- across image rows, images and levels (the cases the always-zero LDS row
    and the DMA's range check exist for): exact."""
    from stmask_amd.planar import PlanarConv
    C, B, sizes = 32, 2, [(7, 9), (5, 33), (3, 4)]
    xs = [rnd(B, h, w, C, seed=60 + i) for i, (h, w) in enumerate(sizes)]
    flat = torch.cat([x.reshape(-1, C) for x in xs], 0)
    starts = [0]
    for h, w in sizes:
        starts.append(starts[-1] + B * h * w)
    for (ky, kx) in [(0, 0), (0, 4), (2, 2), (1, 3), (2, 0)]:
        w5 = torch.zeros(C, C, 3, 5)
        w5[:, :, ky, kx] = torch.eye(C)                      # y[oy, ox] = x[oy + ky - 1, ox + kx - 2]
        conv = PlanarConv(w5.to(DEV), None, 1, (1, 2), relu=False, fmt=1)
        conv.kxr_min_pixels = 0
        assert conv.kxr
        y = conv(ops.split_planes(flat.to(DEV), 1), ("levels", B, sizes), out="f32").cpu()
        for l, (h, w) in enumerate(sizes):
            exp = torch.zeros(B, h, w, C)
            dy, dx = ky - 1, kx - 2
            ys, xs_ = slice(max(0, -dy), min(h, h - dy)), slice(max(0, -dx), min(w, w - dx))
            yd, xd = slice(max(0, dy), min(h, h + dy)), slice(max(0, dx), min(w, w + dx))
            exp[:, ys, xs_] = xs[l][:, yd, xd]
            got = y[starts[l]:starts[l + 1]].view(B, h, w, C)
            assert (got - exp).abs().max().item() <= 2.0 ** -21 * xs[l].abs().max().item(), (ky, kx, l)


# ---------------------------------------------------------------------------------------------- the stem as one kernel
@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("hw", [(64, 96), (37, 53), (384, 640)])
def test_stem_fused_vs_fp64(hw, fmt):
    """stm_stem_fused_f32 (conv1 7x7 / 2 / 3 + folded-BN bias + ReLU + MaxPool2d(3, 2, 1) in one kernel, csrc/stem_fused.hip)
    against the same chain in float64: 2e-6 of sum |x w| with two fp16 planes (the bound of the planar convolutions), 1e-3 with one;
    odd sizes exercise partial tiles, pool windows and conv windows that leave the frame on every side."""
    H, W = hw
    B = 2 if H < 300 else 1
    x = rnd(B, H, W, 3, seed=70)
    w = rnd(64, 3, 7, 7, seed=71, scale=147 ** -0.5)
    b = rnd(64, seed=72)
    xd, wd = x.permute(0, 3, 1, 2).double(), w.double()
    ref = F.max_pool2d(F.relu(F.conv2d(xd, wd, b.double(), stride=2, padding=3)), 3, 2, 1).permute(0, 2, 3, 1)
    mag = F.max_pool2d(F.conv2d(xd.abs(), wd.abs(), b.abs().double(), stride=2, padding=3), 3, 2, 1).permute(0, 2, 3, 1)
    packed, osc = ops.stem_pack_weights(w.to(DEV), fmt)
    planes, (Hp, Wp) = ops.stem_fused(x.to(DEV), packed, osc, b.to(DEV), fmt)
    assert (Hp, Wp) == tuple(ref.shape[1:3]) and planes.shape == ((2 if fmt == 1 else 1), 2, B * Hp * Wp, 32)
    y = planes_to_f32(planes.cpu()).view(B, Hp, Wp, 64).double()
    tol = 2e-6 if fmt == 1 else 1e-3
    # (the pooled maximum may come from a neighbouring position whose value is within the bound: compare values, bound by the
    # window's largest magnitude)
    assert ((y - ref).abs() / mag.clamp_min(1e-6)).max().item() < tol + (2.0 ** -21 if fmt == 1 else 2.0 ** -10)
    torch.cuda.synchronize()
    assert int(ops.planar_range_flag().item()) == 0


def test_stem_fused_equals_three_kernel_stem():
    """The one-kernel stem against round 2's three launches (row patches -> (7 x 1) planar convolution -> bias + ReLU + max-pool):
    same products, same fp32 accumulation per kernel row; the K order inside a slab differs (pixel-major 4-channel groups instead of
    21 interleaved values), so the results agree to fp32 rounding of the sums, not bit for bit."""
    import stmask_amd.planar as pl
    from stmask_amd.backbone import ResNetBackbone
    torch.manual_seed(0)
    x = rnd(2, 96, 160, 3, seed=80).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last).to(DEV)
    w, b = rnd(64, 3, 7, 7, seed=81, scale=147 ** -0.5).to(DEV), rnd(64, seed=82).to(DEV)
    packed, osc = ops.stem_pack_weights(w, 1)
    fused, (Hp, Wp) = ops.stem_fused(x.permute(0, 2, 3, 1), packed, osc, b, 1)
    wr = torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(64, 7, 21), (0, 11)).permute(0, 2, 1).reshape(64, 32, 7, 1).contiguous()
    conv = pl.PlanarConv(wr, None, (2, 1), (3, 0), relu=False, fmt=1)
    rp, Wo = ops.stem_rows_planes(x.permute(0, 2, 3, 1), 7, 2, 3, 1)
    y = conv(rp, ("img", 2, 96, Wo), out="f32").view(2, 48, Wo, 64)
    old, _ = ops.bias_relu_maxpool_planes(y, b, 1)
    a, c = planes_to_f32(fused.cpu()), planes_to_f32(old.cpu())
    assert a.shape == c.shape and (a - c).abs().max().item() < 2e-6 * max(1.0, c.abs().max().item())


# ---------------------------------------------------------------------------------------------- conv3 + projection shortcut as one product
@pytest.mark.parametrize("fmt", [1, 2])
@pytest.mark.parametrize("case", [(2, 24, 40, 64, 64, 256, 1, 64), (2, 25, 39, 128, 256, 512, 2, 64), (1, 12, 20, 256, 512, 1024, 2, 128),
                                  (3, 6, 10, 512, 1024, 2048, 2, 128)])
def test_conv_dual_source_vs_oracle(case, fmt):
    """stm_conv2d_planar_dual_f32: relu(W3 mid + b3 + Wds x[::s, ::s] + bds) as one two-source 1x1 product (the first bottleneck of a ResNet
    stage, backbone.py:38-58) against the fp64 oracle of the two convolutions -- the stated bound of the planar kernel, 2e-6 of sum |x w|
    (1e-3 with one fp16 plane) -- on both tile widths, odd sizes, split-K grids, and against the two-launch form it replaces."""
    from stmask_amd.planar import PlanarConv
    B, H2, W2, P, Cin, O, s, tile_n = case
    Ho, Wo = (H2 - 1) // s + 1, (W2 - 1) // s + 1
    mid, x = rnd(B, Ho, Wo, P, seed=90), rnd(B, H2, W2, Cin, seed=91)
    w3, wd = rnd(O, P, 1, 1, seed=92, scale=P ** -0.5), rnd(O, Cin, 1, 1, seed=93, scale=Cin ** -0.5)
    b3, bd = rnd(O, seed=94), rnd(O, seed=95)
    xs = x[:, ::s, ::s].contiguous()
    ref = torch.relu(oracle.conv2d_nhwc(mid, w3, b3, None) + oracle.conv2d_nhwc(xs, wd, bd, None))
    mag = oracle.conv2d_nhwc(mid.abs(), w3.abs(), b3.abs(), None) + oracle.conv2d_nhwc(xs.abs(), wd.abs(), bd.abs(), None)
    conv = PlanarConv(torch.cat([w3, wd], 1).to(DEV), (b3 + bd).to(DEV), 1, 0, relu=True, fmt=fmt, tile_n=tile_n)
    midp, xp = ops.split_planes(mid.to(DEV), fmt), ops.split_planes(x.to(DEV), fmt)
    y32, ypl = conv(midp, ("img", B, Ho, Wo), out="both", x2=(xp, H2, W2, s))
    y32 = y32.cpu().view(ref.shape)
    tol = 2e-6 if fmt == 1 else 1e-3
    assert ((y32 - ref).abs() / mag.clamp_min(1e-6)).max().item() < tol
    assert (planes_to_f32(ypl.cpu()) - y32.view(-1, O)).abs().max().item() <= 2.0 ** (-21 if fmt == 1 else -10) * max(1.0, y32.abs().max().item())
    # the two-launch form: projection -> planes, then conv3 with that residual (one more rounding of the projection to the plane format)
    ds = PlanarConv(wd.to(DEV), bd.to(DEV), s, 0, relu=False, fmt=fmt, tile_n=tile_n)
    c3 = PlanarConv(w3.to(DEV), b3.to(DEV), 1, 0, relu=True, fmt=fmt, tile_n=tile_n)
    y2 = c3(midp, ("img", B, Ho, Wo), out="f32", residual=ds(xp, ("img", B, H2, W2))).cpu().view(ref.shape)
    assert (y2 - y32).abs().max().item() < (1e-5 if fmt == 1 else 2e-2) * max(1.0, ref.abs().max().item())


# ---------------------------------------------------------------------------------------------- layer1's bottleneck chain as one kernel
def _chain_layers(seed=0):
    """conv2 3x3 64 -> 64, conv3 1x1 64 -> 256, next conv1 1x1 256 -> 64 with BatchNorm-folded-like biases."""
    w2 = rnd(64, 64, 3, 3, seed=seed + 1, scale=(64 * 9) ** -0.5)
    w3 = rnd(256, 64, 1, 1, seed=seed + 2, scale=64 ** -0.5)
    w1 = rnd(64, 256, 1, 1, seed=seed + 3, scale=256 ** -0.5)
    b2, b3, b1 = rnd(64, seed=seed + 4, scale=0.3), rnd(256, seed=seed + 5, scale=0.3), rnd(64, seed=seed + 6, scale=0.3)
    return w2, w3, w1, b2, b3, b1


def _chain_pack(w2, w3, w1):
    from stmask_amd import _lib
    g = _lib.ConvGeom()
    g.C, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.groups, g.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
    ops.planar_range_flag()
    w2p, s2 = ops.conv_pack_weights_kxr(w2.to(DEV), g)
    tail, s3, s1 = ops.chain_pack_tail(w3.to(DEV), w1.to(DEV) if w1 is not None else None)
    return w2p, tail, (s2, s3, s1)


@pytest.mark.parametrize("case", [(2, 24, 40), (3, 13, 17), (1, 5, 7), (1, 96, 160)])
@pytest.mark.parametrize("with_next", [True, False])
def test_bottleneck_chain_vs_fp64_and_vs_three_launches(case, with_next):
    """stm_bottleneck_chain_f32 (csrc/conv_chain.hip): relu(conv3(relu(conv2(mid1))) + x) and the next block's relu(conv1(.)) from one
    launch against (a) the same three layers in fp64 -- bound 4e-6 of the magnitude sums, twice a single layer's (two chained products
    of plane-split intermediates) -- and (b) the three stm_conv2d_planar_f32 launches it replaces: same products, intermediates split
    into the same fp16 planes, only the order of the sums inside a K-slab differs.  Tiles straddle rows and images, the last is ragged."""
    from stmask_amd.planar import PlanarConv
    B, H, W = case
    w2, w3, w1, b2, b3, b1 = _chain_layers()
    mid1 = rnd(B, H, W, 64, seed=11).abs()                       # post-ReLU input, like the real one
    x = rnd(B, H, W, 256, seed=12).abs()
    w2p, tail, scales = _chain_pack(w2, w3, w1 if with_next else None)
    m1p, xp = ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x.to(DEV), 1)
    y, z = ops.bottleneck_chain(m1p, xp, w2p, tail, b2.to(DEV), b3.to(DEV), b1.to(DEV) if with_next else None, scales, B, H, W, want_z=with_next)
    assert (z is None) == (not with_next)
    # (a) fp64 of the layers, with the magnitude sums as the scale of the bound
    d = torch.float64
    c2 = F.conv2d(mid1.permute(0, 3, 1, 2).to(d), w2.to(d), b2.to(d), padding=1).relu()
    c3 = (F.conv2d(c2, w3.to(d), b3.to(d)) + x.permute(0, 3, 1, 2).to(d)).relu()
    mag3 = F.conv2d(F.conv2d(mid1.permute(0, 3, 1, 2).to(d), w2.abs().to(d), b2.abs().to(d), padding=1), w3.abs().to(d), b3.abs().to(d)) + x.permute(0, 3, 1, 2).to(d)
    got_y = planes_to_f32(y).cpu().view(B, H, W, 256).permute(0, 3, 1, 2).to(d)
    assert ((got_y - c3).abs() / mag3.clamp_min(1e-6)).max().item() < 4e-6
    if with_next:
        c1 = F.conv2d(c3, w1.to(d), b1.to(d)).relu()
        mag1 = F.conv2d(mag3, w1.abs().to(d), b1.abs().to(d))
        got_z = planes_to_f32(z).cpu().view(B, H, W, 64).permute(0, 3, 1, 2).to(d)
        assert ((got_z - c1).abs() / mag1.clamp_min(1e-6)).max().item() < 6e-6
    # (b) the three launches
    l2 = PlanarConv(w2.to(DEV), b2.to(DEV), 1, 1, relu=True, fmt=1)
    l3 = PlanarConv(w3.to(DEV), b3.to(DEV), 1, 0, relu=True, fmt=1)
    m2 = l2(m1p, ("img", B, H, W))
    y3 = l3(m2, ("img", B, H, W), residual=xp)
    ya, yb = planes_to_f32(y), planes_to_f32(y3)
    assert (ya - yb).abs().max().item() < 2e-5 * max(1.0, yb.abs().max().item())
    if with_next:
        l1 = PlanarConv(w1.to(DEV), b1.to(DEV), 1, 0, relu=True, fmt=1)
        za, zb = planes_to_f32(z), planes_to_f32(l1(y3, ("img", B, H, W)))
        assert (za - zb).abs().max().item() < 2e-5 * max(1.0, zb.abs().max().item())


def test_bottleneck_chain_known_answers_and_arguments():
    """Identity weights: conv2 = centre tap identity, conv3 = 4 copies, conv1' = pick-one-copy -> y = relu(mid1 copies + x), z = a copy of
    y's channels; exact to the 22 bits of the planes.  One-hot off-centre taps shift the image with zero padding across rows / images.
    Bad shapes are refused."""
    B, H, W = 2, 6, 9
    mid1, x = rnd(B, H, W, 64, seed=3).abs(), rnd(B, H, W, 256, seed=4)
    for (ky, kx) in [(1, 1), (0, 0), (2, 1), (1, 2)]:
        w2 = torch.zeros(64, 64, 3, 3)
        w2[:, :, ky, kx] = torch.eye(64)
        w3 = torch.zeros(256, 64)
        for k in range(4):
            w3[64 * k:64 * (k + 1)] = torch.eye(64) * (k + 1)
        w1 = torch.zeros(64, 256)
        w1[:, 64:128] = torch.eye(64)                         # z = y[:, 64:128]
        w2p, tail, scales = _chain_pack(w2, w3.view(256, 64, 1, 1), w1.view(64, 256, 1, 1))
        y, z = ops.bottleneck_chain(ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x.to(DEV), 1), w2p, tail, None, None, None, scales, B, H, W)
        sh = torch.zeros(B, H, W, 64)
        dy, dx = ky - 1, kx - 1
        ys, xs_ = slice(max(0, -dy), min(H, H - dy)), slice(max(0, -dx), min(W, W - dx))
        yd, xd = slice(max(0, dy), min(H, H + dy)), slice(max(0, dx), min(W, W + dx))
        sh[:, ys, xs_] = mid1[:, yd, xd]
        exp = (torch.cat([sh * (k + 1) for k in range(4)], -1) + x).relu()
        got = planes_to_f32(y).cpu().view(B, H, W, 256)
        assert (got - exp).abs().max().item() <= 2.0 ** -20 * exp.abs().max().item(), (ky, kx)
        gz = planes_to_f32(z).cpu().view(B, H, W, 64)
        assert (gz - got[..., 64:128]).abs().max().item() <= 2.0 ** -20 * exp.abs().max().item(), (ky, kx)
    with pytest.raises(StmError):
        ops.bottleneck_chain(ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x.to(DEV), 1)[:, :4].contiguous(), w2p, tail, None, None, None, scales, B, H, W)


@pytest.mark.parametrize("case", [(2, 24, 40), (3, 13, 17), (1, 96, 160)])
def test_bottleneck_chain_projection_form(case):
    """stm_bottleneck_chain_proj_f32: a stage's first block -- y = relu(conv3(relu(conv2(mid1))) + proj(x0) + b3 + bds), z = relu(conv1'(y))
    -- against fp64 and against the launches it replaces (3x3, then the two-source product stm_conv2d_planar_dual_f32, then the 1x1)."""
    from stmask_amd.planar import PlanarConv
    B, H, W = case
    w2, w3, w1, b2, b3, b1 = _chain_layers(seed=20)
    wds, bds = rnd(256, 64, 1, 1, seed=31, scale=64 ** -0.5), rnd(256, seed=32, scale=0.3)
    mid1, x0 = rnd(B, H, W, 64, seed=13).abs(), rnd(B, H, W, 64, seed=14).abs()
    from stmask_amd import _lib
    g = _lib.ConvGeom()
    g.C, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.groups, g.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
    ops.planar_range_flag()
    w2p, s2 = ops.conv_pack_weights_kxr(w2.to(DEV), g)
    tail, s3, s1 = ops.chain_pack_tail(w3.to(DEV), w1.to(DEV), wds.to(DEV))
    m1p, x0p = ops.split_planes(mid1.to(DEV), 1), ops.split_planes(x0.to(DEV), 1)
    y, z = ops.bottleneck_chain(m1p, x0p, w2p, tail, b2.to(DEV), (b3 + bds).to(DEV), b1.to(DEV), (s2, s3, s1), B, H, W, proj=True)
    d = torch.float64
    n = lambda t: t.permute(0, 3, 1, 2).to(d)
    c2 = F.conv2d(n(mid1), w2.to(d), b2.to(d), padding=1).relu()
    c3 = (F.conv2d(c2, w3.to(d), b3.to(d)) + F.conv2d(n(x0), wds.to(d), bds.to(d))).relu()
    c1 = F.conv2d(c3, w1.to(d), b1.to(d)).relu()
    mag3 = (F.conv2d(F.conv2d(n(mid1), w2.abs().to(d), b2.abs().to(d), padding=1), w3.abs().to(d), b3.abs().to(d))
            + F.conv2d(n(x0), wds.abs().to(d), bds.abs().to(d)))
    mag1 = F.conv2d(mag3, w1.abs().to(d), b1.abs().to(d))
    got_y = planes_to_f32(y).cpu().view(B, H, W, 256).permute(0, 3, 1, 2).to(d)
    got_z = planes_to_f32(z).cpu().view(B, H, W, 64).permute(0, 3, 1, 2).to(d)
    assert ((got_y - c3).abs() / mag3.clamp_min(1e-6)).max().item() < 4e-6
    assert ((got_z - c1).abs() / mag1.clamp_min(1e-6)).max().item() < 6e-6
    l2 = PlanarConv(w2.to(DEV), b2.to(DEV), 1, 1, relu=True, fmt=1)
    l3 = PlanarConv(torch.cat([w3, wds], 1).to(DEV), (b3 + bds).to(DEV), 1, 0, relu=True, fmt=1)
    l1 = PlanarConv(w1.to(DEV), b1.to(DEV), 1, 0, relu=True, fmt=1)
    y3 = l3(l2(m1p, ("img", B, H, W)), ("img", B, H, W), x2=(x0p, H, W, 1))
    ya, yb = planes_to_f32(y), planes_to_f32(y3)
    assert (ya - yb).abs().max().item() < 2e-5 * max(1.0, yb.abs().max().item())
    za, zb = planes_to_f32(z), planes_to_f32(l1(y3, ("img", B, H, W)))
    assert (za - zb).abs().max().item() < 2e-5 * max(1.0, zb.abs().max().item())


# ---------------------------------------------------------------------------------------------- window launches (TemporalNet's border classes)
@pytest.mark.parametrize("hw", [(7, 7), (5, 9), (3, 3)])
def test_conv_window_launches_equal_the_padded_convolution(hw):
    """stm_conv_geom.win_*: a 3x3 / pad-1 convolution computed as nine window launches -- row classes {0}, {1..H-2}, {H-1} x column
    classes, each with the sub-kernel of its real taps and negative padding -- writes the same tensor as the single launch: the skipped
    taps only ever added exact zeros, so the fp32 sums are the same sums (bit-equal where no launch is split along K; both against fp64)."""
    from stmask_amd.planar import PlanarConv
    H, W = hw
    n, C, O = 37, 64, 128
    x = rnd(n, H, W, C, seed=70)
    w = rnd(O, C, 3, 3, seed=71, scale=(C * 9) ** -0.5)
    b = rnd(O, seed=72)
    xp = ops.split_planes(x.to(DEV), 1)
    ref_conv = PlanarConv(w.to(DEV), b.to(DEV), 1, 1, relu=True, fmt=1, tile_n=64)
    ref32, refpl = ref_conv(xp, ("img", n, H, W), out="both")
    out32 = torch.full((n * H * W, O), float("nan"), device=DEV)
    outpl = torch.zeros_like(refpl)
    rows, cols = ((0, 1), (1, H - 1), (H - 1, H)), ((0, 1), (1, W - 1), (W - 1, W))
    for ry, (ky0, ky1) in enumerate(((1, 3), (0, 3), (0, 2))):
        for rx, (kx0, kx1) in enumerate(((1, 3), (0, 3), (0, 2))):
            (y0, y1), (x0, x1) = rows[ry], cols[rx]
            if y1 <= y0 or x1 <= x0:
                continue
            conv = PlanarConv(w[:, :, ky0:ky1, kx0:kx1].contiguous().to(DEV), b.to(DEV), 1, 0, relu=True, fmt=1, tile_n=64)
            win = (y0, x0, y1 - y0, x1 - x0, 1 - y0 - ky0, 1 - x0 - kx0, H, W)
            conv(xp, ("img", n, H, W), out="both", out_planes=outpl, out_f32=out32, window=win)
    assert torch.isfinite(out32).all()                     # every output row was written by exactly the class that owns it
    assert torch.equal(out32, ref32) and torch.equal(outpl, refpl)
    exp = oracle.conv2d_nhwc(x, w, b, None, padding=1, relu=True)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs(), None, padding=1)
    assert ((out32.cpu().view(n, H, W, O) - exp).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
    with pytest.raises(StmError):                          # a window that leaves the image is refused
        conv(xp, ("img", n, H, W), out="both", out_planes=outpl, out_f32=out32, window=(H - 1, 0, 2, 1, 0, 0, H, W))
    # ... and all windows as ONE grid (stm_conv2d_planar_windows_f32, the form TemporalNet uses): 128-channel tiles, one weight scale
    wsc = ops._pow2_wscale(w)
    wins, packed = [], []
    for ry, (ky0, ky1) in enumerate(((1, 3), (0, 3), (0, 2))):
        for rx, (kx0, kx1) in enumerate(((1, 3), (0, 3), (0, 2))):
            (y0, y1), (x0, x1) = rows[ry], cols[rx]
            if y1 <= y0 or x1 <= x0:
                continue
            wins.append((ky1 - ky0, kx1 - kx0, 1 - y0 - ky0, 1 - x0 - kx0, y1 - y0, x1 - x0, y0, x0))
            packed.append(ops.conv_pack_weights(w[:, :, ky0:ky1, kx0:kx1].contiguous().to(DEV), tile_n=128, fmt=1, wscale=wsc)[0])
    o32 = torch.full((n * H * W, O), float("nan"), device=DEV)
    opl = torch.zeros_like(refpl)
    ops.conv2d_planar_windows(xp, packed, wins, b.to(DEV), n, H, W, C, O, H, W, 1.0 / wsc, relu=True, out_f32=o32, out_planes=opl)
    assert torch.equal(o32, ref32) and torch.equal(opl, refpl)
    with pytest.raises(StmError):
        ops.conv2d_planar_windows(xp, packed, [(3, 3, 0, 0, H, W, 1, 0)] + wins[1:], b.to(DEV), n, H, W, C, O, H, W, 1.0 / wsc, out_f32=o32)


def test_temporalnet_border_classes_equal_the_three_launches():
    """PlanarTemporalNet with the border-class launches against the same object running its three padded convolutions (equal up to the
    split of small launches along K, which changes the order of the fp32 sums)."""
    from stmask_amd import planar
    import types
    tn = types.SimpleNamespace()
    g = torch.Generator().manual_seed(5)
    mk = lambda o, c: torch.nn.Conv2d(c, o, 3, padding=1)
    tn.conv1, tn.conv2, tn.conv3 = mk(128, 96), mk(128, 128), mk(256, 128)
    tn.fc, tn.fc_coeff = torch.nn.Linear(256, 4), torch.nn.Linear(256, 8)
    for m in (tn.conv1, tn.conv2, tn.conv3, tn.fc, tn.fc_coeff):
        m.to(DEV)
    old_fmt = planar.FMT
    planar.FMT = 1
    try:
        net = planar.PlanarTemporalNet(tn, corr_channels=32)
        assert net.border is not None
        feats = torch.randn(340, 96, 7, 7, generator=g).to(DEV)
        a_loc, a_co = net(feats)
        net.border = None
        b_loc, b_co = net(feats)
    finally:
        planar.FMT = old_fmt
    assert (a_loc - b_loc).abs().max().item() < 1e-6 and (a_co - b_co).abs().max().item() < 1e-6


def test_temporalnet_pooled_epilogue_equals_pool_of_the_written_tensor(monkeypatch):
    """conv3 + ReLU + AvgPool2d as one launch (stm_conv2d_planar_windows_pool_f32: pooled sums in 32.32 fixed point, integer atomics) and the
    fc / fc_coeff tail (stm_temporal_pool_fc_f32) against the same layers writing the [n * 49, O] tensor, torch.mean and torch linear layers:
    the means agree to fp32 rounding of a 49-term sum, the tail to 1e-6; two runs of the pooled path are bit-equal (the accumulation is integer:
    arrival order cannot matter); the pooled sums are back to zero afterwards."""
    from stmask_amd import planar
    import types
    tn = types.SimpleNamespace()
    g = torch.Generator().manual_seed(9)
    mk = lambda o, c: torch.nn.Conv2d(c, o, 3, padding=1)
    tn.conv1, tn.conv2, tn.conv3 = mk(128, 96), mk(128, 128), mk(256, 128)
    tn.fc, tn.fc_coeff = torch.nn.Linear(256, 4), torch.nn.Linear(256, 32)
    for m in (tn.conv1, tn.conv2, tn.conv3, tn.fc, tn.fc_coeff):
        m.to(DEV)
    monkeypatch.setattr(planar, "FMT", 1)
    net = planar.PlanarTemporalNet(tn, corr_channels=32)
    assert net.border is not None
    n = 347                                                       # 17 003 pixels: the classes' last tiles are partial, RoIs straddle 64-row blocks
    feats = torch.randn(n, 96, 7, 7, generator=g).to(DEV)
    monkeypatch.setattr(planar, "TN_POOL", True)
    a_loc, a_co = net(feats)
    assert a_loc.shape == (n, 4) and a_co.shape == (n, 32) and a_loc.is_contiguous() and a_co.is_contiguous()
    assert int(net._pool.abs().max()) == 0                        # consumed and cleared
    a2_loc, a2_co = net(feats)
    assert torch.equal(a_loc, a2_loc) and torch.equal(a_co, a2_co)
    monkeypatch.setattr(planar, "TN_POOL", False)
    b_loc, b_co = net(feats)
    assert (a_loc - b_loc).abs().max().item() < 1e-6 and (a_co - b_co).abs().max().item() < 1e-6
    # the pooled means themselves, against the mean of conv3's written output
    x = F.pad(feats.index_select(1, net.perm.to(feats.device)).permute(0, 2, 3, 1), (0, net.cpad - 96)).contiguous()
    xp = ops.split_planes(x, 1)
    x2 = net._border_layer(1, net._border_layer(0, xp, n, 7, 7, "planes"), n, 7, 7, "planes")
    y = net._border_layer(2, x2, n, 7, 7, "f32").view(n, 49, -1)
    pool = net._border_layer(2, x2, n, 7, 7, "pool")
    sums = pool[:n].double() / 4294967296.0
    assert (sums - y.double().sum(1)).abs().max().item() <= 49 * 2.0 ** -32 + 4e-6 * y.abs().sum(1).max().item()
    (out, pooled) = ops.temporal_pool_fc(pool, n, 49, net.w_tail, net.b_tail, want_pooled=True)
    assert (pooled - y.mean(1)).abs().max().item() <= 2e-6 * max(1.0, y.abs().max().item())
    ref = torch.cat([tn.fc(y.mean(1)), tn.fc_coeff(y.mean(1))], 1)
    assert (out - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item())
    assert int(pool.abs().max()) == 0
    with pytest.raises(StmError):
        ops.conv2d_planar_windows_pool(xp, [], [], None, n, 7, 7, 96, 256, 7, 7, 1.0, torch.zeros(n, 256, device=DEV))   # not int64


KX3_CASES = [
    # B, sizes (levels) or one (H, W), C per group, O, groups, kh, residual
    ("img 48x80", 8, [(48, 80)], 128, 256, 1, 3, False),
    ("img 37x53 odd, residual, partial last tile", 19, [(37, 53)], 64, 128, 1, 3, True),
    ("levels, grouped towers", 6, [(24, 40), (12, 20), (6, 10), (3, 5), (2, 3)], 64, 512, 4, 3, False),
    ("levels 5x3 kernel", 5, [(24, 40), (12, 20), (6, 10), (3, 5), (2, 3)], 64, 128, 1, 5, False),
    ("1x3 kernel", 4, [(40, 64)], 96, 128, 1, 1, False),
]


@pytest.mark.parametrize("case", KX3_CASES, ids=[c[0] for c in KX3_CASES])
def test_conv_planar_kx3_staging_is_bit_equal_to_the_ring_kernel(case, tunables):
    """conv_planar_kx3_kernel (STM_CONV_KX3=1: one staged run of BM + 2 pixels per (channel slab, ky) serves the three taps of a kernel row, the
    image-row borders read an always-zero staged row) multiplies the same fragments in the same order as the ring kernel that stages every tap:
    bit-equal fp32 and plane outputs -- single image size with odd widths and a partial last tile, the concatenated-levels pixel axis of the shared
    head (tiles that span two levels), grouped layers, kh = 1 / 3 / 5, residual + ReLU; and against the fp64 oracle on the plain case."""
    from stmask_amd import _lib
    from stmask_amd.planar import PlanarConv
    name, B, sizes, C, O, groups, kh, with_res = case
    x = torch.cat([rnd(B, h, w, C * groups, seed=7 + i).reshape(-1, C * groups) for i, (h, w) in enumerate(sizes)], 0)      # [M, C] concatenated levels
    M = x.shape[0]
    w = rnd(O, C, kh, 3, seed=3, scale=(C * kh * 3) ** -0.5)
    b = rnd(O, seed=4)
    conv = PlanarConv(w.to(DEV), b.to(DEV), 1, (kh // 2, 1), relu=True, groups=groups, tile_n=128, fmt=1)
    xp = ops.split_planes(x.to(DEV), 1)
    res = ops.split_planes(rnd(M, O, seed=5).to(DEV), 1) if with_res else None
    shape = ("levels", B, sizes) if len(sizes) > 1 else ("img", B, sizes[0][0], sizes[0][1])
    outs = []
    for kx3 in ("0", "1"):
        tunables.set(STM_CONV_KX3=kx3, STM_CONV_MG="2", STM_CONV_SPLITK="1")     # (256-pixel tiles and no split along K whatever the pixel count)
        n0 = _lib.lib().stm_debug_launch_count(c_int(0))
        y32, ypl = conv(xp, shape, out="both", residual=res)
        torch.cuda.synchronize()
        assert _lib.lib().stm_debug_launch_count(c_int(0)) - n0 == (1 if kx3 == "1" else 0), "the kx-reuse kernel did not take this layer"
        outs.append((y32.clone(), ypl.clone()))
        tunables.clear("STM_CONV_KX3", "STM_CONV_MG", "STM_CONV_SPLITK")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if name == "img 48x80":
        xi = x.view(B, sizes[0][0], sizes[0][1], C)
        ref = oracle.conv2d_nhwc(xi, w, b, None, padding=(1, 1), relu=True)
        mag = oracle.conv2d_nhwc(xi.abs(), w.abs(), b.abs(), None, padding=(1, 1))
        assert ((outs[1][0].cpu().view(ref.shape) - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6


@pytest.mark.parametrize("kh,ph", [(3, 0), (5, 1), (3, 2)])
def test_conv_planar_kx3_leaves_other_height_paddings_to_the_ring_kernel(kh, ph, tunables):
    """The kx-reuse kernel decodes a pixel index once for the output and the staged input rows: it is only right for "same" padding in height as
    well as in width.  A kw = 3 / pw = 1 layer whose height padding is NOT kh // 2 (Ho != H) must stay on conv_planar_kernel -- the launch counter
    does not move, and the result equals the run with the kernel switched off and the fp64 oracle."""
    from stmask_amd import _lib
    from stmask_amd.planar import PlanarConv
    B, H, W, C, O = 8, 48, 80, 64, 128
    x = rnd(B, H, W, C, seed=11)
    w = rnd(O, C, kh, 3, seed=12, scale=(C * kh * 3) ** -0.5)
    b = rnd(O, seed=13)
    conv = PlanarConv(w.to(DEV), b.to(DEV), 1, (ph, 1), relu=True, tile_n=128, fmt=1)
    xp = ops.split_planes(x.to(DEV), 1)
    outs = []
    for kx3 in ("0", "1"):
        tunables.set(STM_CONV_KX3=kx3, STM_CONV_MG="2", STM_CONV_SPLITK="1")
        n0 = _lib.lib().stm_debug_launch_count(c_int(0))
        y32 = conv(xp, ("img", B, H, W), out="f32")
        torch.cuda.synchronize()
        assert _lib.lib().stm_debug_launch_count(c_int(0)) == n0, "a layer without same padding in height went to the kx-reuse kernel"
        outs.append(y32.clone())
        tunables.clear("STM_CONV_KX3", "STM_CONV_MG", "STM_CONV_SPLITK")
    assert torch.equal(outs[0], outs[1])
    ref = oracle.conv2d_nhwc(x, w, b, None, padding=(ph, 1), relu=True)
    mag = oracle.conv2d_nhwc(x.abs(), w.abs(), b.abs(), None, padding=(ph, 1))
    assert ((outs[1].cpu().view(ref.shape) - ref).abs() / mag.clamp_min(1e-6)).max().item() < 2e-6
