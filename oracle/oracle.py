"""ctypes front-end of ``libstm_oracle.so`` (plain-C restatement, see stm_oracle.c).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Every wrapper takes/returns CPU ``torch`` tensors (fp32 / int64, made contiguous) so tests
read like the reference's own torch code.  The ``Oracle*`` modules at the bottom are the
CPU stand-ins plugged into the reference's import names by ``tests/golden/gen_golden.py``
(dcn_v2.DCN, mmcv.ops.DeformConv2d / roi_align, spatial_correlation_sample).
"""
import ctypes
import math
import os
import subprocess

import torch
import torch.nn as nn

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libstm_oracle.so")
_lib = None

c_f = ctypes.c_float
c_d = ctypes.c_double
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_p = ctypes.c_void_p


def build(force=False):
    """Compile the oracle with gcc (seconds).  Called by __graft_entry__.build() and lazily here."""
    src = os.path.join(_HERE, "stm_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_expf.restype = c_f
        _lib.orc_expf.argtypes = [c_f]
        _lib.orc_candidate_filter.restype = c_l
        _lib.orc_cc_fast_nms.restype = c_l
        _lib.orc_fast_nms.restype = c_l
        _lib.orc_num_threads.restype = c_i
    return _lib


def _f32(t):
    return t.detach().to(torch.float32).contiguous().cpu()


def _ptr(t):
    return c_p(t.data_ptr()) if t is not None else c_p(0)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(c_i(int(n)))


# ----------------------------------------------------------------------------- exact ops
def expf(x):
    x = _f32(x)
    y = torch.empty_like(x)
    lib().orc_expf_arr(_ptr(x), _ptr(y), c_l(x.numel()))
    return y


def decode(loc, priors):
    """box_utils.py:238-283"""
    loc, priors = _f32(loc), _f32(priors)
    out = torch.empty_like(loc)
    lib().orc_decode_boxes(_ptr(loc), _ptr(priors), c_l(loc.shape[0]), _ptr(out))
    return out


def center_size(boxes):
    """box_utils.py:25-35"""
    boxes = _f32(boxes)
    out = torch.empty_like(boxes)
    lib().orc_center_size(_ptr(boxes), c_l(boxes.shape[0]), _ptr(out))
    return out


def make_priors(conv_h, conv_w, aspect_ratios=((3, 3), (3, 5), (5, 3))):
    """prediction_head_FC.py:224-247 -> [1, h*w*len(ars), 4]"""
    ars = (c_d * (2 * len(aspect_ratios)))(*[float(v) for ar in aspect_ratios for v in ar])
    out = torch.empty(conv_h * conv_w * len(aspect_ratios), 4, dtype=torch.float32)
    lib().orc_make_priors(c_i(conv_h), c_i(conv_w), ars, c_i(len(aspect_ratios)), _ptr(out))
    return out.view(1, -1, 4)


def jaccard(a, b):
    """box_utils.py:60-88 (2-D inputs)"""
    a, b = _f32(a), _f32(b)
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32)
    lib().orc_jaccard(_ptr(a), c_l(a.shape[0]), _ptr(b), c_l(b.shape[0]), _ptr(out))
    return out


def candidate_filter(conf, thresh=0.05):
    """TF_utils.py:68-71 -> kept row indices (ascending)"""
    conf = _f32(conf)
    idx = torch.empty(conf.shape[0], dtype=torch.int64)
    k = lib().orc_candidate_filter(_ptr(conf), c_l(conf.shape[0]), c_i(conf.shape[1]), c_f(thresh), _ptr(idx))
    return idx[:k].clone()


def cc_fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200):
    """detection_TF.py:85-134.  conf [K, ncls] incl. background column 0."""
    conf, boxes = _f32(conf), _f32(boxes)
    cen = _f32(centerness) if centerness is not None else None
    K = conf.shape[0]
    idx = torch.empty(top_k, dtype=torch.int64)
    cls = torch.empty(top_k, dtype=torch.int64)
    sc = torch.empty(top_k, dtype=torch.float32)
    n = lib().orc_cc_fast_nms(_ptr(conf), _ptr(boxes), _ptr(cen), c_l(K), c_i(conf.shape[1]), c_f(iou_thr),
                              c_l(top_k), _ptr(idx), _ptr(cls), _ptr(sc))
    return idx[:n].clone(), cls[:n].clone(), sc[:n].clone()


def fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200, conf_thresh=0.05, max_det=100):
    """detection_TF.py:136-204"""
    conf, boxes = _f32(conf), _f32(boxes)
    cen = _f32(centerness) if centerness is not None else None
    K = conf.shape[0]
    idx = torch.empty(max_det, dtype=torch.int64)
    cls = torch.empty(max_det, dtype=torch.int64)
    sc = torch.empty(max_det, dtype=torch.float32)
    n = lib().orc_fast_nms(_ptr(conf), _ptr(boxes), _ptr(cen), c_l(K), c_i(conf.shape[1]), c_f(iou_thr),
                           c_l(top_k), c_f(conf_thresh), c_l(max_det), _ptr(idx), _ptr(cls), _ptr(sc))
    return idx[:n].clone(), cls[:n].clone(), sc[:n].clone()


def sanitize_hw(boxes, h, w):
    """box_utils.py:319-337"""
    boxes = _f32(boxes)
    out = torch.empty_like(boxes)
    lib().orc_sanitize_hw(_ptr(boxes), c_l(boxes.shape[0]), c_i(h), c_i(w), _ptr(out))
    return out


def fcb_ali_offsets(loc, kh, kw):
    """Featurealign.py:46-69.  loc [B,4,H,W] -> [B, 2*kh*kw, H, W]"""
    loc = _f32(loc)
    B, _, H, W = loc.shape
    out = torch.empty(B, 2 * kh * kw, H, W, dtype=torch.float32)
    lib().orc_fcb_ali_offsets(_ptr(loc), c_i(B), c_i(H), c_i(W), c_i(kh), c_i(kw), _ptr(out))
    return out


# ------------------------------------------------------------------------- tolerance ops
def generate_mask(proto, coeff, boxes=None, apply_tanh=True):
    """mask_utils.py:111-128 + box_utils.py:341-364.  proto [h,w,m] -> [n,h,w]"""
    proto, coeff = _f32(proto), _f32(coeff)
    bx = _f32(boxes) if boxes is not None else None
    h, w, m = proto.shape
    n = coeff.shape[0]
    out = torch.empty(n, h, w, dtype=torch.float32)
    lib().orc_lincomb_sigmoid_crop(_ptr(proto), _ptr(coeff), _ptr(bx), c_i(h), c_i(w), c_i(m), c_l(n),
                                   c_i(1 if apply_tanh else 0), _ptr(out))
    return out


def mask_iou(m1, m2, thr=0.5):
    """box_utils.py:435-447 on (m > thr) masks.  m1 [n1,h,w], m2 [n2,h,w] soft masks."""
    m1, m2 = _f32(m1), _f32(m2)
    n1, n2 = m1.shape[0], m2.shape[0]
    hw = m1[0].numel() if n1 else (m2[0].numel() if n2 else 0)
    out = torch.empty(n1, n2, dtype=torch.float32)
    lib().orc_mask_iou(_ptr(m1), c_l(n1), _ptr(m2), c_l(n2), c_l(hw), c_f(thr), _ptr(out))
    return out


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    return Ho, Wo


def deform_im2col(x, offset, mask, kernel_size, stride=1, padding=0, dilation=1, deform_groups=1):
    """-> cols [B, C*kh*kw, Ho*Wo]"""
    x, offset = _f32(x), _f32(offset)
    mk = _f32(mask) if mask is not None else None
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    Ho, Wo = _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert offset.shape == (B, deform_groups * 2 * kh * kw, Ho, Wo), (offset.shape, (Ho, Wo))
    cols = torch.empty(B, C * kh * kw, Ho * Wo, dtype=torch.float32)
    lib().orc_deform_im2col(_ptr(x), _ptr(offset), _ptr(mk), c_i(B), c_i(C), c_i(H), c_i(W), c_i(kh), c_i(kw),
                            c_i(sh), c_i(sw), c_i(ph), c_i(pw), c_i(dh), c_i(dw), c_i(deform_groups), c_i(Ho),
                            c_i(Wo), _ptr(cols))
    return cols


def deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, deform_groups=1):
    """Deformable convolution (mask=None -> v1).  Double accumulation.  -> [B,O,Ho,Wo]"""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    mk = _f32(mask) if mask is not None else None
    bs = _f32(bias) if bias is not None else None
    O, Cw, kh, kw = weight.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    assert Cw == C, "groups != 1 is outside the hot path"
    Ho, Wo = _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    assert offset.shape == (B, deform_groups * 2 * kh * kw, Ho, Wo), (offset.shape, (Ho, Wo))
    y = torch.empty(B, O, Ho, Wo, dtype=torch.float32)
    lib().orc_deform_conv(_ptr(x), _ptr(offset), _ptr(mk), _ptr(weight), _ptr(bs), c_i(B), c_i(C), c_i(H), c_i(W),
                          c_i(O), c_i(kh), c_i(kw), c_i(sh), c_i(sw), c_i(ph), c_i(pw), c_i(dh), c_i(dw),
                          c_i(deform_groups), c_i(Ho), c_i(Wo), _ptr(y))
    return y


def conv2d_nhwc(x, weight, bias=None, residual=None, stride=1, padding=0, relu=False):
    """nn.Conv2d semantics on an NHWC activation [B,H,W,C] with OIHW weights; y = act(conv + bias + residual), NHWC."""
    x, weight = _f32(x), _f32(weight)
    bs = _f32(bias) if bias is not None else None
    rs = _f32(residual) if residual is not None else None
    B, H, W, C = x.shape
    O, Cw, kh, kw = weight.shape
    assert Cw == C
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    Ho, Wo = _out_hw(H, W, kh, kw, sh, sw, ph, pw, 1, 1)
    y = torch.empty(B, Ho, Wo, O, dtype=torch.float32)
    lib().orc_conv2d_nhwc(_ptr(x), _ptr(weight), _ptr(bs), _ptr(rs), c_i(B), c_i(H), c_i(W), c_i(C), c_i(O), c_i(kh),
                          c_i(kw), c_i(sh), c_i(sw), c_i(ph), c_i(pw), c_i(Ho), c_i(Wo), c_i(C), c_i(O), c_i(O),
                          c_i(1 if relu else 0), _ptr(y))
    return y


def corr_patch(f1, f2, patch_size=11, dilation_patch=1, scale=1.0, leaky=1.0):
    """spatial_correlation_sample(kernel_size=1, stride=1, padding=0) -> [B,P,P,H,W]"""
    f1, f2 = _f32(f1), _f32(f2)
    B, C, H, W = f1.shape
    out = torch.empty(B, patch_size, patch_size, H, W, dtype=torch.float32)
    lib().orc_corr_patch(_ptr(f1), _ptr(f2), c_i(B), c_i(C), c_i(H), c_i(W), c_i(patch_size), c_i(dilation_patch),
                         c_d(scale), c_d(leaky), _ptr(out))
    return out


def correlate(f1, f2, patch_size=11, dilation_patch=1):
    """track_to_segment_head.py:40-62: sampler, /C, leaky_relu(0.1) -> [B,P*P,H,W]"""
    B, C, H, W = f1.shape
    return corr_patch(f1, f2, patch_size, dilation_patch, 1.0 / C, 0.1).view(B, patch_size * patch_size, H, W)


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True):
    """mmcv.ops.roi_align forward (avg) -> [n,C,ph,pw]"""
    assert pool_mode == "avg"
    feat, rois = _f32(feat), _f32(rois)
    ph, pw = _pair(output_size)
    B, C, H, W = feat.shape
    n = rois.shape[0]
    out = torch.empty(n, C, ph, pw, dtype=torch.float32)
    lib().orc_roi_align_avg(_ptr(feat), c_i(B), c_i(C), c_i(H), c_i(W), _ptr(rois), c_l(n), c_i(ph), c_i(pw),
                            c_f(spatial_scale), c_i(sampling_ratio), c_i(1 if aligned else 0), _ptr(out))
    return out


# ------------------------------------------------------------------------------ output stage
def mask_resize_threshold(mask, crop_h, crop_w, out_h, out_w, thr=0.5):
    """output_utils.py:85-95 for one mask [mh,mw] -> uint8 [out_h,out_w]"""
    mask = _f32(mask)
    mh, mw = mask.shape
    out = torch.empty(out_h, out_w, dtype=torch.uint8)
    lib().orc_mask_resize_threshold(_ptr(mask), c_i(mh), c_i(mw), c_i(crop_h), c_i(crop_w), c_i(out_h), c_i(out_w), c_f(thr),
                                    _ptr(out))
    return out


def rle_encode(img):
    """COCO RLE counts (column-major, starting with the zero run) of a binary [h,w] image -> int64 tensor"""
    img = img.to(torch.uint8).contiguous()
    h, w = img.shape
    counts = torch.empty(h * w + 1, dtype=torch.int32)
    lib().orc_rle_encode.restype = c_l
    n = lib().orc_rle_encode(_ptr(img), c_i(h), c_i(w), _ptr(counts), c_l(counts.numel()))
    return counts[:n].to(torch.int64)


def rle_to_string(counts):
    c = counts.to(torch.int32).contiguous()
    buf = ctypes.create_string_buffer(6 * c.numel() + 1)
    lib().orc_rle_to_string.restype = c_l
    n = lib().orc_rle_to_string(_ptr(c), c_l(c.numel()), buf)
    return buf.raw[:n]


def rle_from_string(s, max_runs=1 << 22):
    counts = torch.empty(max_runs, dtype=torch.int32)
    lib().orc_rle_from_string.restype = c_l
    n = lib().orc_rle_from_string(ctypes.c_char_p(bytes(s)), _ptr(counts), c_l(max_runs))
    return counts[:n].to(torch.int64)


def rle_decode(counts, h, w):
    """inverse of rle_encode -> uint8 [h,w]"""
    flat = torch.repeat_interleave(torch.arange(len(counts)) % 2, counts.to(torch.int64)).to(torch.uint8)
    return flat.view(w, h).t().contiguous()


# ------------------------------------------------------------------------------ pre-processing (row f3)
def preprocess_frames(img_u8, size=(640, 360), divisor=32, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375),
                      mode=1):
    """eval.py:703-717: uint8 [n,H0,W0,3] -> fp32 [n,3,Hp,Wp] (cv2-style 8-bit bilinear resize to size=(w,h), numpy
    normalisation in float64, zero pad to a multiple of `divisor`, CHW)."""
    img = img_u8.to(torch.uint8).contiguous()
    n, H0, W0, c = img.shape
    assert c == 3
    w, h = size
    Hp, Wp = -(-h // divisor) * divisor, -(-w // divisor) * divisor
    out = torch.empty(n, 3, Hp, Wp, dtype=torch.float32)
    m3, s3 = (c_d * 3)(*mean), (c_d * 3)(*std)
    lib().orc_preprocess_u8(_ptr(img), c_i(n), c_i(H0), c_i(W0), c_i(h), c_i(w), c_i(Hp), c_i(Wp), m3, s3, c_i(mode), _ptr(out))
    return out


# --------------------------------------------------------------- reference-import stand-ins
class OracleDCN(nn.Module):
    """CPU stand-in for dcn_v2.DCN (backbone.py:21-26,45): same parameters and state-dict keys."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        n = in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        ch = deformable_groups * 3 * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = nn.Conv2d(in_channels, ch, self.kernel_size, self.stride, self.padding, bias=True)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, x):
        out = self.conv_offset_mask(x)
        o1, o2, mask = torch.chunk(out, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(mask)
        return deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)


class OracleDeformConv2d(nn.Module):
    """CPU stand-in for mmcv.ops.DeformConv2d (Featurealign.py:27-31,72): weight only, no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False):
        super().__init__()
        assert not bias and groups == 1
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deform_groups = deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        n = in_channels * self.kernel_size[0] * self.kernel_size[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset):
        return deform_conv(x, offset, None, self.weight, None, self.stride, self.padding, self.dilation,
                           self.deform_groups)


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    assert kernel_size == 1 and stride == 1 and padding == 0 and dilation == 1
    return corr_patch(input1, input2, patch_size, dilation_patch)
