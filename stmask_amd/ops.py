"""Tensor-level wrappers over the C ABI (include/stmask_hip.h).

Each function takes CUDA(=HIP) torch tensors, allocates outputs with torch (device memory + stream plumbing only)
and enqueues the hand-written gfx950 kernel on the current stream.  CPU tensors are rejected: the product path has
no CPU fallback (oracle/ is test infrastructure and is never imported from here).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import DeformGeom, StmError, c_f, c_i, c_l, c_p, c_sz, check


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise StmError("stmask_amd ops need tensors on the MI355X (got a CPU tensor); there is no CPU fallback")


def _f32c(t):
    if t.dtype != torch.float32:
        raise StmError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return c_p(t.data_ptr()) if t is not None else c_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """The current HIP stream of the current device as a C pointer.  torch.cuda.current_stream() builds a Stream object through four Python layers
    (8 us a call, ~22 calls per single-stream step: 16 % of it, profiles/r06_prof_host_clips1.txt); the raw getter is one C call."""
    if _raw_stream is not None:
        return c_p(_raw_stream(torch.cuda.current_device()))
    return c_p(torch.cuda.current_stream().cuda_stream)


def conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def _geom(x, kernel_size, stride, padding, dilation, dg):
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    Ho, Wo = conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    return DeformGeom(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, dg, Ho, Wo)


def _offset_mask_views(offset, mask, g, fused_om):
    """Returns (off_tensor, off_bstride, mask_tensor_or_None, mask_ptr, mask_bstride)."""
    K, HWo = g.kh * g.kw, g.Ho * g.Wo
    if fused_om is not None:
        # raw conv_offset_mask output [B, dg*3K, Ho, Wo]: dcn_v2 chunks it into (o1, o2, mask) and uses
        # cat(o1, o2) as the offset, i.e. channels [0, 2*dg*K) are the offsets and [2*dg*K, 3*dg*K) the mask logits
        om = _f32c(fused_om)
        if tuple(om.shape) != (g.B, g.dg * 3 * K, g.Ho, g.Wo):
            raise StmError(f"conv_offset_mask output {tuple(om.shape)} != {(g.B, g.dg * 3 * K, g.Ho, g.Wo)}")
        bs = g.dg * 3 * K * HWo
        return om, bs, om, om.data_ptr() + 4 * g.dg * 2 * K * HWo, bs
    offset = _f32c(offset)
    if tuple(offset.shape) != (g.B, g.dg * 2 * K, g.Ho, g.Wo):
        raise StmError(f"offset shape {tuple(offset.shape)} != {(g.B, g.dg * 2 * K, g.Ho, g.Wo)}")
    if mask is not None:
        mask = _f32c(mask)
        if tuple(mask.shape) != (g.B, g.dg * K, g.Ho, g.Wo):
            raise StmError(f"mask shape {tuple(mask.shape)} != {(g.B, g.dg * K, g.Ho, g.Wo)}")
        return offset, g.dg * 2 * K * HWo, mask, mask.data_ptr(), g.dg * K * HWo
    return offset, g.dg * 2 * K * HWo, None, 0, 0


def deform_im2col(x, offset, mask, kernel_size, stride=1, padding=0, dilation=1, deform_groups=1, variant=0,
                  fused_om=None, mask_is_logit=False, out=None):
    """Modulated deformable im2col -> cols [B, C*kh*kw, Ho*Wo] (mask=None -> v1)."""
    _dev(x, offset, mask, fused_om)
    x = _f32c(x)
    g = _geom(x, kernel_size, stride, padding, dilation, deform_groups)
    off, obs, mk, mk_ptr, mbs = _offset_mask_views(offset, mask, g, fused_om)
    cols = out if out is not None else torch.empty(g.B, g.C * g.kh * g.kw, g.Ho * g.Wo, device=x.device, dtype=torch.float32)
    rc = _lib.lib().stm_deform_im2col_f32(_p(x), _p(off), c_l(obs), c_p(mk_ptr), c_l(mbs),
                                          c_i(1 if (mask_is_logit or fused_om is not None) else 0), _p(cols),
                                          ctypes.byref(g), c_i(variant), _stream())
    check(rc, "stm_deform_im2col_f32")
    return cols


_ws_cache = {}
_ws_scope = None   # a dict owned by a pipeline while it captures HIP graphs (workspace_scope)


class workspace_scope:
    """`with workspace_scope(store):` -- every _workspace() request inside comes from `store` (keyed by device and tag, not by
    stream) and the owner of `store` keeps the buffers alive.  BatchedClipPipeline captures its trunk graphs under one: the
    scratch pointers baked into a graph (split-K partial sums, column buffers) then belong to the pipeline, not to the
    throw-away capture stream's slot of the global cache -- PyTorch recycles stream handles from a pool of 32, so a later
    pipeline could otherwise be handed the same (device, stream) key, outgrow the buffer and free it under a live graph.
    A buffer that must grow inside a scope is retired into the store, never freed."""

    def __init__(self, store):
        self.store = store

    def __enter__(self):
        global _ws_scope
        self.saved, _ws_scope = _ws_scope, self.store
        return self.store

    def __exit__(self, *exc):
        global _ws_scope
        _ws_scope = self.saved
        return False


_ws_branch = ""    # suffix of every workspace tag while a side branch of the trunk runs (workspace_branch)


class workspace_branch:
    """`with workspace_branch("proto"):` -- scratch requested inside gets its own buffers (tag + suffix).  PlanarGraph runs independent parts of the
    trunk on a second stream while a HIP graph is captured; inside a workspace_scope scratch is keyed by tag, not by stream, so two branches that
    both park split-K partial sums would otherwise share one buffer."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _ws_branch
        self.saved, _ws_branch = _ws_branch, "/" + self.name
        return self

    def __exit__(self, *exc):
        global _ws_branch
        _ws_branch = self.saved
        return False


def _workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, stream, tag) -- or per (device, tag) of the active workspace_scope
    (cols buffers are GBs at large batch: never allocated per call)."""
    tag = tag + _ws_branch
    if _ws_scope is not None:
        key = (device.index, tag)
        buf = _ws_scope.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                _ws_scope.setdefault("retired", []).append(buf)     # a captured graph may still write into it
            buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
            _ws_scope[key] = buf
        return buf
    key = (device.index, torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_im2col_timing = None  # when a list: (start_event, end_event, algorithmic_bytes) per im2col launch (bench.py roofline)


def im2col_timing(enable):
    """bench.py: time every im2col launch of deform_conv with HIP events on the launch stream (live roofline)."""
    global _im2col_timing
    old = _im2col_timing
    _im2col_timing = [] if enable else None
    return old


_fused_dcn_timing = None  # when a list: (start_event, end_event, algorithmic bytes, algorithmic flops, MFMA products per product) per fused DCN launch


def fused_dcn_timing(enable):
    """bench.py: time every stm_deform_conv_fused_planar_f32 launch with HIP events on the launch stream (live roofline)."""
    global _fused_dcn_timing
    old = _fused_dcn_timing
    _fused_dcn_timing = [] if enable else None
    return old


_conv_timing = None  # when a list: (start_event, end_event, algorithmic_flops) per planar-conv launch (bench.py roofline)


def conv_timing(enable):
    """bench.py: time every stm_conv2d_planar_f32 launch with HIP events on the launch stream (live MFMA roofline)."""
    global _conv_timing
    old = _conv_timing
    _conv_timing = [] if enable else None
    return old


def im2col_algorithmic_bytes(g, has_mask):
    """4 * (C*Hin*Win + (3K | 2K)*dg*Ho*Wo + C*K*Ho*Wo) per image."""
    K, HWo = g.kh * g.kw, g.Ho * g.Wo
    return 4 * g.B * (g.C * g.H * g.W + (3 if has_mask else 2) * K * g.dg * HWo + g.C * K * HWo)


def deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, deform_groups=1, relu=False,
                fused_om=None, mask_is_logit=False):
    """Deformable convolution forward: hand-written im2col + fp32 MFMA GEMM (+bias, +ReLU) -> [B,O,Ho,Wo]."""
    _dev(x, offset, mask, weight, bias, fused_om)
    x, weight = _f32c(x), _f32c(weight)
    O, Cw, kh, kw = weight.shape
    if Cw != x.shape[1]:
        raise StmError("groups != 1 is outside the hot path (STM_EUNSUPPORTED)")
    g = _geom(x, (kh, kw), stride, padding, dilation, deform_groups)
    off, obs, mk, mk_ptr, mbs = _offset_mask_views(offset, mask, g, fused_om)
    bias = _f32c(bias) if bias is not None else None
    y = torch.empty(g.B, O, g.Ho, g.Wo, device=x.device, dtype=torch.float32)
    need = _lib.lib().stm_deform_conv_workspace_bytes(ctypes.byref(g))
    ws = _workspace(need, x.device, "cols")
    if _im2col_timing is not None:  # same two kernels, launched separately so the im2col can be bracketed by events
        logit = c_i(1 if (mask_is_logit or fused_om is not None) else 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().stm_deform_im2col_f32(_p(x), _p(off), c_l(obs), c_p(mk_ptr), c_l(mbs), logit, _p(ws),
                                               ctypes.byref(g), c_i(0), _stream()), "stm_deform_im2col_f32")
        e1.record()
        _im2col_timing.append((e0, e1, im2col_algorithmic_bytes(g, mk_ptr != 0)))
        CK, HWo = g.C * g.kh * g.kw, g.Ho * g.Wo
        cols_bytes = (g.B * CK * HWo * 4 + 255) // 256 * 256   # same split of the workspace as stm_deform_conv_fwd_f32
        part = ws[cols_bytes:]
        check(_lib.lib().stm_gemm_bias_ws_f32(_p(weight), _p(ws), _p(bias), _p(y), c_i(O), c_i(HWo), c_i(CK), c_i(g.B),
                                              c_l(CK * HWo), c_l(O * HWo), c_i(1 if relu else 0), _p(part),
                                              c_sz(part.numel()), _stream()), "stm_gemm_bias_ws_f32")
        return y
    rc = _lib.lib().stm_deform_conv_fwd_f32(_p(x), _p(off), c_l(obs), c_p(mk_ptr), c_l(mbs),
                                            c_i(1 if (mask_is_logit or fused_om is not None) else 0), _p(weight),
                                            _p(bias), _p(y), c_i(O), c_i(1 if relu else 0), ctypes.byref(g), _p(ws),
                                            c_sz(ws.numel()), _stream())
    check(rc, "stm_deform_conv_fwd_f32")
    return y


def gemm_bias(A, Bm, bias=None, relu=False):
    """C[b] = A[M,K] @ B[b][K,N] (+bias[m]) on the fp32 MFMA pipe.  Bm is [K,N] or [batch,K,N]."""
    _dev(A, Bm, bias)
    A, Bm = _f32c(A), _f32c(Bm)
    squeeze = Bm.dim() == 2
    if squeeze:
        Bm = Bm[None]
    batch, K, N = Bm.shape
    M = A.shape[0]
    assert A.shape[1] == K
    C = torch.empty(batch, M, N, device=A.device, dtype=torch.float32)
    ws = _workspace(_lib.lib().stm_gemm_workspace_bytes(c_i(M), c_i(N), c_i(batch)), A.device, "gemm")
    rc = _lib.lib().stm_gemm_bias_ws_f32(_p(A), _p(Bm), _p(_f32c(bias) if bias is not None else None), _p(C), c_i(M),
                                         c_i(N), c_i(K), c_i(batch), c_l(K * N), c_l(M * N), c_i(1 if relu else 0),
                                         _p(ws), c_sz(ws.numel()), _stream())
    check(rc, "stm_gemm_bias_ws_f32")
    return C[0] if squeeze else C


def deform_sample_planar(x_pix, B, H, W, C, offsets, kernel_size, padding, out, out_off, fmt=0):
    """mmcv DeformConv2d's sampling half (no mask, stride 1, one deformable group) for the planar graph: x_pix fp32
    [B*H*W, ld >= C] (a channel slice of a wider pixel-major tensor: only the row stride must be a multiple of 4 floats),
    offsets fp32 [B*H*W, 2*kh*kw] pixel-major -> columns written at pixels [out_off, out_off + B*H*W) of the planes
    out [P, kh*kw*C/32, N, 32] (K index = tap*C + channel).  stm_deform_sample_planar_f32."""
    _dev(x_pix, offsets, out)
    if x_pix.dtype != torch.float32 or x_pix.dim() != 2 or x_pix.stride(1) != 1 or x_pix.shape[0] != B * H * W or x_pix.shape[1] != C:
        raise StmError(f"deform_sample_planar: x must be fp32 [B*H*W, C] with unit channel stride, got {tuple(x_pix.shape)} {x_pix.stride()}")
    offsets = _f32c(offsets)
    kh, kw = _pair(kernel_size)
    ph, pw = _pair(padding)
    if offsets.shape[0] != B * H * W or offsets.shape[1] != 2 * kh * kw:
        raise StmError(f"deform_sample_planar: offsets {tuple(offsets.shape)} != {(B * H * W, 2 * kh * kw)}")
    if out.dim() != 4 or out.shape[1] * 32 != kh * kw * C or not out.is_contiguous():
        raise StmError(f"deform_sample_planar: planes {tuple(out.shape)} do not hold {kh * kw * C} column channels")
    g = DeformGeom(B, C, H, W, kh, kw, 1, 1, ph, pw, 1, 1, 1, H, W)
    check(_lib.lib().stm_deform_sample_planar_f32(_p(x_pix), c_i(x_pix.stride(0)), _p(offsets), c_i(offsets.shape[1]), c_i(0), _p(out),
                                                  c_i(out.shape[2]), c_i(out_off), c_l(0), ctypes.byref(g), c_i(fmt), _stream()),
          "stm_deform_sample_planar_f32")
    return out


def fcb_ali_offsets(loc, kh, kw):
    """Featurealign.py:46-69: loc [B,4,H,W] -> offsets [B,2*kh*kw,H,W]."""
    _dev(loc)
    loc = _f32c(loc)
    B, four, H, W = loc.shape
    assert four == 4
    off = torch.empty(B, 2 * kh * kw, H, W, device=loc.device, dtype=torch.float32)
    check(_lib.lib().stm_fcb_ali_offsets_f32(_p(loc), _p(off), c_i(B), c_i(H), c_i(W), c_i(kh), c_i(kw), _stream()),
          "stm_fcb_ali_offsets_f32")
    return off


def corr_patch(f1, f2, patch_size=11, dilation_patch=1, scale=1.0, leaky_slope=1.0):
    """spatial_correlation_sample(kernel_size=1, stride=1, padding=0) -> [B,P,P,H,W] (optionally scaled + leaky)."""
    _dev(f1, f2)
    f1, f2 = _f32c(f1), _f32c(f2)
    if f1.shape != f2.shape:
        raise StmError(f"correlation inputs differ in shape: {tuple(f1.shape)} vs {tuple(f2.shape)}")
    B, C, H, W = f1.shape
    out = torch.empty(B, patch_size, patch_size, H, W, device=f1.device, dtype=torch.float32)
    check(_lib.lib().stm_corr_patch_f32(_p(f1), _p(f2), _p(out), c_i(B), c_i(C), c_i(H), c_i(W), c_i(patch_size),
                                        c_i(dilation_patch), c_f(scale), c_f(leaky_slope), _stream()), "stm_corr_patch_f32")
    return out


def corr_patch_nhwc(f1, f2, patch_size=11, scale=1.0, leaky_slope=1.0, ld=None):
    """corr_patch with the displacement channels last: [B, H, W, ld] (ld >= P*P, default P*P rounded up to 8; channels past P*P are
    not written) -- the layout roi_align_planes(corr_nhwc=...) gathers from.  f1 / f2 are [B, C, H, W] tensors in either memory
    format: channels_last ones (the trunk's outputs) are read in place."""
    _dev(f1, f2)
    if f1.shape != f2.shape:
        raise StmError(f"correlation inputs differ in shape: {tuple(f1.shape)} vs {tuple(f2.shape)}")
    B, C, H, W = f1.shape
    cl = (f1.dtype == f2.dtype == torch.float32 and C > 1 and f1.is_contiguous(memory_format=torch.channels_last)
          and f2.is_contiguous(memory_format=torch.channels_last) and not f1.is_contiguous())
    if not cl:
        f1, f2 = _f32c(f1), _f32c(f2)
    ld = ld or -(-patch_size * patch_size // 8) * 8
    out = torch.empty(B, H, W, ld, device=f1.device, dtype=torch.float32)
    check(_lib.lib().stm_corr_patch_nhwc_f32(_p(f1), _p(f2), _p(out), c_i(B), c_i(C), c_i(H), c_i(W), c_i(patch_size), c_i(1), c_f(scale),
                                             c_f(leaky_slope), c_i(ld), c_i(1 if cl else 0), _stream()), "stm_corr_patch_nhwc_f32")
    return out


def roi_align(feat, rois, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    _dev(feat, rois)
    feat, rois = _f32c(feat), _f32c(rois)
    ph, pw = _pair(output_size)
    B, C, H, W = feat.shape
    n = rois.shape[0]
    if n and rois.shape[1] != 5:
        raise StmError("rois must be [n,5] = (batch, x1, y1, x2, y2)")
    out = torch.empty(n, C, ph, pw, device=feat.device, dtype=torch.float32)
    check(_lib.lib().stm_roi_align_avg_f32(_p(feat), _p(rois), _p(out), c_i(B), c_i(C), c_i(H), c_i(W), c_i(n), c_i(ph),
                                           c_i(pw), c_f(spatial_scale), c_i(sampling_ratio), c_i(1 if aligned else 0),
                                           _stream()), "stm_roi_align_avg_f32")
    return out


def decode(loc, priors):
    """box_utils.py:238-283, bit-exact vs the oracle."""
    _dev(loc, priors)
    loc, priors = _f32c(loc), _f32c(priors)
    boxes = torch.empty_like(loc)
    check(_lib.lib().stm_decode_boxes_f32(_p(loc), _p(priors), _p(boxes), c_l(loc.shape[0]), _stream()),
          "stm_decode_boxes_f32")
    return boxes


def generate_candidates(loc, priors, conf, thresh=0.05):
    """TF_utils.py:54-82 core.  loc [B,N,4], priors [N,4], conf [B,N,ncls] soft-maxed ->
    keep_idx [B,N] (first count[b] valid, ascending), cand_box [B,N,4], count [B] (device int32)."""
    _dev(loc, priors, conf)
    loc, priors, conf = _f32c(loc), _f32c(priors), _f32c(conf)
    B, N, ncls = conf.shape
    keep_idx = torch.empty(B, N, dtype=torch.int64, device=conf.device)
    cand_box = torch.empty(B, N, 4, dtype=torch.float32, device=conf.device)
    count = torch.empty(B, dtype=torch.int32, device=conf.device)
    check(_lib.lib().stm_generate_candidates_f32(_p(loc), _p(priors), _p(conf), c_i(N), c_i(ncls), c_f(thresh), c_i(B),
                                                 _p(keep_idx), _p(cand_box), _p(count), _stream()),
          "stm_generate_candidates_f32")
    return keep_idx, cand_box, count


NMS_LDS_KEYS = 16384   # keys the one-workgroup Fast NMS sorts in LDS (csrc/postproc.hip NMS_MAX_KEYS)


def cc_fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200, k_dev=None):
    """detection_TF.py:85-134 on candidate rows.  conf [K,ncls] or [B,K,ncls].  Returns padded
    (idx [B,top_k] int64, cls, score, box [B,top_k,4], count [B] int32) -- all on device, no sync."""
    _dev(conf, boxes, centerness)
    conf, boxes = _f32c(conf), _f32c(boxes)
    squeeze = conf.dim() == 2
    if squeeze:
        conf, boxes = conf[None], boxes[None]
        centerness = centerness[None] if centerness is not None else None
    B, K, ncls = conf.shape
    cen = _f32c(centerness) if centerness is not None else None
    dev = conf.device
    idx = torch.empty(B, top_k, dtype=torch.int64, device=dev)
    cls = torch.empty(B, top_k, dtype=torch.int64, device=dev)
    sc = torch.empty(B, top_k, dtype=torch.float32, device=dev)
    bx = torch.empty(B, top_k, 4, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    if K > NMS_LDS_KEYS and k_dev is None:
        # more candidate rows than the one-workgroup LDS sort holds: scores to a workspace, exact top-k select in the kernel
        ws = _workspace(_lib.lib().stm_cc_fast_nms_workspace_bytes(c_i(K), c_i(B)), dev, "ccnms")
        check(_lib.lib().stm_cc_fast_nms_ws_f32(_p(conf), _p(boxes), _p(cen), c_i(K), c_i(ncls), c_f(iou_thr), c_i(top_k), c_i(B),
                                                _p(idx), _p(cls), _p(sc), _p(bx), _p(cnt), _p(ws), c_sz(ws.numel()), _stream()),
              "stm_cc_fast_nms_ws_f32")
    else:
        check(_lib.lib().stm_cc_fast_nms_f32(_p(conf), _p(boxes), _p(cen), c_i(K), c_i(ncls), _p(k_dev), c_f(iou_thr),
                                             c_i(top_k), c_i(B), _p(idx), _p(cls), _p(sc), _p(bx), _p(cnt), _stream()),
              "stm_cc_fast_nms_f32")
    if squeeze:
        return idx[0], cls[0], sc[0], bx[0], cnt[0]
    return idx, cls, sc, bx, cnt


def detect_cc(loc, priors, conf, centerness, conf_thresh=0.05, iou_thr=0.5, top_k=200, logits=False):
    """Fused generate_candidate + cc_fast_nms (STMask.py:317-320) with no host sync.
    loc [B,N,4], priors [N,4], conf [B,N,ncls] soft-maxed (logits=True: the raw class logits, soft-maxed per row inside the kernel:
    stm_detect_cc_logits_f32), centerness [B,N] or [B,N,1] -> (prior_idx [B,top_k], cls, score, box [B,top_k,4], count [B])."""
    _dev(loc, priors, conf, centerness)
    loc, priors, conf = _f32c(loc), _f32c(priors), _f32c(conf)
    B, N, ncls = conf.shape
    cen = _f32c(centerness.reshape(B, N)) if centerness is not None else None
    dev = conf.device
    idx = torch.empty(B, top_k, dtype=torch.int64, device=dev)
    cls = torch.empty(B, top_k, dtype=torch.int64, device=dev)
    sc = torch.empty(B, top_k, dtype=torch.float32, device=dev)
    bx = torch.empty(B, top_k, 4, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    need = _lib.lib().stm_detect_cc_workspace_bytes(c_i(N), c_i(B))
    ws = _workspace(need, dev, "detect")
    fn = _lib.lib().stm_detect_cc_logits_f32 if logits else _lib.lib().stm_detect_cc_f32
    check(fn(_p(loc), _p(priors), _p(conf), _p(cen), c_i(N), c_i(ncls), c_f(conf_thresh), c_f(iou_thr), c_i(top_k), c_i(B), _p(idx), _p(cls),
             _p(sc), _p(bx), _p(cnt), _p(ws), c_sz(ws.numel()), _stream()), "stm_detect_cc_f32")
    return idx, cls, sc, bx, cnt


def fast_nms(conf, boxes, centerness, iou_thr=0.5, top_k=200, conf_thresh=0.05, max_det=100, k_dev=None):
    """detection_TF.py:136-204 (per-class).  Returns padded (idx, cls, score, box, count)."""
    _dev(conf, boxes, centerness)
    conf, boxes = _f32c(conf), _f32c(boxes)
    K, ncls = conf.shape
    cen = _f32c(centerness) if centerness is not None else None
    dev = conf.device
    idx = torch.empty(max_det, dtype=torch.int64, device=dev)
    cls = torch.empty(max_det, dtype=torch.int64, device=dev)
    sc = torch.empty(max_det, dtype=torch.float32, device=dev)
    bx = torch.empty(max_det, 4, dtype=torch.float32, device=dev)
    cnt = torch.empty(1, dtype=torch.int32, device=dev)
    need = _lib.lib().stm_fast_nms_workspace_bytes(c_i(K), c_i(ncls), c_i(top_k))
    ws = _workspace(need, dev, "pcnms")
    check(_lib.lib().stm_fast_nms_f32(_p(conf), _p(boxes), _p(cen), c_i(K), c_i(ncls), _p(k_dev), c_f(iou_thr), c_i(top_k),
                                      c_f(conf_thresh), c_i(max_det), _p(idx), _p(cls), _p(sc), _p(bx), _p(cnt), _p(ws),
                                      c_sz(ws.numel()), _stream()), "stm_fast_nms_f32")
    return idx, cls, sc, bx, cnt[0]


def detect_pc(loc, priors, conf, centerness, conf_thresh=0.05, iou_thr=0.5, top_k=200, max_det=100):
    """generate_candidate + per-class Fast NMS (detection_TF.py:136-204) for a whole frame batch with no host sync: the candidate pass
    (stm_generate_candidates_f32) leaves keep lists, compacted boxes and counts on the device, and ONE launch pair of stm_fast_nms_batched_f32 runs
    every frame's 40 class sorts through the keep lists.  loc [B,N,4], priors [N,4], conf [B,N,ncls] SOFT-MAXED, centerness [B,N] or [B,N,1] ->
    (prior_idx [B,max_det], cls, score, box [B,max_det,4], count [B]), the layout detect_cc returns."""
    _dev(loc, priors, conf, centerness)
    loc, priors, conf = _f32c(loc), _f32c(priors), _f32c(conf)
    B, N, ncls = conf.shape
    cen = _f32c(centerness.reshape(B, N)) if centerness is not None else None
    dev = conf.device
    if N > NMS_LDS_KEYS:
        raise StmError(f"detect_pc: {N} priors exceed the {NMS_LDS_KEYS} keys of the per-class LDS sort (use the per-frame layer API)")
    keep_idx, cand_box, count = generate_candidates(loc, priors, conf, conf_thresh)
    idx = torch.empty(B, max_det, dtype=torch.int64, device=dev)
    cls = torch.empty(B, max_det, dtype=torch.int64, device=dev)
    sc = torch.empty(B, max_det, dtype=torch.float32, device=dev)
    bx = torch.empty(B, max_det, 4, dtype=torch.float32, device=dev)
    cnt = torch.empty(B, dtype=torch.int32, device=dev)
    ws = _workspace(_lib.lib().stm_fast_nms_batched_workspace_bytes(c_i(N), c_i(ncls), c_i(top_k), c_i(B)), dev, "pcnms_b")
    check(_lib.lib().stm_fast_nms_batched_f32(_p(conf), c_l(N * ncls), _p(keep_idx), _p(cand_box), _p(cen), c_l(N), c_i(N), c_i(ncls), _p(count), c_f(iou_thr),
                                              c_i(top_k), c_f(conf_thresh), c_i(max_det), c_i(B), _p(idx), _p(cls), _p(sc), _p(bx), _p(cnt), _p(ws),
                                              c_sz(ws.numel()), _stream()), "stm_fast_nms_batched_f32")
    return idx, cls, sc, bx, cnt


def jaccard(a, b):
    """box_utils.py:60-88 (2-D form), bit-exact."""
    _dev(a, b)
    a, b = _f32c(a), _f32c(b)
    out = torch.empty(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    check(_lib.lib().stm_jaccard_f32(_p(a), c_i(a.shape[0]), _p(b), c_i(b.shape[0]), _p(out), _stream()), "stm_jaccard_f32")
    return out


def lincomb_sigmoid_crop_bits(proto, coeff, boxes, row_proto, thr=0.5, apply_tanh=True):
    """lincomb_sigmoid_crop that also returns the binarised masks (value > thr) bit-packed, [n, ceil(h*w/64)] int64 words -- the
    form mask_iou_bits consumes, produced in the pass that writes the soft masks instead of a second read of them."""
    _dev(proto, coeff, boxes, row_proto)
    proto, coeff = _f32c(proto), _f32c(coeff)
    assert proto.dim() == 4 and row_proto.dtype == torch.int32 and row_proto.numel() == coeff.shape[0]
    h, w, m = proto.shape[1:]
    n = coeff.shape[0]
    out = torch.empty(n, h, w, dtype=torch.float32, device=proto.device)
    bits = torch.empty(n, (h * w + 63) // 64, dtype=torch.int64, device=proto.device)
    check(_lib.lib().stm_lincomb_sigmoid_crop_bits_f32(_p(proto), _p(coeff), _p(_f32c(boxes)), _p(out), c_i(h), c_i(w), c_i(m), c_i(n),
                                                       c_i(1 if apply_tanh else 0), c_p(0), _p(row_proto), _p(bits), c_f(thr), _stream()),
          "stm_lincomb_sigmoid_crop_bits_f32")
    return out, bits


def mask_iou_bits(bits1, bits2, hw, group1=None, group2=None):
    """box_utils.py:435-447 on bit-packed binary masks (lincomb_sigmoid_crop_bits) -> [n1, n2]; groups as in mask_iou."""
    _dev(bits1, bits2, group1, group2)
    n1, n2 = bits1.shape[0], bits2.shape[0]
    if n1 == 0 or n2 == 0:
        return torch.zeros(n1, n2, dtype=torch.float32, device=bits1.device)
    out = torch.empty(n1, n2, dtype=torch.float32, device=bits1.device)     # (the kernel writes every element: zeros for pairs of different groups)
    if bits1.dtype != torch.int64 or bits2.dtype != torch.int64 or bits1.shape[1] != (hw + 63) // 64 or bits2.shape[1] != bits1.shape[1]:
        raise StmError("mask_iou_bits: bit tables must be int64 [n, ceil(hw / 64)]")
    check(_lib.lib().stm_mask_iou_bits_f32(_p(bits1.contiguous()), c_i(n1), _p(bits2.contiguous()), c_i(n2), c_i(hw), _p(out),
                                           _p(group1.contiguous()) if group1 is not None else c_p(0),
                                           _p(group2.contiguous()) if group2 is not None else c_p(0), _stream()), "stm_mask_iou_bits_f32")
    return out


def lincomb_sigmoid_crop(proto, coeff, boxes=None, apply_tanh=True, n_dev=None, row_proto=None):
    """generate_mask (mask_utils.py:111-128) + crop.  proto [h,w,m], coeff [n,m], boxes [n,4] -> [n,h,w].
    With row_proto (int32 [n]) proto is [P,h,w,m] and row i uses proto[row_proto[i]] (rows of many clips, one launch)."""
    _dev(proto, coeff, boxes, row_proto)
    proto, coeff = _f32c(proto), _f32c(coeff)
    if row_proto is not None:
        assert proto.dim() == 4 and row_proto.dtype == torch.int32 and row_proto.numel() == coeff.shape[0]
        h, w, m = proto.shape[1:]
    else:
        h, w, m = proto.shape
    n = coeff.shape[0]
    bx = _f32c(boxes) if boxes is not None else None
    out = torch.empty(n, h, w, dtype=torch.float32, device=proto.device)
    check(_lib.lib().stm_lincomb_sigmoid_crop_f32(_p(proto), _p(coeff), _p(bx), _p(out), c_i(h), c_i(w), c_i(m), c_i(n),
                                                  c_i(1 if apply_tanh else 0), _p(n_dev), _p(row_proto), _stream()),
          "stm_lincomb_sigmoid_crop_f32")
    return out


def mask_iou(m1, m2, thr=0.5, group1=None, group2=None):
    """box_utils.py:435-447 on (m > thr).  m1 [n1,h,w], m2 [n2,h,w] soft masks -> [n1,n2].  group1 / group2 (int32, any
    order; sorted rows skip whole workgroups): only pairs of the same group are computed, the others stay 0 (stm_mask_iou_grouped_f32)."""
    _dev(m1, m2, group1, group2)
    m1, m2 = _f32c(m1), _f32c(m2)
    n1, n2 = m1.shape[0], m2.shape[0]
    out = torch.zeros(n1, n2, dtype=torch.float32, device=m1.device)
    if n1 == 0 or n2 == 0:
        return out
    hw = m1[0].numel()
    need = _lib.lib().stm_mask_iou_workspace_bytes(c_i(n1), c_i(n2), c_i(hw))
    ws = _workspace(need, m1.device, "miou")
    if group1 is not None:
        if group1.dtype != torch.int32 or group2.dtype != torch.int32 or group1.numel() != n1 or group2.numel() != n2:
            raise StmError("mask_iou: group arrays must be int32 of lengths n1 and n2")
        check(_lib.lib().stm_mask_iou_grouped_f32(_p(m1), c_i(n1), _p(m2), c_i(n2), c_i(hw), c_f(thr), _p(out), _p(group1.contiguous()),
                                                  _p(group2.contiguous()), _p(ws), c_sz(ws.numel()), _stream()), "stm_mask_iou_grouped_f32")
        return out
    check(_lib.lib().stm_mask_iou_f32(_p(m1), c_i(n1), _p(m2), c_i(n2), c_i(hw), c_f(thr), _p(out), _p(ws), c_sz(ws.numel()),
                                      _stream()), "stm_mask_iou_f32")
    return out


def bias_act_(y, bias, residual=None, relu=True):
    """In place y = act(y + bias[c] (+ residual)) for a 4-D activation in NCHW-contiguous or channels_last layout
    (fused BN-folded conv epilogue).  Returns y."""
    _dev(y, bias, residual)
    if y.dtype != torch.float32 or y.dim() != 4:
        raise StmError("bias_act_ expects a 4-D float32 activation")
    B, C, H, W = y.shape
    if y.is_contiguous():
        inner = H * W
    elif y.is_contiguous(memory_format=torch.channels_last):
        inner = 1
    else:
        raise StmError("bias_act_: activation must be NCHW-contiguous or channels_last")
    if residual is not None:
        if residual.shape != y.shape or residual.stride() != y.stride():
            residual = residual.contiguous(memory_format=torch.channels_last if inner == 1 else torch.contiguous_format)
    check(_lib.lib().stm_bias_act_f32(_p(y), _p(_f32c(bias)), _p(residual), c_l(y.numel()), c_i(C), c_l(inner),
                                      c_i(1 if relu else 0), _stream()), "stm_bias_act_f32")
    return y


def mask_resize_rle(masks, crop_h, crop_w, out_h, out_w, thr=0.5, max_runs=4096):
    """Mask leg of postprocess_ytbvis on the device: [n,mh,mw] soft masks -> (counts [n,max_runs] int32, n_runs [n] int32)
    = COCO run lengths of the un-padded, bilinearly resized, thresholded masks (column-major)."""
    _dev(masks)
    masks = _f32c(masks)
    n, mh, mw = masks.shape
    counts = torch.zeros(n, max_runs, dtype=torch.int32, device=masks.device)
    n_runs = torch.zeros(n, dtype=torch.int32, device=masks.device)
    if n == 0:
        return counts, n_runs
    need = _lib.lib().stm_mask_rle_workspace_bytes(c_i(n), c_i(out_h), c_i(out_w), c_i(max_runs))
    ws = _workspace(need, masks.device, "rle")
    check(_lib.lib().stm_mask_resize_rle_f32(_p(masks), c_i(n), c_i(mh), c_i(mw), c_i(crop_h), c_i(crop_w), c_i(out_h),
                                             c_i(out_w), c_f(thr), _p(counts), c_i(max_runs), _p(n_runs), _p(ws),
                                             c_sz(ws.numel()), _stream()), "stm_mask_resize_rle_f32")
    return counts, n_runs


def conv_pack_weights(weight, planes=3, tile_n=128, fmt=0, wscale=None):
    """OIHW fp32 weights -> the pre-split, pre-tiled image the convolution kernels stream (done once per layer).
    tile_n = 64 packs for the 128 x 64-tile planar kernel (stm_conv_geom.tile_n must say so too).
    fmt = 1 / 2: two / one fp16 plane(s) of weight * wscale (power of two bringing max |w| to ~2^10) -> returns
    (packed, 1 / wscale)."""
    _dev(weight)
    weight = _f32c(weight)
    O, C, kh, kw = weight.shape
    if fmt >= 1:
        planes = 2 if fmt == 1 else 1
    nbytes = _lib.lib().stm_conv_packed_weight_bytes_tiled(c_i(O), c_i(C), c_i(kh), c_i(kw), c_i(planes), c_i(tile_n))
    if nbytes == 0:
        raise StmError(f"conv_pack_weights: unsupported weight shape {tuple(weight.shape)} (Cin must be a multiple of 32)")
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    if fmt >= 1:
        import math
        if wscale is None:          # (given: several weight tensors packed under one scale, e.g. the sub-kernels of a window set)
            wmax = float(weight.abs().max())
            wscale = 2.0 ** (10 - math.floor(math.log2(wmax))) if wmax > 0 else 1.0
        check(_lib.lib().stm_conv_pack_weights_fmt_f32(_p(weight), _p(packed), c_i(O), c_i(C), c_i(kh), c_i(kw), c_i(tile_n), c_i(fmt),
                                                       c_f(wscale), _stream()), "stm_conv_pack_weights_fmt_f32")
        return packed, 1.0 / wscale
    check(_lib.lib().stm_conv_pack_weights_tiled_f32(_p(weight), _p(packed), c_i(O), c_i(C), c_i(kh), c_i(kw), c_i(planes),
                                                     c_i(tile_n), _stream()), "stm_conv_pack_weights_tiled_f32")
    return packed


def conv2d_planar_windows(xp, packed_list, windows, bias, B, H, W, C, O, out_h, out_w, out_scale, relu=True, out_f32=None, out_planes=None):
    """stm_conv2d_planar_windows_f32: several window launches of one fp16x2 layer as one grid.  xp [2, C/32, >= B*H*W, 32]; packed_list[i] /
    windows[i] = (kh, kw, ph, pw, Ho, Wo, y0, x0): sub-kernel weights (conv_pack_weights(..., tile_n=128, fmt=1, wscale=common)) and window of
    the out_h x out_w output image; out_f32 [B*out_h*out_w, O] and / or out_planes [2, O/32, B*out_h*out_w, 32] are written in place."""
    _dev(xp)
    if out_f32 is None and out_planes is None:
        raise StmError("conv2d_planar_windows: no output given")
    n = len(windows)
    g = _lib.ConvGeom()
    g.B, g.H, g.W, g.C, g.Cout, g.sh, g.sw, g.planes, g.fmt, g.tile_n = B, H, W, C, O, 1, 1, 2, 1, 128
    g.kh, g.kw, g.Ho, g.Wo = windows[0][0], windows[0][1], windows[0][4], windows[0][5]
    g.win_h, g.win_w = out_h, out_w
    g.out_scale = out_scale
    g.x_np, g.x_plane_stride = xp.shape[2], xp.shape[1] * xp.shape[2] * 32
    if out_planes is not None:
        g.out_np, g.out_plane_stride = out_planes.shape[2], out_planes.shape[1] * out_planes.shape[2] * 32
    if out_f32 is not None:
        g.out_ld = out_f32.shape[-1]
    wins = (_lib.ConvWindow * n)()
    for i, wv in enumerate(windows):
        wins[i].kh, wins[i].kw, wins[i].ph, wins[i].pw, wins[i].Ho, wins[i].Wo, wins[i].y0, wins[i].x0 = wv
    ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p in packed_list])
    check(_lib.lib().stm_conv2d_planar_windows_f32(_p(xp), ptrs, wins, c_i(n), _p(bias) if bias is not None else None,
                                                   _p(out_f32) if out_f32 is not None else None, _p(out_planes) if out_planes is not None else None,
                                                   ctypes.byref(g), c_i(1 if relu else 0), _stream()), "stm_conv2d_planar_windows_f32")
    return out_f32 if out_f32 is not None else out_planes


def conv2d_planar_windows_pool(xp, packed_list, windows, bias, B, H, W, C, O, out_h, out_w, out_scale, pool_fix):
    """stm_conv2d_planar_windows_pool_f32: the window set of conv2d_planar_windows with ReLU and the average pool over each out_h x out_w image in
    the epilogue: pool_fix [>= B, O] int64 (32.32 fixed point, zero before the first use) receives the pooled SUMS; nothing is written per pixel."""
    _dev(xp)
    if pool_fix.dtype != torch.int64 or pool_fix.dim() != 2 or pool_fix.shape[0] < B or pool_fix.shape[1] != O or not pool_fix.is_contiguous():
        raise StmError("conv2d_planar_windows_pool: pool_fix must be a contiguous int64 [>= B, O] tensor")
    n = len(windows)
    g = _lib.ConvGeom()
    g.B, g.H, g.W, g.C, g.Cout, g.sh, g.sw, g.planes, g.fmt, g.tile_n = B, H, W, C, O, 1, 1, 2, 1, 128
    g.kh, g.kw, g.Ho, g.Wo = windows[0][0], windows[0][1], windows[0][4], windows[0][5]
    g.win_h, g.win_w = out_h, out_w
    g.out_scale = out_scale
    g.x_np, g.x_plane_stride = xp.shape[2], xp.shape[1] * xp.shape[2] * 32
    wins = (_lib.ConvWindow * n)()
    for i, wv in enumerate(windows):
        wins[i].kh, wins[i].kw, wins[i].ph, wins[i].pw, wins[i].Ho, wins[i].Wo, wins[i].y0, wins[i].x0 = wv
    ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p in packed_list])
    check(_lib.lib().stm_conv2d_planar_windows_pool_f32(_p(xp), ptrs, wins, c_i(n), _p(bias) if bias is not None else None, _p(pool_fix),
                                                        ctypes.byref(g), _stream()), "stm_conv2d_planar_windows_pool_f32")
    return pool_fix


def temporal_pool_fc(pool_fix, n, npix, weight, bias, n_first=None, clear=True, want_pooled=False):
    """stm_temporal_pool_fc_f32: mean = pool_fix / 2^32 / npix, y = mean @ weight.t() + bias for the first n rows of pool_fix [>= n, C] (int64
    fixed-point sums of conv2d_planar_windows_pool); zeroes the consumed rows when clear.  Returns y [n, n_out] -- or, with n_first, the two
    contiguous blocks (y[:, :n_first], y[:, n_first:]) -- and the means [n, C] when want_pooled."""
    _dev(pool_fix, weight)
    if pool_fix.dtype != torch.int64 or pool_fix.dim() != 2 or not pool_fix.is_contiguous() or pool_fix.shape[0] < n or n < 0:
        raise StmError(f"temporal_pool_fc: pool_fix must be contiguous int64 [>= {n}, C], got {pool_fix.dtype} {tuple(pool_fix.shape)}")
    if weight.dim() != 2 or weight.shape[1] != pool_fix.shape[1] or (bias is not None and bias.numel() != weight.shape[0]):
        raise StmError(f"temporal_pool_fc: weight {tuple(weight.shape)} / bias do not match C = {pool_fix.shape[1]}")
    C, n_out = pool_fix.shape[1], weight.shape[0]
    weight = _f32c(weight)
    dev = pool_fix.device
    nf = n_out if n_first is None else int(n_first)
    out = torch.empty(n, nf, device=dev, dtype=torch.float32)
    out2 = torch.empty(n, n_out - nf, device=dev, dtype=torch.float32) if n_first is not None else None
    pooled = torch.empty(n, C, device=dev, dtype=torch.float32) if want_pooled else None
    check(_lib.lib().stm_temporal_pool_fc_f32(_p(pool_fix), c_i(n), c_i(C), c_i(npix), _p(weight), _p(_f32c(bias)) if bias is not None else None,
                                              c_i(n_out), c_i(nf), _p(out), _p(out2) if out2 is not None else None,
                                              _p(pooled) if pooled is not None else None, c_i(1 if clear else 0), _stream()),
          "stm_temporal_pool_fc_f32")
    res = (out, out2) if n_first is not None else (out,)
    if want_pooled:
        res = res + (pooled,)
    return res if len(res) > 1 else res[0]


def conv_kxr_supported(O, C, kh, kw, stride, padding, groups, group_cout, fmt, max_tiles=4):
    """Layers the kx-reuse narrow-output kernel (csrc/conv_kxr.hip, stm_conv2d_planar_kxr_f32) takes: stride 1, same padding,
    kw = 3 or 5, fp16 plane formats, at most 16 * max_tiles real output channels per group, at most 4 groups, and a three-stage
    LDS ring that fits (the library refuses the others)."""
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    if fmt not in (1, 2) or (sh, sw) != (1, 1) or kw not in (3, 5) or not (1 <= kh <= 8) or 2 * ph != kh - 1 or 2 * pw != kw - 1:
        return False
    if C % 32 or groups < 1 or groups > 4 or O % groups:
        return False
    cg = O // groups
    real = [(group_cout[i] if group_cout and 0 < group_cout[i] < cg else cg) for i in range(groups)]
    npl = 2 if fmt == 1 else 1
    for r in real:
        nc = -(-r // 16)
        if nc > max_tiles or 16 * nc > cg or cg % 4:       # whole 16-channel tiles are written: they must fit the group's row stride
            return False
        # csrc/conv_kxr.hip kx_pt / kx_depth: some tile of 64 * pt pixels (pt = 4, 3, 2) must leave room for three stages
        if 3 * (npl * (64 * 2 + 16) * 64 + kw * npl * nc * 1024) > 160 * 1024:
            return False
    return True


def conv_pack_weights_kxr(weight, geom):
    """OIHW fp32 weights (grouped layers: group g = rows [g * O / groups, ...)) -> the image stm_conv2d_planar_kxr_f32 streams;
    geom: a ConvGeom with C (per group), Cout, kh, kw, groups, group_cout, fmt set.  Returns (packed, 1 / wscale)."""
    _dev(weight)
    weight = _f32c(weight)
    nbytes = _lib.lib().stm_conv_kxr_packed_bytes(ctypes.byref(geom))
    if nbytes == 0:
        raise StmError("conv_pack_weights_kxr: " + _lib.lib().stm_last_error_string().decode(errors="replace"))
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    import math
    wmax = float(weight.abs().max())
    wscale = 2.0 ** (10 - math.floor(math.log2(wmax))) if wmax > 0 else 1.0
    check(_lib.lib().stm_conv_pack_weights_kxr_f32(_p(weight), _p(packed), ctypes.byref(geom), c_f(wscale), _stream()),
          "stm_conv_pack_weights_kxr_f32")
    return packed, 1.0 / wscale


def _pow2_wscale(weight):
    import math
    wmax = float(weight.abs().max())
    return 2.0 ** (10 - math.floor(math.log2(wmax))) if wmax > 0 else 1.0


def chain_pack_tail(w3, w1_next=None, wds=None):
    """conv3 [256, 64(,1,1)] and (optionally) the next block's conv1 [64, 256(,1,1)] -> the tail image of stm_bottleneck_chain_f32; with
    wds [256, 64(,1,1)] (the projection shortcut of a stage's first block) the image of stm_bottleneck_chain_proj_f32.
    Returns (packed, out_scale3, out_scale1)."""
    _dev(w3)
    w3 = _f32c(w3.reshape(256, 64))
    if wds is not None:
        wds = _f32c(wds.reshape(256, 64))
    ws3 = _pow2_wscale(torch.cat([w3, wds], 1) if wds is not None else w3)
    ws1 = 1.0
    if w1_next is not None:
        w1_next = _f32c(w1_next.reshape(64, 256))
        ws1 = _pow2_wscale(w1_next)
    L = _lib.lib()
    packed = torch.zeros(L.stm_chain_tail_weight_bytes_proj() if wds is not None else L.stm_chain_tail_weight_bytes(), device=w3.device, dtype=torch.uint8)
    pw1 = _p(w1_next) if w1_next is not None else None
    if wds is not None:
        check(L.stm_chain_pack_tail_proj_f32(_p(w3), _p(wds), pw1, _p(packed), c_f(ws3), c_f(ws1), _stream()), "stm_chain_pack_tail_proj_f32")
    else:
        check(L.stm_chain_pack_tail_f32(_p(w3), pw1, _p(packed), c_f(ws3), c_f(ws1), _stream()), "stm_chain_pack_tail_f32")
    return packed, 1.0 / ws3, 1.0 / ws1


def bottleneck_chain(mid1, x, w2_packed, tail_packed, b2, b3, b1_next, scales, B, H, W, y=None, z=None, want_z=True, proj=False):
    """csrc/conv_chain.hip: mid1 [2, 2, BHW, 32] and x [2, 8, BHW, 32] fp16 planes -> (y [2, 8, BHW, 32], z [2, 2, BHW, 32] or None):
    relu(conv3(relu(conv2(mid1))) + x) and the next block's relu(conv1(y)) in one launch (reference backbone.py:38-58).  proj=True: x is
    the 64-channel input [2, 2, BHW, 32] of a stage's first block and the shortcut its projection (tail from chain_pack_tail(..., wds=))."""
    _dev(mid1)
    n = B * H * W
    xs = 2 if proj else 8
    if tuple(mid1.shape) != (2, 2, n, 32) or tuple(x.shape) != (2, xs, n, 32) or mid1.dtype != torch.float16 or x.dtype != torch.float16:
        raise StmError(f"bottleneck_chain: planes {tuple(mid1.shape)} / {tuple(x.shape)} do not match [2, 2, {n}, 32] / [2, {xs}, {n}, 32] fp16")
    if not (mid1.is_contiguous() and x.is_contiguous()):
        raise StmError("bottleneck_chain: planes must be dense")
    if y is None:
        y = torch.empty(2, 8, n, 32, device=mid1.device, dtype=torch.float16)
    if want_z and z is None:
        z = torch.empty(2, 2, n, 32, device=mid1.device, dtype=torch.float16)
    s2, s3, s1 = scales
    fn = _lib.lib().stm_bottleneck_chain_proj_f32 if proj else _lib.lib().stm_bottleneck_chain_f32
    check(fn(_p(mid1), _p(x), _p(y), _p(z) if want_z else None, _p(w2_packed), _p(tail_packed),
             _p(b2) if b2 is not None else None, _p(b3) if b3 is not None else None,
             _p(b1_next) if (want_z and b1_next is not None) else None, c_f(s2), c_f(s3), c_f(s1),
             c_i(B), c_i(H), c_i(W), _stream()), "stm_bottleneck_chain_proj_f32" if proj else "stm_bottleneck_chain_f32")
    return y, (z if want_z else None)


def plane_layout(fmt):
    """(number of planes, element type) of a planar format: 0 = bf16 x 3, 1 = fp16 x 2, 2 = fp16 x 1 (include/stmask_hip.h)."""
    if fmt == 0:
        return 3, torch.bfloat16
    if fmt in (1, 2):
        return (2 if fmt == 1 else 1), torch.float16
    raise StmError(f"unknown planar format {fmt}")


def _empty_planes(fmt, slabs, n, device):
    P, dt = plane_layout(fmt)
    return torch.empty(P, slabs, n, 32, device=device, dtype=dt)


def split_planes(x, fmt=0):
    """fp32 NHWC tensor [..., C] (C % 32 == 0) -> planes [P, C/32, N, 32] (N = product of the leading dims) whose fp32 sum is
    x: fmt 0 = three bf16 planes (exact), fmt 1 = two fp16 planes (22 bits; |x| < 65504).  Channel-slab-major: see
    include/stmask_hip.h."""
    _dev(x)
    x = _f32c(x)
    C = x.shape[-1]
    N = x.numel() // C
    if C % 32:
        raise StmError(f"split_planes: channel count {C} is not a multiple of 32")
    planes = _empty_planes(fmt, C // 32, N, x.device)
    check(_lib.lib().stm_split_planes_fmt_f32(_p(x), _p(planes), c_l(N), c_i(C), c_i(fmt), _stream()), "stm_split_planes_fmt_f32")
    return planes


F16_LOW_SCALE = 2048.0   # fp16 plane format: x = h + l / 2048 (include/stmask_hip.h, stm_conv_geom.fmt)

_range_flags = {}


def planar_range_flag():
    """The sticky fp16-range flag of this device (stm_planar_set_range_flag): an int32 tensor of one element that every
    producer of fp16 planes sets to 1 when it meets |x| > 65504, inf or nan.  Registered with the library on first use."""
    dev = torch.cuda.current_device()
    flag = _range_flags.get(dev)
    if flag is None:
        flag = torch.zeros(1, device=f"cuda:{dev}", dtype=torch.int32)
        with torch.cuda.device(dev):         # the library keeps one flag per device: register on the device that owns it
            check(_lib.lib().stm_planar_set_range_flag(_p(flag)), "stm_planar_set_range_flag")
        _range_flags[dev] = flag
    return flag


class RangeError(StmError):
    """An activation left the range of the fp16 plane formats (the sticky device flag of stm_planar_set_range_flag was found set).  The batched
    pipeline catches it, rebuilds the inference graph with bf16x3 planes (any fp32 range) and repeats the step."""


RANGE_MESSAGE = ("an activation left the range of the fp16x2 planar format (|x| > 65504, inf or nan): results of this step are "
                 "invalid; build the graph with optimize_for_inference(net, planar=True, planes='bf16x3')")


def check_planar_range():
    """Synchronising check of the flag (callers with a host read of their own fold the flag into it instead)."""
    flag = _range_flags.get(torch.cuda.current_device())
    if flag is not None and int(flag.item()):
        flag.zero_()
        raise RangeError(RANGE_MESSAGE)


def counts_to_host(cnt, extra=None):
    """cnt.tolist() for an integer device tensor -- the host read every detection step does -- with the fp16 range flag
    of the device (if a fp16 planar graph registered one) riding in the same copy.  Raises StmError when the flag is set.
    extra: an fp32 tensor that travels in the same copy (the Fast-NMS scores the tracker's host logic needs); returns
    (counts, extra as a flat numpy array) then."""
    flag = _range_flags.get(cnt.device.index) if cnt.is_cuda else None
    if flag is None and extra is None:
        return cnt.tolist()
    parts = [cnt.reshape(-1).to(torch.int32)]
    if flag is not None:
        parts.append(flag)
    if extra is not None:
        parts.append(extra.reshape(-1).view(torch.int32))
    host = torch.cat(parts).cpu()
    n = cnt.numel()
    if flag is not None and int(host[n]):
        flag.zero_()
        raise RangeError(RANGE_MESSAGE)
    counts = host[:n].tolist()
    if extra is None:
        return counts
    return counts, host[n + (1 if flag is not None else 0):].view(torch.float32).numpy()


# ---- tracker bookkeeping of the batched pipeline (csrc/tracker.hip) -----------------------------------------------------
def gather_detections(idx, cls, score, box, cnt, mask_coeff, track, centerness, D):
    """Fast-NMS survivors of detect_cc ([B,top_k] slots, device counts) -> concatenated detection rows (dict of [D, ...]
    tensors: box, class, score, mask_coeff, track, centerness, clip).  D = sum of the counts as read by the host."""
    _dev(idx, cls, score, box, cnt, mask_coeff, track, centerness)
    B, top_k = idx.shape
    N, mdim, edim = mask_coeff.shape[1], mask_coeff.shape[2], track.shape[2]
    dev = idx.device
    out = {"box": torch.empty(D, 4, device=dev), "class": torch.empty(D, dtype=torch.int64, device=dev), "score": torch.empty(D, device=dev),
           "mask_coeff": torch.empty(D, mdim, device=dev), "track": torch.empty(D, edim, device=dev), "centerness": torch.empty(D, device=dev),
           "clip": torch.empty(D, dtype=torch.int32, device=dev)}
    cen = _f32c(centerness.reshape(B, N)) if centerness is not None else None
    check(_lib.lib().stm_gather_detections_f32(_p(idx), _p(cls), _p(_f32c(score)), _p(_f32c(box)), _p(cnt), _p(_f32c(mask_coeff)), _p(_f32c(track)),
                                               _p(cen), c_i(B), c_i(top_k), c_i(N), c_i(mdim), c_i(edim), c_i(D), _p(out["box"]), _p(out["class"]),
                                               _p(out["score"]), _p(out["mask_coeff"]), _p(out["track"]), _p(out["centerness"]), _p(out["clip"]),
                                               _stream()), "stm_gather_detections_f32")
    return out


def shift_rois(box, clip, feat_h, feat_w):
    """CandidateShift's RoIs: [n, 5] = (clip, x1, y1, x2, y2 in feature-map pixels, order-fixed and clamped)."""
    _dev(box, clip)
    n = box.shape[0]
    rois = torch.empty(n, 5, device=box.device)
    check(_lib.lib().stm_shift_rois_f32(_p(_f32c(box)), _p(clip), _p(rois), c_i(n), c_i(feat_h), c_i(feat_w), _stream()), "stm_shift_rois_f32")
    return rois


def shift_apply_(loc_shift, coeff_shift, box, mask_coeff, score, decay=0.95):
    """In place: box = decode(loc_shift, center_size(box)); mask_coeff += coeff_shift; score *= decay (TF_utils.py:40-48)."""
    _dev(loc_shift, coeff_shift, box, mask_coeff, score)
    for t in (box, mask_coeff, score):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise StmError("shift_apply_: box / mask_coeff / score must be contiguous fp32 (modified in place)")
    check(_lib.lib().stm_shift_apply_f32(_p(_f32c(loc_shift)), _p(_f32c(coeff_shift)), _p(box), _p(mask_coeff), _p(score), c_i(box.shape[0]),
                                         c_i(mask_coeff.shape[1]), c_f(decay), _stream()), "stm_shift_apply_f32")


def match_scores(cos, miou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip, prev_offsets, match_coeff, dummy_iou=0.3):
    """compute_comp_scores + argmax over [dummy | prev rows of the same clip] -> int32 [D]: 0 = new instance, 1 + prev row."""
    _dev(cos, miou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip, prev_offsets)
    D, Pn = det_box.shape[0], prev_box.shape[0]
    match = torch.empty(D, dtype=torch.int32, device=det_box.device)
    c4 = (ctypes.c_float * 4)(*[float(v) for v in match_coeff])
    check(_lib.lib().stm_match_scores_f32(_p(_f32c(cos)), _p(_f32c(miou)), _p(_f32c(det_box)), _p(_f32c(prev_box)), _p(_f32c(det_score)),
                                          _p(det_cls), _p(prev_cls), _p(det_clip), _p(prev_offsets), c_i(D), c_i(Pn), c4, c_f(dummy_iou),
                                          _p(match), _stream()), "stm_match_scores_f32")
    return match


def match_scores_embed(det_track, prev_track, miou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip, prev_offsets, match_coeff,
                       dummy_iou=0.3):
    """match_scores with the cosine term computed in the kernel from the track embeddings (same-clip pairs only): no [D, Pn]
    matrix product in front of it."""
    _dev(det_track, prev_track, miou, det_box, prev_box, det_score, det_cls, prev_cls, det_clip, prev_offsets)
    D, Pn = det_box.shape[0], prev_box.shape[0]
    match = torch.empty(D, dtype=torch.int32, device=det_box.device)
    c4 = (ctypes.c_float * 4)(*[float(v) for v in match_coeff])
    check(_lib.lib().stm_match_scores_embed_f32(_p(_f32c(det_track)), _p(_f32c(prev_track)), c_i(det_track.shape[1]), _p(_f32c(miou)),
                                                _p(_f32c(det_box)), _p(_f32c(prev_box)), _p(_f32c(det_score)), _p(det_cls), _p(prev_cls),
                                                _p(det_clip), _p(prev_offsets), c_i(D), c_i(Pn), c4, c_f(dummy_iou), _p(match), _stream()),
          "stm_match_scores_embed_f32")
    return match


def gather_rows2(a_rows, b_rows, plan, n_a):
    """out_t[r] = plan[r] < n_a ? a_t[plan[r]] : b_t[plan[r] - n_a] for lists of row tensors (<= 8 per launch); plan int32."""
    _dev(plan, *a_rows, *b_rows)
    if plan.dtype != torch.int32:
        raise StmError("gather_rows2: plan must be int32")
    R = plan.shape[0]
    outs = []
    for i in range(0, len(a_rows), 8):
        aa, bb = [t.contiguous() for t in a_rows[i:i + 8]], [t.contiguous() for t in b_rows[i:i + 8]]
        oo = [torch.empty((R,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in aa]
        n = len(aa)
        rb = [t[0].numel() * t.element_size() if t.shape[0] else (bb[k][0].numel() * bb[k].element_size()) for k, t in enumerate(aa)]
        pa = (ctypes.c_void_p * n)(*[t.data_ptr() for t in aa])
        pb = (ctypes.c_void_p * n)(*[t.data_ptr() for t in bb])
        po = (ctypes.c_void_p * n)(*[t.data_ptr() for t in oo])
        prb = (ctypes.c_int * n)(*rb)
        check(_lib.lib().stm_gather_rows2(pa, pb, po, prb, c_i(n), _p(plan), c_i(R), c_i(n_a), _stream()), "stm_gather_rows2")
        outs += oo
    return outs


def pack_tracked(mask, score, tracked, offsets, box, cls, mask_coeff, B, top_k, cols, max_age=10, score_thr=0.05):
    """Keep rule of track_TF.py:158-165 + scatter into the fixed-shape [B, top_k, cols] detection rows (stmask_amd.dist layout)."""
    _dev(mask, score, tracked, offsets, box, cls, mask_coeff)
    n = box.shape[0]
    out = torch.empty(B, top_k, cols, device=offsets.device, dtype=torch.float32)
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=offsets.device)
    hw = mask[0].numel() if n else 1
    check(_lib.lib().stm_pack_tracked_f32(_p(_f32c(mask)) if n else c_p(0), _p(score), _p(tracked), _p(offsets), _p(box), _p(cls), _p(mask_coeff),
                                          c_i(n), c_i(hw), c_i(B), c_i(top_k), c_i(cols), c_i(mask_coeff.shape[1] if n else cols - 8), c_i(max_age),
                                          c_f(score_thr), _p(keep), _p(out), _stream()), "stm_pack_tracked_f32")
    return out


def pack_tracked_bits(bits, score, tracked, offsets, box, cls, mask_coeff, B, top_k, cols, max_age=10, score_thr=0.05):
    """pack_tracked with the keep rule's pixel count taken from the masks' bit words [n, words] (lincomb_sigmoid_crop_bits)."""
    _dev(bits, score, tracked, offsets, box, cls, mask_coeff)
    n = box.shape[0]
    out = torch.empty(B, top_k, cols, device=offsets.device, dtype=torch.float32)
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=offsets.device)
    if n and (bits.dtype != torch.int64 or bits.shape[0] != n or not bits.is_contiguous()):
        raise StmError("pack_tracked_bits: bits must be contiguous int64 words [n, words]")
    check(_lib.lib().stm_pack_tracked_bits_f32(_p(bits) if n else c_p(0), c_i(bits.shape[1] if n else 1), _p(score), _p(tracked), _p(offsets), _p(box),
                                               _p(cls), _p(mask_coeff), c_i(n), c_i(B), c_i(top_k), c_i(cols),
                                               c_i(mask_coeff.shape[1] if n else cols - 8), c_i(max_age), c_f(score_thr), _p(keep), _p(out), _stream()),
          "stm_pack_tracked_bits_f32")
    return out


def resize_bilinear_planes(x_nhwc, size, fmt=0):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) of an fp32 NHWC tensor [B,H,W,C], returned as planes
    [P, C/32, B*Ho*Wo, 32] (split_planes' format) without the fp32 intermediate."""
    _dev(x_nhwc)
    x = _f32c(x_nhwc)
    B, H, W, C = x.shape
    Ho, Wo = size
    if C % 32:
        raise StmError(f"resize_bilinear_planes: channel count {C} is not a multiple of 32")
    planes = _empty_planes(fmt, C // 32, B * Ho * Wo, x.device)
    check(_lib.lib().stm_resize_bilinear_planes_f32(_p(x), _p(planes), c_i(B), c_i(H), c_i(W), c_i(C), c_i(Ho), c_i(Wo), c_i(fmt), _stream()),
          "stm_resize_bilinear_planes_f32")
    return planes


def bias_relu_maxpool_planes(x_nhwc, bias, fmt=0):
    """relu(max_pool2d(x, 3, 2, 1) + bias) of an fp32 NHWC tensor [B,H,W,C] (the ResNet stem's convolution output), returned as
    planes [P, C/32, B*Ho*Wo, 32] plus (Ho, Wo)."""
    _dev(x_nhwc, bias)
    x = _f32c(x_nhwc)
    B, H, W, C = x.shape
    if C % 32:
        raise StmError(f"bias_relu_maxpool_planes: channel count {C} is not a multiple of 32")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    planes = _empty_planes(fmt, C // 32, B * Ho * Wo, x.device)
    check(_lib.lib().stm_bias_relu_maxpool_planes_f32(_p(x), _p(_f32c(bias)) if bias is not None else c_p(0), _p(planes), c_i(B), c_i(H), c_i(W),
                                                      c_i(C), c_i(fmt), _stream()), "stm_bias_relu_maxpool_planes_f32")
    return planes, (Ho, Wo)


def roi_align_planes(t2s_prev_nhwc, t2s_nhwc, corr, rois, output_size=7, fmt=0, corr_nhwc=0):
    """relu(cat([corr, T2S_prev, T2S], 1)) -> roi_align(output_size, aligned, adaptive grid) as planes
    [P, Cpad/32, n*ph*pw, 32] with channel order [T2S_prev | T2S | corr | zero padding] (stm_roi_align_planes_f32).  corr is
    [B, Cc, H, W], or with corr_nhwc = Cc the channels-last [B, H, W, ld] of corr_patch_nhwc."""
    _dev(t2s_prev_nhwc, t2s_nhwc, corr, rois)
    a, b, c, rois = _f32c(t2s_prev_nhwc), _f32c(t2s_nhwc), _f32c(corr), _f32c(rois)
    B, H, W, C1 = a.shape
    ok = (tuple(c.shape[:3]) == (B, H, W) and c.shape[3] >= corr_nhwc) if corr_nhwc else (c.shape[0] == B and tuple(c.shape[2:]) == (H, W))
    if tuple(b.shape) != (B, H, W, C1) or not ok:
        raise StmError(f"roi_align_planes: shapes {tuple(a.shape)}, {tuple(b.shape)}, {tuple(c.shape)} do not match")
    Cc, n = (corr_nhwc or c.shape[1]), rois.shape[0]
    ph, pw = _pair(output_size)
    cpad = -(-(2 * C1 + Cc) // 32) * 32
    planes = _empty_planes(fmt, cpad // 32, n * ph * pw, a.device)
    if n:
        check(_lib.lib().stm_roi_align_planes_nhwc_f32(_p(a), _p(b), _p(c), c_i(c.shape[3] if corr_nhwc else 0), _p(rois), _p(planes), c_i(B),
                                                       c_i(H), c_i(W), c_i(C1), c_i(Cc), c_i(n), c_i(ph), c_i(pw), c_i(fmt), _stream()),
              "stm_roi_align_planes_f32")
    return planes


def stem_rows_planes(x_nhwc, kw, sw, pw, fmt=0):
    """fp32 frame [B,H,W,Cin] (kw*Cin <= 32) -> planes R [P, 1, B*H*Wo, 32]: the kw*Cin contiguous values one kernel row reads for
    each output column (stm_stem_rows_planes_f32); returns (planes, Wo)."""
    _dev(x_nhwc)
    x = _f32c(x_nhwc)
    B, H, W, Cin = x.shape
    Wo = (W + 2 * pw - kw) // sw + 1
    planes = _empty_planes(fmt, 1, B * H * Wo, x.device)
    check(_lib.lib().stm_stem_rows_planes_f32(_p(x), _p(planes), c_i(B), c_i(H), c_i(W), c_i(Cin), c_i(kw), c_i(sw), c_i(pw), c_i(fmt),
                                              _stream()), "stm_stem_rows_planes_f32")
    return planes, Wo


def stem_pack_weights(weight, fmt):
    """conv1 weights [64, 3, 7, 7] -> the fragment image stm_stem_fused_f32 keeps in registers; returns (packed, 1 / wscale)."""
    _dev(weight)
    weight = _f32c(weight)
    O = weight.shape[0]
    nbytes = _lib.lib().stm_stem_packed_weight_bytes(c_i(O), c_i(fmt))
    if nbytes == 0 or tuple(weight.shape[1:]) != (3, 7, 7):
        raise StmError(f"stem_pack_weights: unsupported stem {tuple(weight.shape)} / format {fmt}")
    import math
    wmax = float(weight.abs().max())
    wscale = 2.0 ** (10 - math.floor(math.log2(wmax))) if wmax > 0 else 1.0
    packed = torch.empty(nbytes, device=weight.device, dtype=torch.uint8)
    check(_lib.lib().stm_stem_pack_weights_f32(_p(weight), _p(packed), c_i(O), c_i(fmt), c_f(wscale), _stream()), "stm_stem_pack_weights_f32")
    return packed, 1.0 / wscale


def stem_fused(x_nhwc, packed, out_scale, bias, fmt, out_fmt=None):
    """conv1 (7x7 / 2 / 3, BN folded) + ReLU + MaxPool2d(3, 2, 1) in one kernel: fp32 frame [B,H,W,3] -> (planes [P, 2, B*Hp*Wp, 32],
    (Hp, Wp)).  stm_stem_fused_f32."""
    _dev(x_nhwc, packed, bias)
    x = _f32c(x_nhwc)
    B, H, W, Cin = x.shape
    if Cin != 3:
        raise StmError(f"stem_fused: 3-channel frames only, got {Cin}")
    out_fmt = fmt if out_fmt is None else out_fmt
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    planes = _empty_planes(out_fmt, 2, B * Hp * Wp, x.device)
    check(_lib.lib().stm_stem_fused_f32(_p(x), _p(packed), _p(_f32c(bias)) if bias is not None else c_p(0), _p(planes), c_i(B), c_i(H), c_i(W),
                                        c_i(64), c_i(fmt), c_i(out_fmt), c_f(out_scale), _stream()), "stm_stem_fused_f32")
    return planes, (Hp, Wp)


def planes_to_f32(planes):
    """[P, S, N, 32] planes -> fp32 [N, 32*S]."""
    v = planes[0].float()
    if planes.dtype == torch.float16:
        if planes.shape[0] == 2:
            v = v + planes[1].float() / F16_LOW_SCALE
    else:
        for p in range(1, planes.shape[0]):
            v = v + planes[p].float()
    return v.permute(1, 0, 2).reshape(v.shape[1], -1)


def conv2d_planar(xp, packed, weight_shape, hw, bias=None, residual=None, stride=1, padding=0, relu=False, planes=3,
                  out="planes", tile_n=128, fmt=0, out_scale=1.0):
    """The same convolution on the planar activation format: xp [P, C/32, B*H*W, 32] (split_planes / a previous layer's
    output), hw = (B, H, W).  `residual` may be fp32 [B*Ho*Wo, O] or planes [P, O/32, B*Ho*Wo, 32].
    out: "planes" | "f32" | "both"; fp32 result [B*Ho*Wo, O].  fmt 1: fp16 planes, `packed` / `out_scale` from
    conv_pack_weights(..., fmt=1)."""
    _dev(xp, packed, bias, residual)
    P, dt = plane_layout(fmt)
    if fmt >= 1:
        planes = P
    if xp.dtype != dt or xp.dim() != 4 or xp.shape[0] != P or xp.shape[3] != 32 or not xp.is_contiguous():
        raise StmError(f"conv2d_planar: expected contiguous {dt} planes [{P},C/32,N,32], got {xp.dtype} {tuple(xp.shape)}")
    O, C, kh, kw = weight_shape
    B, H, W = hw
    if xp.shape[1] * 32 != C or xp.shape[2] != B * H * W:
        raise StmError(f"conv2d_planar: planes {tuple(xp.shape)} do not match C={C}, B*H*W={B * H * W}")
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    Ho, Wo = conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, 1, 1)
    M = B * Ho * Wo
    y32 = torch.empty(M, O, device=xp.device, dtype=torch.float32) if out in ("f32", "both") else None
    ypl = torch.empty(P, -(-O // 32), M, 32, device=xp.device, dtype=dt) if out in ("planes", "both") else None
    r32 = rpl = None
    if residual is not None:
        if residual.dtype == dt:
            if tuple(residual.shape) != (P, -(-O // 32), M, 32) or not residual.is_contiguous():
                raise StmError(f"conv2d_planar: residual planes {tuple(residual.shape)} != {(P, -(-O // 32), M, 32)}")
            rpl = residual
        else:
            r32 = _f32c(residual)
            if r32.numel() != M * O:
                raise StmError(f"conv2d_planar: residual has {r32.numel()} elements, output {M * O}")
    g = _lib.ConvGeom(B, H, W, C, Ho, Wo, O, kh, kw, sh, sw, ph, pw, 0, 0, 0, planes)
    g.tile_n, g.fmt, g.out_scale = tile_n, fmt, out_scale
    check(_lib.lib().stm_conv2d_planar_f32(_p(xp), _p(packed), _p(_f32c(bias) if bias is not None else None), _p(r32), _p(rpl),
                                           _p(y32), _p(ypl), ctypes.byref(g), c_i(1 if relu else 0), _stream()),
          "stm_conv2d_planar_f32")
    return (y32, ypl) if out == "both" else (y32 if out == "f32" else ypl)


def preprocess_frames(img_u8, size=(640, 360), divisor=32, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375),
                      mode=1):
    """eval.py:703-717 on the device: uint8 [n,H0,W0,3] -> fp32 [n,3,Hp,Wp]; size = (w, h) as mmcv.imresize takes it."""
    _dev(img_u8)
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 4 or img_u8.shape[-1] != 3:
        raise StmError(f"preprocess_frames: expected uint8 [n,H,W,3], got {img_u8.dtype} {tuple(img_u8.shape)}")
    img = img_u8.contiguous()
    n, H0, W0, _ = img.shape
    w, h = size
    Hp, Wp = -(-h // divisor) * divisor, -(-w // divisor) * divisor
    out = torch.empty(n, 3, Hp, Wp, device=img.device, dtype=torch.float32)
    m3, s3 = (ctypes.c_double * 3)(*mean), (ctypes.c_double * 3)(*std)
    check(_lib.lib().stm_preprocess_u8_f32(_p(img), _p(out), c_i(n), c_i(H0), c_i(W0), c_i(h), c_i(w), c_i(Hp), c_i(Wp), m3, s3,
                                           c_i(mode), _stream()), "stm_preprocess_u8_f32")
    return out


def head_assemble(small, trk, B, sizes, n_cls, mask_dim, embed_dim, group_pad):
    """prediction_head_FC.py:168-195 for the planar head: small / trk = per-kernel-shape lists of [pixels, 3*group_pad] /
    [pixels, embed] fp32 matrices over the concatenated levels `sizes` = [(H, W), ...] with B images each.
    -> conf [B,N,n_cls], loc [B,N,4], mask [B,N,mask_dim], track [B,N,embed] (normalised), centerness [B,N,1] (tanh)."""
    _dev(*small, *trk)
    K = len(small)
    L = _lib.HeadLayout()
    L.B, L.K, L.n_levels, L.n_cls, L.mask_dim, L.embed_dim, L.group_pad = B, K, len(sizes), n_cls, mask_dim, embed_dim, group_pad
    L.small_ld, L.trk_ld = small[0].shape[-1], trk[0].shape[-1]
    start = 0
    for l, (h, w) in enumerate(sizes):
        L.lvl_start[l], L.lvl_hw[l] = start, h * w
        start += B * h * w
    N = K * sum(h * w for h, w in sizes)
    for t in list(small) + list(trk):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape[0] != start:
            raise StmError("head_assemble: inputs must be contiguous fp32 matrices over all level pixels")
    dev = small[0].device
    conf = torch.empty(B, N, n_cls, device=dev)
    loc = torch.empty(B, N, 4, device=dev)
    mask = torch.empty(B, N, mask_dim, device=dev)
    track = torch.empty(B, N, embed_dim, device=dev)
    cen = torch.empty(B, N, 1, device=dev)
    sp = (ctypes.c_void_p * 4)(*([t.data_ptr() for t in small] + [0] * (4 - K)))
    tp = (ctypes.c_void_p * 4)(*([t.data_ptr() for t in trk] + [0] * (4 - K)))
    check(_lib.lib().stm_head_assemble_f32(sp, tp, ctypes.byref(L), _p(conf), _p(loc), _p(mask), _p(track), _p(cen), _stream()),
          "stm_head_assemble_f32")
    return conf, loc, mask, track, cen


def deform_conv_fused_supported(C, O, kernel_size, has_mask, fmt, deformable_groups=1):
    """True when stm_deform_conv_fused_planar_f32 takes this layer (one deformable group, <= 15 taps, 9 with mask, C % 64 == 0,
    O % 128 == 0, an fp16 plane format)."""
    kh, kw = _pair(kernel_size)
    g = DeformGeom(1, C, 8, 8, kh, kw, 1, 1, kh // 2, kw // 2, 1, 1, deformable_groups, 8, 8)
    return bool(_lib.lib().stm_deform_conv_fused_planar_supported(ctypes.byref(g), c_i(O), c_i(1 if has_mask else 0), c_i(fmt)))


def deform_conv_fused_tiles(B, Ho, Wo, O):
    """Workgroups stm_deform_conv_fused_planar_f32 launches for B images of Ho x Wo output pixels and O channels: tiles of 64 pixels x 256 channels
    where O is a multiple of 256 (STM_DCN_FUSED_WIDE, default on), 128 x 128 otherwise; patches chosen as csrc/dcn_fused.hip pick_patch does (fewest
    wasted tile pixels, then the squarest) -- the callers' small-grid rule."""
    wide = O % 256 == 0 and os.environ.get("STM_DCN_FUSED_WIDE", "1") != "0"
    tp, bn = (64, 256) if wide else (128, 128)
    best, bt, bper = -1.0, 1, 1 << 30
    for tw in range(4, tp + 1):
        th = tp // tw
        if th < 1:
            break
        thc, twc = min(th, Ho), min(tw, Wo)
        tiles = -(-Ho // thc) * -(-Wo // twc)
        eff = Ho * Wo / (tiles * float(tp))
        per = thc + twc
        if eff > best + 1e-9 or (eff > best - 1e-9 and per < bper):
            best, bt, bper = eff, tiles, per
    return B * bt * (O // bn)


def deform_conv_fused_planar(x_pix, B, H, W, C, om, packed, out_scale, bias, O, kernel_size=3, stride=1, padding=1, dilation=1, has_mask=True,
                             relu=False, fmt=1, out_fmt=None, out=None, out_off=0):
    """The whole deformable convolution of the planar graph as one kernel (csrc/dcn_fused.hip: sampler -> plane split -> MFMA product,
    no column buffer): dcn_v2.DCN (backbone.py:20-26,45; has_mask, bias, ReLU) or mmcv DeformConv2d as FeatureAlign uses it
    (Featurealign.py:27-31,72).  x_pix fp32 [B*H*W, ld >= C] pixel-major (unit channel stride), om fp32 [B*Ho*Wo, >= 2K (+K)] raw offsets
    (and mask logits), packed = conv_pack_weights(weight [O, C, kh, kw], tile_n=128, fmt=fmt)[0] with out_scale its second value.
    Returns / fills planes [P, O/32, N, 32] in out_fmt at pixels [out_off, out_off + B*Ho*Wo)."""
    _dev(x_pix, om, packed)
    if x_pix.dtype != torch.float32 or x_pix.dim() != 2 or x_pix.stride(1) != 1 or x_pix.shape[0] != B * H * W or x_pix.shape[1] != C:
        raise StmError(f"deform_conv_fused_planar: x must be fp32 [B*H*W, C] with unit channel stride, got {tuple(x_pix.shape)} {x_pix.stride()}")
    om = _f32c(om)
    kh, kw = _pair(kernel_size)
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ho, Wo = conv_out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw)
    M, K = B * Ho * Wo, kh * kw
    if om.dim() != 2 or om.shape[0] != M or om.shape[1] < (3 if has_mask else 2) * K:
        raise StmError(f"deform_conv_fused_planar: offsets {tuple(om.shape)} do not match {M} output pixels x {(3 if has_mask else 2) * K}")
    out_fmt = fmt if out_fmt is None else out_fmt
    if out is None:
        out = _empty_planes(out_fmt, O // 32, M, x_pix.device)
        out_off = 0
    elif out.dim() != 4 or out.shape[1] * 32 != O or not out.is_contiguous() or out.shape[0] < plane_layout(out_fmt)[0]:
        raise StmError(f"deform_conv_fused_planar: planes {tuple(out.shape)} do not hold {O} channels in format {out_fmt}")
    if bias is not None:
        _dev(bias)
        bias = _f32c(bias)
    g = DeformGeom(B, C, H, W, kh, kw, sh, sw, ph, pw, dh, dw, 1, Ho, Wo)
    timing = _fused_dcn_timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().stm_deform_conv_fused_planar_f32(_p(x_pix), c_i(x_pix.stride(0)), _p(om), c_i(om.shape[1]), c_i(1 if has_mask else 0), _p(packed),
                                                      _p(bias), _p(out), c_i(out.shape[2]), c_i(out_off), c_l(0), c_i(O), c_i(1 if relu else 0),
                                                      c_f(out_scale), ctypes.byref(g), c_i(fmt), c_i(out_fmt), _stream()),
          "stm_deform_conv_fused_planar_f32")
    if timing is not None:
        e1.record()
        # SURVEY.md section 8(d), fused form: input once, offsets (+ mask) per output pixel, output planes, weights (as packed planes); no columns
        npl, npo = plane_layout(fmt)[0], plane_layout(out_fmt)[0]
        nbytes = 4 * B * C * H * W + 4 * (3 if has_mask else 2) * K * M + 2 * npo * O * M + 2 * npl * O * C * K
        timing.append((e0, e1, float(nbytes), 2.0 * M * O * C * K, {1: 3, 2: 1}[fmt]))
    return out


def dcn_sample_planar(x_nhwc, om, stride=1, padding=1, dilation=1, fmt=0):
    """Deformable 3x3 sampling for the planar graph: x fp32 [B,H,W,C], om fp32 [B*Ho*Wo, >=27] (raw conv_offset_mask output,
    pixel-major) -> bf16 planes [3, 9C/32, B*Ho*Wo, 32] with K index = tap*C + channel."""
    _dev(x_nhwc, om)
    x = _f32c(x_nhwc)
    om = _f32c(om)
    B, H, W, C = x.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Ho, Wo = conv_out_hw(H, W, 3, 3, sh, sw, ph, pw, dh, dw)
    M = B * Ho * Wo
    if om.shape[0] != M or om.shape[1] < 27:
        raise StmError(f"dcn_sample_planar: offset/mask matrix {tuple(om.shape)} does not match {M} output pixels x 27")
    g = DeformGeom(B, C, H, W, 3, 3, sh, sw, ph, pw, dh, dw, 1, Ho, Wo)
    out = _empty_planes(fmt, 9 * C // 32, M, x.device)
    timing = _im2col_timing
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().stm_dcn_sample_planar_fmt_f32(_p(x), _p(om), c_i(om.shape[1]), _p(out), c_i(M), c_l(0), ctypes.byref(g),
                                                   c_i(fmt), _stream()), "stm_dcn_sample_planar_fmt_f32")
    if timing is not None:
        e1.record()
        # algorithmic bytes: input once, 27 offset/mask values per output pixel, columns as planes (6 or 4 B / element)
        timing.append((e0, e1, 4 * B * C * H * W + 4 * 27 * M + 2 * plane_layout(fmt)[0] * 9 * C * M))
    return out
