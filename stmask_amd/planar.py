"""Planar inference graph: the ResNet bottlenecks, FPN laterals / prediction / downsample layers, the proto-net, the shared
prediction head and TemporalNet run on stm_conv2d_planar_f32 (include/stmask_hip.h) -- activations stay split into planes
between layers (two fp16 planes by default, three bf16 planes on request: set_format / DESIGN.md section 1), the
five FPN levels of the shared head go through every layer in ONE launch (pixel axis = concatenated levels), the four
branch towers are one Cout=1024 layer followed by one grouped layer, and the per-kernel-shape output layers are one
grouped launch each.  Values are those of the reference's fp32 convolutions to fp32 rounding (tests/test_gpu_conv.py,
tests/test_gpu_model.py); what changes is the schedule:

    reference (prediction_head_FC.py:146-195, per level, 5 levels): 1 + 8 + 15 convolutions, each + bias + ReLU kernels
    here: 1 + 1 + 1 + 3 launches for all levels together.

Built by fuse.optimize_for_inference(net, planar=True) from the (BN-folded) modules; parameters are read once and packed
(stm_conv_pack_weights_fmt_f32).  The FCB class branch (use_dcn_class) runs FeatureAlign's deformable convolution as
planar sampler + planar 1x1 convolution over all levels (stm_deform_sample_planar_f32) and its trailing conv planar; FCB
on the track / mask branches keeps the module path for the head (FPN and proto-net still run planar).
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import StmError, c_i, check


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# Plane format of the graphs built next (fuse.optimize_for_inference sets it): 0 = three bf16 planes, six MFMA products per
# fp32 product, any fp32 range; 1 = two fp16 planes, three products -- half the matrix work at the same fp32-level error,
# activations limited to fp16's range (|x| < 65504; beyond it the outputs are non-finite and the pipeline raises);
# 2 = ONE fp16 plane, one product: genuine fp16 convolutions with fp32 accumulation (~1e-3 relative; BASELINE config 5).
FMT = 0
# Format of the ResNet backbone's convolutions when it differs from FMT (fuse: planes="fp16x1" -> backbone 2, the rest 1)
BACKBONE_FMT = None

# Graph-construction choices.  Plain module constants since round 6 (environment switches STM_* in rounds 2-5; every default has held since it was
# measured -- DESIGN.md section 9 keeps the A/B figures): tests that want the other form of a layer set the attribute.
TILE64_MAX_SLABS = 8      # K-slab limit of the short-K rule of the 128 x 64 tiles
FCB_PLANAR = True         # FCB class branch on the planar kernels
STEM_PLANAR = True        # stem on the planar kernels
C3DS_FUSED = True         # conv3 + projection shortcut of a stage's first block as one two-source product
STEM_FUSED = True         # conv1 + ReLU + max-pool as one kernel (csrc/stem_fused.hip)
CONV_KXR = True           # narrow stride-1 layers on the kx-reuse kernel (csrc/conv_kxr.hip)
TN_BORDER = True          # TemporalNet's 3x3 layers as nine window launches without the zero taps of the RoI borders
TN_POOL = True            # ... and its AvgPool2d in conv3's epilogue, fc + fc_coeff as one launch (needs TN_BORDER)
CONV_CHAIN = True         # layer1's bottlenecks on csrc/conv_chain.hip
# Environment switches that remain (read once at import, never per call): the fused deformable convolution and the chain kernel's mode
# the deformable 3x3 layers (and the FCB class branch) as ONE kernel -- sampler -> plane split -> MFMA product, no column buffer (csrc/dcn_fused.hip) --
# from DCN_FUSED_MIN_TILES workgroups on (below that the grid leaves CUs idle and the sampler + split-K product pair is faster); 0 = the pair everywhere
DCN_FUSED = os.environ.get("STM_DCN_FUSED", "1") != "0"
DCN_FUSED_MIN_TILES = int(os.environ.get("STM_DCN_FUSED_MIN_TILES", "200"))
# which layers: 0 = every layer whose grid is large enough; 1 = only where the fused kernel wins in isolation (profiles/r05_dcn_fused_forms.txt section 0): one
# 128-channel tile per pixel patch, or two at stride 2 -- with more channel tiles every tile samples the patch again
DCN_FUSED_RULE = 0   # (in the step both rules measure the same within 0.1 ms on R50 and R101: profiles/r05_dcn_fused_forms.txt)
# ... the FCB class branch (FeatureAlign's DeformConv2d, 256 -> 256 channels, 9 / 15 / 15 taps over five levels) on the same kernel.  ON by default since the
# kernel has 64-pixel x 256-channel tiles (every pixel sampled once): R50 FCB-ada at 32 clips 36.6 -> 36.1 ms per step, R101 FCB-ali 34.5 -> 34.0; on the
# 128 x 128 tiles (two channel tiles per patch, every pixel sampled twice) it lost, 764 vs 918 frames/s (profiles/r05_dcn_fused_forms.txt).  STM_FCB_FUSED=0: the pair
FCB_FUSED = os.environ.get("STM_FCB_FUSED", "1") != "0"
# Independent parts of the trunk on a second stream while a HIP graph is being captured (PlanarGraph.run): 0 off (default), 1 proto-net beside the
# shared head, 2 also the P5 -> P6 -> P7 convolutions beside the finer FPN levels; only batches of at most BRANCH_MAX_IMAGES frames.  Bit-equal, and
# SLOWER at every batch size it was meant for (profiles/r04_trunk_branches_ab.txt: 1 clip 478 -> 426-442 frames/s, 2 clips 728 -> 640-694, 4 clips
# 1034 -> 960-990, 8 clips 1284 -> 1206-1223): a fork / join pair in a replayed HIP graph costs more than the small grids it lets overlap.
TRUNK_BRANCHES = 0
BRANCH_MAX_IMAGES = 16
# bit 0 = the projection form (a stage's first block); bit 1 = the kernel also computes the NEXT block's conv1 (z); bit 2 (diagnostics) = z is
# computed but not used.  Round 3 shipped 1 because the z-producing instantiation (conv_chain_kernel<true, .>) gave wrong y / z beside a
# second process on the GPU.  Round 4 found why (csrc/conv_chain.hip, store16: a 16-byte buffer store with an SGPR soffset reads its data
# after the next VALU write; only those instantiations had the two back to back) and fixed it: 0 differing outputs in 38 400 launches
# beside the second process where round 3's code gave 36 in 12 800 (profiles/r04_ring_stress_chain_variants.txt).
CHAIN_MODE = int(os.environ.get("STM_CHAIN_MODE", "3"))  # layer1's identity-shortcut blocks: conv2 -> conv3 + shortcut -> next conv1 as one kernel (csrc/conv_chain.hip)
CHAIN_MAX_PIXELS = (1 << 21) - 1     # stm_bottleneck_chain_f32 addresses a 256-channel plane pair with 31-bit byte offsets (2 * 8 * M * 64 < 2^31)


def set_format(fmt, backbone_fmt=None):
    global FMT, BACKBONE_FMT
    if fmt not in (0, 1, 2) or backbone_fmt not in (None, 0, 1, 2):
        raise StmError("planar format must be 0 (bf16 x 3), 1 (fp16 x 2) or 2 (fp16 x 1)")
    if backbone_fmt is not None and backbone_fmt != fmt and (backbone_fmt, fmt) != (2, 1):
        raise StmError("a backbone format different from the graph's is only built for fp16x1 under fp16x2")
    FMT, BACKBONE_FMT = fmt, backbone_fmt


def _planes_dtype(fmt):
    return ops.plane_layout(fmt)


class PlanarConv:
    """One packed convolution layer.  x / outputs are described by raw (tensor, pixel offset) pairs so that a layer can
    read from and write into slices of larger plane buffers."""

    SPLITK_WS_BYTES = 64 << 20   # scratch for split-K partial sums (small feature maps with long K)

    def __init__(self, weight, bias, stride=1, padding=0, relu=False, groups=1, planes=3, algo_frac=1.0, tile_n=None,
                 group_cout=None, fmt=None, out_fmt=None):
        """algo_frac: share of the packed layer that is the reference's own arithmetic (zero-padded channels excluded);
        only used for the flop count of the live roofline measurement.  tile_n: 64 / 128 forces the output-channel tile,
        None picks per call from the problem size (weights are packed once per tile width used)."""
        self.weight = weight.detach().float().contiguous()
        self.fmt = FMT if fmt is None else fmt
        self.out_fmt = self.fmt if out_fmt is None else out_fmt     # format of the planes this layer writes (fmt 2 layers may write 1)
        self.algo_frac = algo_frac
        self.group_cout = list(group_cout) if group_cout else None   # real channels of zero-padded groups
        self.O, self.C, self.kh, self.kw = self.weight.shape
        (self.sh, self.sw), (self.ph, self.pw) = _pair(stride), _pair(padding)
        self.relu, self.groups, self.planes, self.tile_n = relu, groups, (planes if self.fmt == 0 else ops.plane_layout(self.fmt)[0]), tile_n
        self._packed = {}
        self.out_scale = 1.0
        self.role = "trunk"          # "temporal" for TemporalNet's layers: bench.py reports the trunk-only roofline beside the overall one
        # few output channels per group, stride 1, kw >= 3: the kx-reuse kernel (stm_conv2d_planar_kxr_f32) -- these layers run at the
        # L2 -> LDS staging rate on the 128 x 64 tiles (head output layers, DCN offset convolutions, layer1's 3x3)
        self.kxr = CONV_KXR and self.out_fmt == self.fmt and ops.conv_kxr_supported(self.O, self.C, self.kh, self.kw, (self.sh, self.sw), (self.ph, self.pw),
                                                                                     self.groups, self.group_cout, self.fmt, max_tiles=3)
        self.kxr_min_pixels = None   # None: the measured thresholds of __call__; tests set 0 to force the kernel on small inputs
        self.bias = bias.detach().float().contiguous() if bias is not None else None

    def packed(self, tile_n):
        if tile_n not in self._packed:
            if self.fmt >= 1:
                ops.planar_range_flag()     # the producers of fp16 planes report |x| > 65504 through it
                self._packed[tile_n], self.out_scale = ops.conv_pack_weights(self.weight, tile_n=tile_n, fmt=self.fmt)
            else:
                self._packed[tile_n] = ops.conv_pack_weights(self.weight, self.planes, tile_n)
        return self._packed[tile_n]

    def pick_tile(self, M):
        """Measured on the R50 layer shapes at batch 8 (scripts/bench_conv.py): 128-channel tiles (256 or 128 pixels, one
        workgroup per CU) win once the grid has a few hundred of them; 128 x 64 tiles (two workgroups per CU) win for
        narrow layers and for grids that would leave CUs idle."""
        if self.tile_n is not None:
            return self.tile_n
        cg = self.O // self.groups
        if self.groups > 1 and cg % 128 != 0:
            return 64
        if cg <= 64 or (cg % 128 != 0 and cg % 128 <= 64 and cg < 256):
            return 64
        tiles128 = -(-M // 128) * -(-self.O // 128)
        slabs = self.C * self.kh * self.kw // 32
        # (long K on a grid of about one 128 x 128 workgroup per CU -- layer4 at 32 clips -- is better off on the wide tiles:
        # 2048 -> 512 at M = 7 680: 58.4 -> 53.5 us, the 4608 -> 512 DCN product 118.1 -> 104.9; scripts/sweep_small_m.py)
        # (round 4, scripts/sweep_tile_small.py: the exception holds up to ONE workgroup per CU only -- the head towers at 1 clip, 320 wide tiles:
        # 98 -> 81 us on the 64-wide ones; the `up` layer at 4 clips: 96 -> 77 us)
        if tiles128 < 400 and not (192 <= tiles128 <= 256 and slabs >= 64):
            return 64
        # short K, wide output (the bottlenecks' expanding 1x1 convs with their residual): HBM-bound, and three resident
        # 128 x 64 workgroups per CU (48 KB each) keep more loads and stores in flight than one 256 x 128 workgroup:
        # 393 -> 302 us (64 -> 256 channels at 96x160, batch 32), 213 -> 175 us, 123 -> 112 us (scripts/ab_shortk.py)
        if slabs <= TILE64_MAX_SLABS:
            return 64
        return 128

    def __call__(self, xp, shape, out="planes", x_off=0, out_planes=None, out_f32=None, out_off=0, residual=None, x_ch_off=0,
                 out_ch_off=0, x2=None, window=None):
        """xp: [P, S, N, 32] planes in this layer's format (channel-slab major; 2 x fp16 or 3 x bf16).  shape: ("img", B, H, W) -> pixels [x_off, x_off + B*H*W)
        of xp are one image batch; ("levels", B, [(H, W), ...]) -> all of xp, concatenated levels.  The layer reads
        groups*C channels starting at channel x_ch_off.  out: "planes" | "f32" | "both" allocates dense outputs
        ([3, O/32, M, 32] / [M, O]) unless out_planes / out_f32 are given, then pixels [out_off, ...) are written."""
        NP, dt = _planes_dtype(self.fmt)
        if xp.dtype != dt or xp.dim() != 4 or xp.shape[0] < NP or xp.shape[3] != 32 or not xp.is_contiguous():
            raise StmError(f"PlanarConv: expected contiguous {dt} planes [{NP}, S, N, 32], got {xp.dtype} {tuple(xp.shape)}")
        # (a fp16x1 layer handed a two-plane fp16 tensor reads plane 0 = RN16(x): the one-plane tensor of the same values)
        S, N = xp.shape[1], xp.shape[2]
        g = _lib.ConvGeom()
        g.C, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw = self.C, self.O, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw
        g.planes, g.groups = self.planes, self.groups
        if self.group_cout:
            for i, c in enumerate(self.group_cout):
                g.group_cout[i] = c
        # x2 = (planes2, H2, W2, stride): a two-source 1x1 layer (stm_conv2d_planar_dual_f32) -- xp holds the first C - C2 input channels at
        # the output's resolution, planes2 the other C2 channels as B images of H2 x W2 read at `stride`
        c_first = self.C - (x2[0].shape[1] * 32 if x2 is not None else 0)
        if S * 32 < x_ch_off + self.groups * c_first or x_ch_off % 32:
            raise StmError(f"PlanarConv: input has {S * 32} channels, layer reads {self.groups} x {c_first} from channel {x_ch_off}")
        if shape[0] == "levels":
            _, B, sizes = shape
            g.n_levels = len(sizes)
            start = 0
            for l, (h, w) in enumerate(sizes):
                g.lvl_start[l], g.lvl_h[l], g.lvl_w[l] = start, h, w
                start += B * h * w
            g.lvl_start[len(sizes)] = start
            M = start
            if x_off or N != start:
                raise StmError("PlanarConv: a multi-level launch covers the whole plane buffer")
        else:
            _, B, H, W = shape
            Ho, Wo = ops.conv_out_hw(H, W, self.kh, self.kw, self.sh, self.sw, self.ph, self.pw, 1, 1)
            out_rows = None
            if window is not None:
                # window launch (stm_conv_geom.win_*): (y0, x0, ho, wo, ph, pw, full_h, full_w) -- only the ho x wo outputs from (y0, x0) of
                # every full_h x full_w output image, written in place into the full output tensors; ph / pw may be negative
                y0, x0, Ho, Wo, g.ph, g.pw, fh, fw = window
                g.win_h, g.win_w, g.win_y0, g.win_x0 = fh, fw, y0, x0
                out_rows = B * fh * fw
                if residual is not None or x2 is not None:
                    raise StmError("PlanarConv: window launches take no residual / second source")
            g.B, g.H, g.W, g.Ho, g.Wo = B, H, W, Ho, Wo
            M = B * Ho * Wo
            if x_off + B * H * W > N:
                raise StmError("PlanarConv: input slice runs past the plane buffer")
        g.x_np, g.x_plane_stride = N, S * N * 32
        # measured against the 128 x 64 tiles (scripts/bench_kxr.py, 1 / 4 / 8 / 32 clips): the head's grouped output layers win from
        # ~20 000 pixels (x1.3-1.8), single-group layers of up to 48 channels from ~30 000 (x1.1-1.4); below that its 256-pixel tiles
        # leave CUs idle, and four channel tiles (layer1's 64 -> 64) stay on the general kernel (x0.65)
        use_kxr = (self.kxr and residual is None and window is None and (shape[0] == "levels" or (Ho, Wo) == (H, W))
                   and M >= (self.kxr_min_pixels if self.kxr_min_pixels is not None else (20000 if self.groups > 1 else 30000)))
        g.tile_n = 0 if use_kxr else self.pick_tile(M)
        dev = xp.device
        NPo, dto = _planes_dtype(self.out_fmt)
        if window is not None and ((out in ("planes", "both") and out_planes is None) or (out in ("f32", "both") and out_f32 is None)):
            raise StmError("PlanarConv: a window launch writes into caller-provided full-size outputs")
        if out in ("planes", "both") and out_planes is None:
            out_planes, out_off_p = torch.empty(NPo, -(-self.O // 32), M, 32, device=dev, dtype=dto), 0
        else:
            out_off_p = out_off
        if out in ("f32", "both") and out_f32 is None:
            out_f32, out_off_f = torch.empty(M, self.O, device=dev, dtype=torch.float32), 0
        else:
            out_off_f = out_off
        if out == "planes":
            out_f32 = None
        if out == "f32":
            out_planes = None
        p_pl = p_f32 = 0
        if out_planes is not None:
            g.out_np, g.out_plane_stride = out_planes.shape[2], out_planes.shape[1] * out_planes.shape[2] * 32
            p_pl = out_planes.data_ptr() + out_off_p * 64
        if out_f32 is not None:
            g.out_ld = out_f32.shape[-1]
            p_f32 = out_f32.data_ptr() + (out_off_f * g.out_ld + out_ch_off) * 4   # fp32 output may start at a column
        r32 = rpl = 0
        if residual is not None:
            if residual.dtype == dt:
                if residual.shape[0] < NP:
                    raise StmError(f"PlanarConv: residual has {residual.shape[0]} plane(s), format {self.fmt} reads {NP}")
                g.res_np, g.res_plane_stride = residual.shape[2], residual.shape[1] * residual.shape[2] * 32
                rpl = residual.data_ptr()
            else:
                g.res_ld = residual.shape[-1]
                r32 = residual.data_ptr()
        timing = ops._conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        x_ptr = xp.data_ptr() + ((x_ch_off // 32) * N + x_off) * 64
        if use_kxr:
            g.fmt = self.fmt
            if "kxr" not in self._packed:
                ops.planar_range_flag()
                self._packed["kxr"] = ops.conv_pack_weights_kxr(self.weight, g)
            packed, g.out_scale = self._packed["kxr"]
            rc = _lib.lib().stm_conv2d_planar_kxr_f32(ctypes.c_void_p(x_ptr), ops._p(packed), ops._p(self.bias), ctypes.c_void_p(p_f32),
                                                      ctypes.c_void_p(p_pl), ctypes.byref(g), c_i(1 if self.relu else 0), ops._stream())
            check(rc, "stm_conv2d_planar_kxr_f32")
            return self._finish(timing, e0 if timing is not None else None, M, shape, g, NP, NPo, dt, out, out_f32, out_planes, residual)
        packed = self.packed(g.tile_n)                     # (sets self.out_scale for the fp16 format)
        g.fmt, g.out_scale = self.fmt, self.out_scale
        g.out_fmt_plus1 = 0 if self.out_fmt == self.fmt else self.out_fmt + 1
        ws = ops._workspace(self.SPLITK_WS_BYTES, dev, "conv_splitk")     # grow-only, shared: split-K partial sums
        if x2 is not None:
            p2, H2, W2, s2 = x2
            if p2.dtype != dt or p2.dim() != 4 or p2.shape[0] < NP or p2.shape[3] != 32 or not p2.is_contiguous() or shape[0] != "img":
                raise StmError("PlanarConv: the second source must be contiguous planes of this layer's format over one image size")
            rc = _lib.lib().stm_conv2d_planar_dual_f32(ctypes.c_void_p(x_ptr), ops._p(p2), c_i(p2.shape[1] * 32), c_i(H2), c_i(W2), c_i(s2),
                                                       ctypes.c_longlong(p2.shape[2]), ctypes.c_longlong(p2.shape[1] * p2.shape[2] * 32),
                                                       ops._p(packed), ops._p(self.bias), ctypes.c_void_p(r32), ctypes.c_void_p(rpl),
                                                       ctypes.c_void_p(p_f32), ctypes.c_void_p(p_pl), ctypes.byref(g), c_i(1 if self.relu else 0),
                                                       ops._p(ws), ctypes.c_size_t(ws.numel()), ops._stream())
            check(rc, "stm_conv2d_planar_dual_f32")
            return self._finish(timing, e0 if timing is not None else None, M, shape, g, NP, NPo, dt, out, out_f32, out_planes, residual)
        rc = _lib.lib().stm_conv2d_planar_ws_f32(ctypes.c_void_p(x_ptr), ops._p(packed),
                                                 ops._p(self.bias), ctypes.c_void_p(r32), ctypes.c_void_p(rpl),
                                                 ctypes.c_void_p(p_f32), ctypes.c_void_p(p_pl), ctypes.byref(g),
                                                 c_i(1 if self.relu else 0), ops._p(ws), ctypes.c_size_t(ws.numel()),
                                                 ops._stream())
        check(rc, "stm_conv2d_planar_f32")
        return self._finish(timing, e0 if timing is not None else None, M, shape, g, NP, NPo, dt, out, out_f32, out_planes, residual)

    def deform(self, x32, B, H, W, om, stride, padding, dilation, has_mask, out=None, out_off=0):
        """This layer as a DEFORMABLE convolution in one kernel (ops.deform_conv_fused_planar, csrc/dcn_fused.hip): x32 fp32 pixel-major
        [B*H*W, C] (a channel slice of a wider tensor is fine), om fp32 [B*Ho*Wo, 2K (+K)]; the weight is this layer's own kh x kw weight
        packed for 128-channel tiles.  Bias and ReLU as configured; planes out."""
        packed = self.packed(128)
        return ops.deform_conv_fused_planar(x32, B, H, W, self.C, om, packed, self.out_scale, self.bias, self.O, (self.kh, self.kw), stride, padding,
                                            dilation, has_mask=has_mask, relu=self.relu, fmt=self.fmt, out_fmt=self.out_fmt, out=out, out_off=out_off)

    def _finish(self, timing, e0, M, shape, g, NP, NPo, dt, out, out_f32, out_planes, residual):
        if timing is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            # algorithmic HBM bytes of the launch: every input / residual / output element and every weight once, in the
            # formats they are stored in (planes 2 B per plane and element, fp32 4 B)
            in_px = M if shape[0] == "levels" else shape[1] * shape[2] * shape[3]
            nbytes = in_px * self.groups * self.C * 2 * NP + self.weight.numel() * 2 * NP
            if out_planes is not None:
                nbytes += M * self.O * 2 * NPo
            if out_f32 is not None:
                nbytes += M * self.O * 4
            if residual is not None:
                nbytes += M * self.O * (2 * NP if residual.dtype == dt else 4)
            # (start, end, algorithmic flops, layer key [tile 0 = the kx-reuse kernel], MFMA products per product of the reference: 6 bf16x3 /
            # 3 fp16x2 / 1 fp16x1, role, algorithmic bytes)
            timing.append((e0, e1, 2.0 * M * self.O * self.C * self.kh * self.kw * self.algo_frac,
                           (M, self.C, self.O, self.kh, self.sh, self.groups, g.tile_n),
                           {0: 6 if self.planes == 3 else 3, 1: 3, 2: 1}[self.fmt], self.role, float(nbytes)))
        if out == "both":
            return out_f32, out_planes
        return out_f32 if out == "f32" else out_planes


def _nhwc(t):
    """[B, C, H, W] (any strides) -> contiguous [B, H, W, C] (a free view when t is channels_last)."""
    return t.permute(0, 2, 3, 1).contiguous()


def _split(t_nhwc, fmt=None):
    """fp32 [B, H, W, C] -> planes [P, C/32, B*H*W, 32] in the current (or given) plane format."""
    return ops.split_planes(t_nhwc, FMT if fmt is None else fmt)


class PlanarGraph:
    GROUP_PAD = 64    # output channels per group of the grouped output layers (41 / 5 / 32 real ones)

    def __init__(self, net):
        cfg = net.cfg
        self.net = net
        self.fmt = FMT
        fpn = net.fpn
        self.n_lat = len(fpn.lat_layers)
        # FPN prediction convs always end in ReLU (FPN.py:96-100); downsample convs do not
        self.fpn_pred = [PlanarConv(m.weight, m.bias, m.stride, m.padding, relu=True) for m in fpn.pred_layers]
        self.fpn_down = [PlanarConv(m.weight, m.bias, m.stride, m.padding, relu=False) for m in fpn.downsample_layers]
        # lateral 1x1 convs (FPN.py:84-93), used when the backbone hands over its outputs as planes (PlanarBackbone): the
        # upsampled coarser level enters as the fp32 residual of the epilogue, the sum leaves as planes for the prediction conv
        self.fpn_lat = [PlanarConv(m.weight, m.bias, 1, 0, relu=False) for m in fpn.lat_layers]
        # proto-net: Conv2d / ReLU / InterpolateModule sequence, then F.relu in STMask.forward_single
        self.proto = []
        mods = [m for m in net.proto_net.children() if not isinstance(m, (torch.nn.ReLU, torch.nn.Identity))]
        for m in mods:
            if isinstance(m, torch.nn.Conv2d):
                self.proto.append(PlanarConv(m.weight, m.bias, m.stride, m.padding, relu=True))
            else:
                self.proto.append(m)   # the bilinear upsample
        self.cor_idx = net.correlation_selected_layer if cfg.temporal_fusion_module else None
        self.timer = None      # a pipeline._StageTimer (STM_PIPE_TIMING=1 diagnosis runs only)
        head = net.prediction_layers[0]
        # FCB on the class branch (FCB-ada / FCB-ali configs) is handled below; FCB on the track / mask branches is not
        # used by any STMask config (config.py:698-701, 793-807) and keeps the module path
        self.head_planar = not (cfg.use_dcn_track or cfg.use_dcn_mask) and cfg.share_prediction_module
        self.fcb = bool(cfg.use_dcn_class)
        if not self.head_planar:
            return
        convs = lambda seq: [m for m in seq.children() if isinstance(m, torch.nn.Conv2d)]
        up = convs(head.upfeature)
        assert len(up) == 1
        self.up = PlanarConv(up[0].weight, up[0].bias, 1, up[0].padding, relu=True)
        towers = [convs(head.conf_extra), convs(head.bbox_extra), convs(head.mask_extra), convs(head.track_extra)]
        assert all(len(t) == 2 for t in towers), "extra_layers (2,2,2,2) is the only head layout on the hot path"
        w1 = torch.cat([t[0].weight for t in towers], 0)
        b1 = torch.cat([t[0].bias for t in towers], 0)
        self.tower1 = PlanarConv(w1, b1, 1, towers[0][0].padding, relu=True)               # 256 -> 4 x 256, shared input
        w2 = torch.cat([t[1].weight for t in towers], 0)
        b2 = torch.cat([t[1].bias for t in towers], 0)
        self.tower2 = PlanarConv(w2, b2, 1, towers[0][1].padding, relu=True, groups=4)     # 4 x (256 -> 256)
        # output layers, per kernel shape k: the conf / centerness+bbox / mask layers read three consecutive 256-channel
        # groups of the tower output -> one grouped launch with 64 output channels per group (41 / 5 / 32 real ones);
        # the track layer (128 channels) reads the fourth group -> its own launch
        self.finals = []
        P = self.GROUP_PAD
        self.dims = (head.num_priors * head.num_classes, head.num_priors * 4, head.num_priors * head.mask_dim,
                     head.num_priors * head.embed_dim)
        for k in range(len(cfg.head_layer_params)):
            mods = [[head.centerness_layer[k], head.bbox_layer[k]], [head.mask_layer[k]]]
            if not self.fcb:
                mods = [[head.conf_layer[k]]] + mods
            ws, bs = [], []
            for grp in mods:
                w = torch.cat([m.weight for m in grp], 0)
                b = torch.cat([m.bias for m in grp], 0)
                assert w.shape[0] <= P
                ws.append(F.pad(w, (0, 0, 0, 0, 0, 0, 0, P - w.shape[0])))
                bs.append(F.pad(b, (0, P - b.shape[0])))
            m0 = mods[0][0]
            real = sum(m.weight.shape[0] for grp in mods for m in grp)
            small = PlanarConv(torch.cat(ws, 0), torch.cat(bs, 0), 1, m0.padding, relu=False, groups=len(mods),
                               algo_frac=real / (len(mods) * float(P)), tile_n=64,
                               group_cout=[sum(m.weight.shape[0] for m in grp) for grp in mods])
            tr = head.track_layer[k]
            entry = [small, PlanarConv(tr.weight, tr.bias, 1, tr.padding, relu=False)]
            if self.fcb:
                # FeatureAlign (Featurealign.py:6-74): offsets from the box regression, DeformConv2d + ReLU on the existing
                # deformable kernels (NCHW fp32), then its trailing conv on the planar kernel over all levels at once
                fa = head.conf_layer[k]
                entry.append(fa)
                # (41 class channels in rows of 48: whole 16-channel tiles for the kx-reuse kernel; the rows land in the 64-column
                # class group of the output buffer)
                n_c = fa.conv.weight.shape[0]
                n_cp = -(-n_c // 16) * 16
                assert n_cp <= P
                entry.append(PlanarConv(F.pad(fa.conv.weight, (0, 0, 0, 0, 0, 0, 0, n_cp - n_c)), F.pad(fa.conv.bias, (0, n_cp - n_c)), 1, fa.conv.padding,
                                        relu=False, tile_n=64, group_cout=[n_c], algo_frac=n_c / n_cp))
                # FeatureAlign's DeformConv2d as planar sampler (columns [pixel][tap*C + c] for all levels) + planar 1x1
                # convolution over kh*kw*C channels (+ ReLU); offsets as one [pixels, 4] x [4, 2K] product (ada)
                ad = fa.conv_adaption
                O, Cin, akh, akw = ad.weight.shape
                if ad.deform_groups == 1 and Cin == 256 and akh * akw in (9, 15):
                    wk = ad.weight.detach().permute(0, 2, 3, 1).reshape(O, akh * akw * Cin, 1, 1)
                    entry.append(PlanarConv(wk, None, 1, 0, relu=True))
                    # ... or, level by level where the grid is large enough, the whole DeformConv2d + ReLU as one kernel (csrc/dcn_fused.hip)
                    entry.append(PlanarConv(ad.weight, None, 1, fa.padding, relu=True)
                                 if DCN_FUSED and FCB_FUSED and self.fmt in (1, 2) and ops.deform_conv_fused_supported(Cin, O, (akh, akw), False, self.fmt) else None)
                else:
                    entry.append(None)
                    entry.append(None)
            self.finals.append(tuple(entry))
        self.head = head

    # ------------------------------------------------------------------------------------------------------------
    def run(self, bb_outs, planes=None):
        """bb_outs: the selected backbone outputs (C3, C4, C5) as fp32 NCHW tensors, or -- planes given -- their planar
        form [(planes, B, H, W), ...] from PlanarBackbone (bb_outs may then hold None).  Returns (fpn_outs, pred) as
        STMask.forward_single."""
        net, fpn = self.net, self.net.fpn
        n = self.n_lat
        toc = self.timer.toc if self.timer is not None else (lambda name: None)
        toc("backbone")
        lat, latp, x = [None] * n, [None] * n, None
        if planes is not None:
            # laterals + top-down pathway on the planar kernel: lat_j = conv1x1(C_j) + upsample(lat_{j+1}) in one epilogue
            B = planes[0][1]
            sizes = [(h, w) for (_, _, h, w) in planes]
            for i, conv in enumerate(self.fpn_lat):
                j = n - 1 - i
                h, w = sizes[j]
                res = None
                if x is not None:
                    if fpn.interpolation_mode == "bilinear":
                        # the upsampled coarser level goes straight to planes (the residual form the epilogue reads fastest):
                        # no fp32 upsampled tensor, no layout copy
                        res = ops.resize_bilinear_planes(x, (h, w), self.fmt)
                    else:
                        up = F.interpolate(x.permute(0, 3, 1, 2), size=(h, w), mode=fpn.interpolation_mode)
                        res = _nhwc(up).view(-1, conv.O)
                if j > 0:                                  # a finer level follows: it upsamples this one (fp32 NHWC)
                    y32, latp[j] = conv(planes[j][0], ("img", B, h, w), out="both", residual=res)
                    x = y32.view(B, h, w, conv.O)
                else:
                    latp[j] = conv(planes[j][0], ("img", B, h, w), out="planes", residual=res)
        else:
            # fp32 NCHW inputs (module backbone): laterals + top-down pathway left to the GEMM library
            B = bb_outs[0].shape[0]
            for i, layer in enumerate(fpn.lat_layers):
                j = n - 1 - i
                lateral = layer(bb_outs[j])
                if x is None:
                    x = lateral
                else:
                    h, w = bb_outs[j].shape[2:]
                    x = F.interpolate(x, size=(h, w), mode=fpn.interpolation_mode, align_corners=False) + lateral
                lat[j] = x
            sizes = [tuple(t.shape[2:]) for t in lat]
        toc("fpn_lateral")
        for d in self.fpn_down:
            h, w = sizes[-1]
            sizes.append(ops.conv_out_hw(h, w, d.kh, d.kw, d.sh, d.sw, d.ph, d.pw, 1, 1))
        starts = [0]
        for h, w in sizes:
            starts.append(starts[-1] + B * h * w)
        ntot, nf = starts[-1], self.fpn_pred[0].O
        dev = (latp[0] if latp[0] is not None else lat[0]).device
        NP, pdt = _planes_dtype(self.fmt)
        feat = torch.empty(NP, nf // 32, ntot, 32, device=dev, dtype=pdt)   # P3..P7, all levels, planar
        feat32 = torch.empty(ntot, nf, device=dev, dtype=torch.float32) if not self.head_planar else None
        fpn_outs = [None] * len(sizes)

        def pred_level(i):
            conv = self.fpn_pred[i]
            j = n - 1 - i
            h, w = sizes[j]
            xp = latp[j] if latp[j] is not None else _split(_nhwc(lat[j]), self.fmt)
            if feat32 is not None:     # module-path head: it wants every level in fp32 as well
                conv(xp, ("img", B, h, w), out="both", out_planes=feat, out_f32=feat32, out_off=starts[j])
                fpn_outs[j] = feat32[starts[j]:starts[j + 1]].view(B, h, w, nf).permute(0, 3, 1, 2)
            elif j == self.cor_idx:
                # the correlation level is also wanted in fp32 (temporal fusion): second output of the same epilogue
                y32, _ = conv(xp, ("img", B, h, w), out="both", out_planes=feat, out_off=starts[j])
                fpn_outs[j] = y32.view(B, h, w, nf).permute(0, 3, 1, 2)
            else:
                conv(xp, ("img", B, h, w), out="planes", out_planes=feat, out_off=starts[j])

        def down_levels():
            for i, conv in enumerate(self.fpn_down):
                j = n + i
                h, w = sizes[j - 1]
                if feat32 is not None:
                    conv(feat, ("img", B, h, w), out="both", x_off=starts[j - 1], out_planes=feat, out_f32=feat32, out_off=starts[j])
                    fpn_outs[j] = feat32[starts[j]:starts[j + 1]].view(B, *sizes[j], nf).permute(0, 3, 1, 2)
                else:
                    conv(feat, ("img", B, h, w), out="planes", x_off=starts[j - 1], out_planes=feat, out_off=starts[j])

        # Second-stream branches (TRUNK_BRANCHES), only while a HIP graph is captured: the graph then holds parallel paths, and on small batches --
        # where a launch fills a fraction of the 256 CUs -- the GPU runs them side by side.  Same kernels on the same data: bit-equal to the plain order.
        side = self._side_stream(dev, B)
        if side is not None and TRUNK_BRANCHES >= 2 and len(self.fpn_down) > 0:
            # coarsest level first on the main stream, then P6 / P7 (which only need it) on the side stream beside the finer levels
            pred_level(0)
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), ops.workspace_branch("fpn_down"):
                down_levels()
            for i in range(1, len(self.fpn_pred)):
                pred_level(i)
            main.wait_stream(side)
        else:
            for i in range(len(self.fpn_pred)):
                pred_level(i)
            down_levels()
        if feat32 is None and self.cor_idx is not None and fpn_outs[self.cor_idx] is None:
            # fp32 NCHW view of the correlation level, rebuilt from its planes (exact)
            j = self.cor_idx
            fpn_outs[j] = ops.planes_to_f32(feat[:, :, starts[j]:starts[j + 1]]).view(B, *sizes[j], nf).permute(0, 3, 1, 2)

        toc("fpn_pred_down")

        # ---- proto-net on P3 (mask_proto_src) ------------------------------------------------------------------
        def proto_net():
            src = net.proto_src
            h, w = sizes[src]
            xp, x_off, cur = feat, starts[src], None
            n_layers = len(self.proto)
            for li, layer in enumerate(self.proto):
                last = li == n_layers - 1
                if isinstance(layer, PlanarConv):
                    nxt_is_interp = (not last) and not isinstance(self.proto[li + 1], PlanarConv)
                    if cur is not None:            # fp32 NHWC tensor pending a split
                        xp, x_off = _split(cur, self.fmt), 0
                        cur = None
                    if last or nxt_is_interp:
                        y = layer(xp, ("img", B, h, w), out="f32", x_off=x_off)
                        cur = y.view(B, h, w, layer.O)
                    else:
                        xp, x_off = layer(xp, ("img", B, h, w), out="planes", x_off=x_off), 0
                else:                              # bilinear upsample
                    kw = layer.kwargs
                    sf = kw.get("scale_factor")
                    nxt_conv = (not last) and isinstance(self.proto[li + 1], PlanarConv)
                    if (nxt_conv and not layer.args and kw.get("mode") == "bilinear" and not kw.get("align_corners", False)
                            and isinstance(sf, (int, float)) and float(sf).is_integer() and set(kw) <= {"scale_factor", "mode", "align_corners"}):
                        # ... straight into the next convolution's planes (no fp32 upsampled tensor)
                        h, w = h * int(sf), w * int(sf)
                        xp, x_off, cur = ops.resize_bilinear_planes(cur, (h, w), self.fmt), 0, None
                    else:
                        t = layer(cur.permute(0, 3, 1, 2))
                        h, w = t.shape[2:]
                        cur = _nhwc(t)
            return cur                         # [B, 2h, 2w, 32], ReLU applied by the last layer (STMask.py:227)

        proto_side = side is not None and TRUNK_BRANCHES >= 1 and self.head_planar
        if proto_side:
            # proto-net and the shared head both start from `feat` and meet in the detection stage: proto-net on the side stream, joined at the end
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side), ops.workspace_branch("proto"):
                proto = proto_net()
        else:
            proto = proto_net()
        toc("proto_net")

        keys = ("mask_coeff", "priors", "loc", "T2S_feat", "centerness", "conf", "track")
        pred = {k: [] for k in keys}
        if not self.head_planar:
            for idx, layer in zip(net.selected_layers, net.prediction_layers):
                p = layer(fpn_outs[idx])
                for k in keys:
                    pred[k].append(p[k])
            for k in keys:
                if k != "T2S_feat":
                    pred[k] = torch.cat(pred[k], 1)
            pred["proto"] = proto
            return fpn_outs, pred

        # ---- shared prediction head, all levels per launch -------------------------------------------------------
        lv = ("levels", B, sizes)
        head = self.head
        if self.cor_idx is not None:
            up32, up = self.up(feat, lv, out="both")
        else:
            up32, up = None, self.up(feat, lv, out="planes")
        t1 = self.tower1(up, lv, out="planes")
        cw = self.tower2.O // 4                                       # channels per branch in t2 (conf, bbox, mask, track)
        P = self.GROUP_PAD
        if not self.fcb:
            t2 = self.tower2(t1, lv, out="planes")
            toc("head_towers")
            outs = [(small(t2, lv, out="f32"), trk(t2, lv, out="f32", x_ch_off=3 * cw)) for small, trk in self.finals]
        else:
            t2_32, t2 = self.tower2(t1, lv, out="both")               # the class branch also leaves as fp32 for the sampler
            toc("head_towers")
            # conf_x per level as NCHW fp32 (shared by the three kernel shapes)
            conf_x = None
            outs = []
            for small, trk, fa, fconv, adconv, adfused in self.finals:
                buf = torch.empty(ntot, 3 * P, device=dev, dtype=torch.float32)   # [conf | centerness+bbox | mask] groups
                small(t2, lv, out="f32", out_f32=buf, x_ch_off=cw, out_ch_off=P)
                npri = head.num_priors
                if adconv is not None and npri == 1 and FCB_PLANAR:
                    # all-planar class branch: offsets pixel-major, sampler per level into one column buffer, one 1x1 conv
                    kh, kw = fa.kernel_size
                    K = kh * kw
                    bbox_pix = buf[:, P + npri:P + npri + 4]                       # [ntot, 4] box regression of this shape
                    if fa.use_pred_offset:
                        off = bbox_pix @ fa.conv_offset.weight.view(2 * K, 4).t()   # Featurealign.py:40-43 (1x1 conv, no bias)
                    NP_, pdt_ = _planes_dtype(self.fmt)
                    # levels whose grid fills the chip take the fused kernel (no columns); the coarser ones, contiguous at the end of the pixel axis,
                    # share one column buffer and one 1x1 product as before -- both write the same feature planes
                    n_fused = 0
                    if adfused is not None:
                        while n_fused < len(sizes) and ops.deform_conv_fused_tiles(B, sizes[n_fused][0], sizes[n_fused][1], adfused.O) >= DCN_FUSED_MIN_TILES:
                            n_fused += 1
                    c0 = starts[n_fused]                                         # first pixel of the column-buffer levels
                    feat_pl = torch.empty(NP_, cw // 32, ntot, 32, device=dev, dtype=pdt_)
                    cols = torch.empty(NP_, K * cw // 32, ntot - c0, 32, device=dev, dtype=pdt_) if ntot > c0 else None
                    for l, (hh, ww) in enumerate(sizes):
                        sl = slice(starts[l], starts[l + 1])
                        if fa.use_pred_offset:
                            off_l = off[sl]
                        else:
                            loc_l = bbox_pix[sl].reshape(B, hh, ww, 4).permute(0, 3, 1, 2).contiguous()
                            off_l = ops.fcb_ali_offsets(loc_l, kh, kw).permute(0, 2, 3, 1).reshape(-1, 2 * K)
                        if l < n_fused:
                            adfused.deform(t2_32[sl, 0:cw], B, hh, ww, off_l, 1, fa.padding, 1, has_mask=False, out=feat_pl, out_off=starts[l])
                        else:
                            ops.deform_sample_planar(t2_32[sl, 0:cw], B, hh, ww, cw, off_l, (kh, kw), fa.padding, cols, starts[l] - c0, self.fmt)
                    if cols is not None:
                        adconv(cols, ("img", 1, 1, ntot - c0), out_planes=feat_pl, out_off=c0)      # DeformConv2d's GEMM + ReLU
                    fconv(feat_pl, lv, out="f32", out_f32=buf)                   # conf logits into columns [0, n_cls)
                    outs.append((buf, trk(t2, lv, out="f32", x_ch_off=3 * cw)))
                    continue
                if conf_x is None:
                    conf_x = [t2_32[starts[l]:starts[l + 1], 0:cw].reshape(B, hh, ww, cw).permute(0, 3, 1, 2).contiguous()
                              for l, (hh, ww) in enumerate(sizes)]
                feat_k = torch.empty(ntot, cw, device=dev, dtype=torch.float32)
                for l, (hh, ww) in enumerate(sizes):
                    sl = slice(starts[l], starts[l + 1])
                    bbox_cur = buf[sl, P + npri:P + npri + 4 * npri].reshape(B, hh, ww, 4 * npri).permute(0, 3, 1, 2).contiguous()
                    if fa.use_pred_offset:
                        offset = fa.conv_offset(bbox_cur)
                    else:
                        offset = ops.fcb_ali_offsets(bbox_cur, fa.kernel_size[0], fa.kernel_size[1])
                    y = ops.deform_conv(conf_x[l], offset, None, fa.conv_adaption.weight, None, 1, fa.padding, 1,
                                        fa.conv_adaption.deform_groups, relu=True)
                    feat_k[sl] = y.permute(0, 2, 3, 1).reshape(-1, cw)
                fconv(ops.split_planes(feat_k, self.fmt), lv, out="f32", out_f32=buf)        # conf logits into columns [0, n_cls)
                outs.append((buf, trk(t2, lv, out="f32", x_ch_off=3 * cw)))
        toc("head_finals")
        P = self.GROUP_PAD
        ncls, nbox, nmask, ntrk = self.dims
        npri = head.num_priors
        t2s = [None] * len(sizes)
        if up32 is not None:
            l = self.cor_idx
            t2s[l] = up32[starts[l]:starts[l + 1]].view(B, *sizes[l], -1).permute(0, 3, 1, 2)
        for hh, ww in sizes:
            pred["priors"].append(head.make_priors(hh, ww, dev))
        pred["priors"] = torch.cat(pred["priors"], 1)
        if npri == 1:
            # one kernel for the reference's cat / view / tanh / normalize tail over all levels and kernel shapes
            conf, loc, mask, track, cen = ops.head_assemble([o[0] for o in outs], [o[1] for o in outs], B, sizes,
                                                            head.num_classes, head.mask_dim, head.embed_dim, P)
            pred["conf"], pred["loc"], pred["mask_coeff"], pred["track"], pred["centerness"] = conf, loc, mask, track, cen
        else:
            conf, loc, mask, track, cen = [], [], [], [], []
            for l, (hh, ww) in enumerate(sizes):
                sl = slice(starts[l], starts[l + 1])
                per_k = [o[0][sl].view(B, hh * ww, 3 * P) for o in outs]
                cat = torch.stack(per_k, dim=2)                                           # [B, HW, K, 3P]
                conf.append(cat[..., 0:ncls].reshape(B, -1, head.num_classes))
                loc.append(cat[..., P + npri:P + npri + nbox].reshape(B, -1, 4))
                mask.append(cat[..., 2 * P:2 * P + nmask].reshape(B, -1, head.mask_dim))
                track.append(torch.stack([o[1][sl].view(B, hh * ww, ntrk) for o in outs], dim=2).reshape(B, -1, head.embed_dim))
                # the reference concatenates centerness along H (prediction_head_FC.py:189): order (k, y, x)
                cen.append(torch.stack([pk[..., P:P + npri] for pk in per_k], dim=1).reshape(B, -1, 1))
            pred["conf"] = torch.cat(conf, 1)
            pred["loc"] = torch.cat(loc, 1)
            pred["mask_coeff"] = torch.cat(mask, 1)
            pred["track"] = F.normalize(torch.cat(track, 1), dim=-1)
            pred["centerness"] = torch.tanh(torch.cat(cen, 1))
        pred["T2S_feat"] = t2s
        pred["proto"] = proto
        if proto_side:
            torch.cuda.current_stream().wait_stream(side)
        toc("head_assemble")
        return fpn_outs, pred

    def _side_stream(self, dev, n_images):
        """The second stream of run()'s branches, or None: branches exist only inside a HIP-graph capture (eager passes keep the plain order: tensors
        crossing streams would need allocator bookkeeping there, and the eager path is not the one that is timed) and only for small batches."""
        if TRUNK_BRANCHES <= 0 or n_images > BRANCH_MAX_IMAGES or self.timer is not None and self.timer.on:
            return None
        if not torch.cuda.is_current_stream_capturing():
            return None
        if getattr(self, "_side", None) is None or self._side.device != torch.device(dev):
            self._side = torch.cuda.Stream(device=dev)
        return self._side


# (row class, column class, first / end kernel row, first / end kernel column) of the nine border classes of a 3x3 / pad-1 convolution
_BORDER_CLASSES = [(ry, rx, k0y, k1y, k0x, k1x) for ry, (k0y, k1y) in enumerate(((1, 3), (0, 3), (0, 2)))
                   for rx, (k0x, k1x) in enumerate(((1, 3), (0, 3), (0, 2)))]


def border_windows(h, w):
    """The non-empty border classes of a 3x3 / pad-1 convolution on an h x w map as windows of stm_conv2d_planar_windows_f32:
    [(class index into _BORDER_CLASSES, (kh, kw, ph, pw, Ho, Wo, y0, x0))].  Class (ry, rx) holds output rows {0} / {1..h-2} / {h-1} and
    columns likewise; its sub-kernel is the taps that can be inside the map; output (y0 + oy, x0 + ox) reads input
    (y0 + oy + k0y - 1 + ky', x0 + ox + k0x - 1 + kx'), i.e. padding 1 - y0 - k0y (<= 0)."""
    rows, cols = ((0, 1), (1, h - 1), (h - 1, h)), ((0, 1), (1, w - 1), (w - 1, w))
    out = []
    for ci, (ry, rx, k0y, k1y, k0x, k1x) in enumerate(_BORDER_CLASSES):
        (y0, y1), (x0, x1) = rows[ry], cols[rx]
        if y1 <= y0 or x1 <= x0:
            continue
        out.append((ci, (k1y - k0y, k1x - k0x, 1 - y0 - k0y, 1 - x0 - k0x, y1 - y0, x1 - x0, y0, x0)))
    return out


class PlanarTemporalNet:
    """TemporalNet (track_to_segment_head.py:10-37: 3 x (3x3 conv + ReLU) on 7x7 RoI features, 7x7 average pool, two
    linear layers) on the planar convolution.  The 633 input channels (121 correlation + 2 x 256 features) are padded to
    640 with zero weights; any number of RoIs goes through in one launch per layer, so the fixed-shape RoI blocks the
    dense-conv library needed (one kernel selection per shape) disappear."""

    def __init__(self, tn, corr_channels=121):
        w1 = tn.conv1.weight.detach()
        self.cin = w1.shape[1]
        self.cpad = -(-self.cin // 32) * 32
        # input channel order of the planes: [T2S_prev | T2S | corr | zero padding] (stm_roi_align_planes_f32 writes them so:
        # the two NHWC feature maps first, 8-channel groups aligned); the module's order is [corr | T2S_prev | T2S]
        self.ncorr = corr_channels if 0 < corr_channels < self.cin and (self.cin - corr_channels) % 16 == 0 else 0
        self.perm = torch.cat([torch.arange(self.ncorr, self.cin), torch.arange(0, self.ncorr)]).to(w1.device)
        w1 = w1.index_select(1, self.perm)
        w1 = F.pad(w1, (0, 0, 0, 0, 0, self.cpad - self.cin))
        self.c1 = PlanarConv(w1, tn.conv1.bias, 1, tn.conv1.padding, relu=True, algo_frac=self.cin / self.cpad)
        self.c2 = PlanarConv(tn.conv2.weight, tn.conv2.bias, 1, tn.conv2.padding, relu=True)
        self.c3 = PlanarConv(tn.conv3.weight, tn.conv3.bias, 1, tn.conv3.padding, relu=True)
        self.fc, self.fc_coeff = tn.fc, tn.fc_coeff
        self.fmt = FMT
        for c in (self.c1, self.c2, self.c3):
            c.role = "temporal"
        # Border classes.  A 3x3 / pad-1 convolution on 7x7 RoI maps multiplies zeros in 80 of its 441 tap-pixels: the top row has no
        # ky = 0 tap, the left column no kx = 0 tap, ...  Output pixels of one (row class, column class) share their set of real taps --
        # a contiguous sub-kernel -- and form a rectangle of every RoI map, so each class is a plain convolution with that sub-kernel on a
        # window of the map, and the nine classes of a layer run as ONE grid (stm_conv2d_planar_windows_f32): 361 / 441 of the matrix work,
        # the same sums (a skipped tap added exact zeros).  Classes: rows {0}, {1 .. H-2}, {H-1} x columns likewise.  (As nine separate
        # launches the small classes -- 2 500 .. 20 000 pixels -- lost in partial last rounds what the skipped taps saved.)
        self.border = None
        ws = [(w1, tn.conv1.bias, self.cin / self.cpad), (tn.conv2.weight.detach(), tn.conv2.bias, 1.0), (tn.conv3.weight.detach(), tn.conv3.bias, 1.0)]
        if (TN_BORDER and self.fmt == 1 and all(tuple(w.shape[2:]) == (3, 3) and w.shape[0] % 128 == 0 for w, _, _ in ws)
                and all(tuple(_pair(c.padding)) == (1, 1) and tuple(_pair(c.stride)) == (1, 1) for c in (tn.conv1, tn.conv2, tn.conv3))):
            self.border = [{"w": w.float().contiguous(), "b": (b.detach().float().contiguous() if b is not None else None), "frac": frac, "packed": None}
                           for w, b, frac in ws]
        # tail (track_to_segment_head.py:33-37): fc (4 rows) and fc_coeff stacked, for stm_temporal_pool_fc_f32
        self.w_tail = torch.cat([tn.fc.weight.detach().float(), tn.fc_coeff.weight.detach().float()], 0).contiguous()
        self.b_tail = torch.cat([tn.fc.bias.detach().float(), tn.fc_coeff.bias.detach().float()], 0).contiguous()
        self.n_fc = tn.fc.weight.shape[0]
        self._pool = None            # int64 [capacity, 1024] fixed-point pooled sums: zero between steps (the tail kernel clears what it reads)

    def __call__(self, roi_feats):
        """roi_feats [n, 633, 7, 7] fp32 -> (loc shift [n, 4], coeff shift [n, 32])."""
        n, c, h, w = roi_feats.shape
        x = F.pad(roi_feats.index_select(1, self.perm.to(roi_feats.device)).permute(0, 2, 3, 1), (0, self.cpad - c)).contiguous()   # NHWC, padded
        return self.forward_planes(ops.split_planes(x, self.fmt), n, h, w)

    _CLASSES = _BORDER_CLASSES

    def _border_layer(self, li, xp, n, h, w, out):
        """One 3x3 layer as its nine border-class windows in one launch; returns planes [2, O/32, n*h*w, 32] or the fp32 matrix [n*h*w, O]."""
        L = self.border[li]
        O, C = L["w"].shape[0], L["w"].shape[1]
        if L["packed"] is None:
            ops.planar_range_flag()
            wscale = ops._pow2_wscale(L["w"])
            L["packed"] = [ops.conv_pack_weights(L["w"][:, :, k0y:k1y, k0x:k1x].contiguous(), tile_n=128, fmt=1, wscale=wscale)[0]
                           for _, _, k0y, k1y, k0x, k1x in self._CLASSES]
            L["out_scale"] = 1.0 / wscale
        dev = xp.device
        out_planes = torch.empty(2, O // 32, n * h * w, 32, device=dev, dtype=torch.float16) if out == "planes" else None
        out_f32 = torch.empty(n * h * w, O, device=dev, dtype=torch.float32) if out == "f32" else None
        if out == "pool" and (self._pool is None or self._pool.shape[0] < n or self._pool.shape[1] != O or self._pool.device != dev):
            self._pool = torch.zeros(max(n, 2 * (self._pool.shape[0] if self._pool is not None else 0)), O, device=dev, dtype=torch.int64)
        wins, packed, macs = [], [], 0
        for ci, win in border_windows(h, w):
            wins.append(win)
            packed.append(L["packed"][ci])
            macs += win[4] * win[5] * win[0] * win[1]
        timing = ops._conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if out == "pool":
            # the launch ADDS into the pooled sums and only the tail kernel re-zeroes them: a failure in between must not leave stale sums for the next step
            try:
                ops.conv2d_planar_windows_pool(xp, packed, wins, L["b"], n, h, w, C, O, h, w, L["out_scale"], self._pool)
            except Exception:
                self._pool.zero_()
                raise
        else:
            ops.conv2d_planar_windows(xp, packed, wins, L["b"], n, h, w, C, O, h, w, L["out_scale"], relu=True, out_f32=out_f32, out_planes=out_planes)
        if timing is not None:
            e1.record()
            M = n * h * w
            nbytes = M * C * 4 + (n * O * 8 if out == "pool" else M * O * 4) + L["w"].numel() * 4
            # algorithmic flops as for every other layer: the reference's 2 M Cout Cin kh kw (its padded taps included), priced against the
            # format's peak (3 MFMA products per fp32 product); 8th field: share of those products that is actually issued (361 / 441 on 7x7)
            timing.append((e0, e1, 2.0 * M * 9 * O * C * L["frac"], (M, C, O, 3, 1, 1, -2), 3, "temporal", float(nbytes), macs / (h * w * 9.0)))
        return out_planes if out == "planes" else (self._pool if out == "pool" else out_f32)

    def forward_planes(self, xp, n, h=7, w=7):
        """xp: the RoI features as planes [P, cpad/32, n*h*w, 32] in this object's channel order (ops.roi_align_planes)."""
        shape = ("img", n, h, w)
        # (under ~16 000 pixels the classes' partial tiles cost what their skipped taps save: 114 RoIs = 28 tiles x 7 taps against 22 x 9)
        if self.border is not None and h >= 3 and w >= 3 and n * h * w >= 16384:
            x2 = self._border_layer(1, self._border_layer(0, xp, n, h, w, "planes"), n, h, w, "planes")
            if TN_POOL:
                # conv3 + ReLU + AvgPool2d in one launch (pooled sums, no [n*49, 1024] tensor), then mean -> fc / fc_coeff in one more
                pool = self._border_layer(2, x2, n, h, w, "pool")
                try:
                    return ops.temporal_pool_fc(pool, n, h * w, self.w_tail, self.b_tail, n_first=self.n_fc)
                except Exception:
                    self._pool.zero_()
                    raise
            y = self._border_layer(2, x2, n, h, w, "f32")
        else:
            y = self.c3(self.c2(self.c1(xp, shape), shape), shape, out="f32")                # [n*h*w, 1024]
        pooled = y.view(n, h * w, -1).mean(dim=1)                                        # AvgPool2d((7, 7)) on a 7x7 map
        return self.fc(pooled), self.fc_coeff(pooled)


class PlanarChain:
    """The tail of one 64-channel bottleneck and the head of the next as one launch (stm_bottleneck_chain_f32, csrc/conv_chain.hip;
    reference backbone.py:38-58): relu(conv3(relu(conv2(mid1))) + x) and, when the next block's conv1 is given, relu(conv1(.))."""

    def __init__(self, c2, c3, c1_next=None, proj=None):
        """proj: the 1x1 / stride-1 projection of a stage's first block (its shortcut), None = identity shortcut."""
        self.wds = proj.weight.detach().float().contiguous() if proj is not None else None
        self.w2, self.b2 = c2.weight.detach().float().contiguous(), (c2.bias.detach().float().contiguous() if c2.bias is not None else None)
        self.w3, self.b3 = c3.weight.detach().float().contiguous(), (c3.bias.detach().float().contiguous() if c3.bias is not None else None)
        self.w1 = c1_next.weight.detach().float().contiguous() if c1_next is not None else None
        self.b1 = c1_next.bias.detach().float().contiguous() if (c1_next is not None and c1_next.bias is not None) else None
        if proj is not None and proj.bias is not None:
            self.b3 = proj.bias.detach().float() + (self.b3 if self.b3 is not None else 0.0)
        self._packed = None
        self.role = "trunk"

    @staticmethod
    def eligible(c2, c3, blk, fmt, out_fmt):
        d = blk.downsample[0] if blk.downsample is not None else None
        proj_ok = d is None or (isinstance(d, torch.nn.Conv2d) and tuple(d.weight.shape) == (256, 64, 1, 1) and tuple(d.stride) == (1, 1)
                                and tuple(d.padding) == (0, 0) and d.groups == 1)
        if d is not None and not (CHAIN_MODE & 1):
            return False
        return (CONV_CHAIN and fmt == 1 and out_fmt == 1 and proj_ok and isinstance(c2, torch.nn.Conv2d)
                and tuple(c2.weight.shape) == (64, 64, 3, 3) and tuple(c2.stride) == (1, 1) and tuple(c2.padding) == (1, 1)
                and tuple(c2.dilation) == (1, 1) and c2.groups == 1 and tuple(c3.weight.shape) == (256, 64, 1, 1) and tuple(c3.stride) == (1, 1))

    @staticmethod
    def takes_z(c1):
        return isinstance(c1, torch.nn.Conv2d) and tuple(c1.weight.shape) == (64, 256, 1, 1) and tuple(c1.stride) == (1, 1) and c1.groups == 1

    def __call__(self, mid1, x, B, H, W):
        if self._packed is None:
            g = _lib.ConvGeom()
            g.C, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, g.groups, g.fmt = 64, 64, 3, 3, 1, 1, 1, 1, 1, 1
            ops.planar_range_flag()
            w2p, s2 = ops.conv_pack_weights_kxr(self.w2, g)
            tail, s3, s1 = ops.chain_pack_tail(self.w3, self.w1, self.wds)
            self._packed = (w2p, tail, (s2, s3, s1))
        w2p, tail, scales = self._packed
        timing = ops._conv_timing
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        y, z = ops.bottleneck_chain(mid1, x, w2p, tail, self.b2, self.b3, self.b1, scales, B, H, W, want_z=self.w1 is not None, proj=self.wds is not None)
        if timing is not None:
            e1.record()
            M = B * H * W
            macs = 64 * 64 * 9 + 256 * 64 * (2 if self.wds is not None else 1) + (256 * 64 if self.w1 is not None else 0)
            nbytes = M * 4 * (64 + (64 if self.wds is not None else 256) + 256 + (64 if self.w1 is not None else 0)) + 4 * macs
            timing.append((e0, e1, 2.0 * M * macs, (M, 64, 256, 3, 1, 1, -1), 3, self.role, float(nbytes)))
        return y, z


class PlanarBackbone:
    """ResNet bottlenecks (backbone.py:38-58 of the reference, eval BatchNorm folded) with every 1x1 convolution, the
    plain 3x3 convolutions and the stride-s downsample projections on the planar convolution; residual add + ReLU in
    conv3's epilogue.  The stem's 7x7 / stride-2 convolution (3 input channels) is a (7 x 1) planar convolution over the
    row-patch tensor of stm_stem_rows_planes_f32, its bias + ReLU + max-pool go straight to planes
    (stm_bias_relu_maxpool_planes_f32); the deformable 3x3 layers run as planar offset conv
    -> planar sampler (columns as planes) -> planar 1x1 convolution over 9C channels (other DCN shapes: NCHW kernels)."""

    OM_PLANAR_MIN_PIXELS = 0    # offset / mask conv of a DCN layer on the planar kernel from this many output pixels (with
                                # split-K the small stages are fine there too: 547-549 vs 538-539 frames/s with the library)

    def __init__(self, bb, selected=None):
        """selected: indices of the stages whose outputs the FPN reads (they leave in the graph's format FMT even when the
        backbone computes in fp16x1); None = all."""
        from .dcn_v2 import DCN
        self.bb = bb
        self.fmt = FMT if BACKBONE_FMT is None else BACKBONE_FMT
        self.graph_fmt = FMT
        fmt = self.fmt
        self.planes_only = False    # fuse: the planar FPN takes the stage outputs as planes; their fp32 copies are not made
        self.out_planes = None      # [(planes, B, H, W)] of the last call, one entry per stage
        self._stem_fused = None     # (packed conv1 weights, 1 / wscale) of the one-kernel stem
        # stem: w'[o][j][ky][0] = w[o][j % Cin][ky][j / Cin] over the row-patch tensor of stm_stem_rows_planes_f32
        c1 = bb.conv1
        O, Cin, kh, kw = c1.weight.shape
        self.stem = None
        if kw * Cin <= 32 and c1.groups == 1 and tuple(c1.dilation) == (1, 1):
            w = c1.weight.detach().permute(0, 2, 3, 1).reshape(O, kh, kw * Cin)          # [o][ky][kx*Cin + c]
            w = F.pad(w, (0, 32 - kw * Cin)).permute(0, 2, 1).reshape(O, 32, kh, 1).contiguous()
            self.stem = PlanarConv(w, None, (c1.stride[0], 1), (c1.padding[0], 0), relu=False, algo_frac=kw * Cin / 32.0, fmt=fmt)
        self.blocks = []
        for si, layer in enumerate(bb.layers):
            blks = []
            for bi, blk in enumerate(layer):
                c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
                # the last block of a stage the FPN reads hands its output over in the graph's format (a fp16x1 layer can
                # write both planes of fp16x2; plane 0 of that tensor is what the next fp16x1 stage reads)
                to_graph = bi == len(layer) - 1 and (selected is None or si in selected)
                e = {"c1": PlanarConv(c1.weight, c1.bias, 1, 0, relu=True, fmt=fmt),
                     "c3": PlanarConv(c3.weight, c3.bias, 1, 0, relu=True, fmt=fmt, out_fmt=self.graph_fmt if to_graph else fmt)}
                if isinstance(c2, DCN):
                    e["dcn"] = c2
                    om = c2.conv_offset_mask
                    # 27 offset / mask channels in rows of 32 (zero weights behind them): whole 16-channel tiles for the kx-reuse
                    # kernel, 16-byte rows for everyone; the sampler reads the first 27 values of a row
                    n_om = om.weight.shape[0]
                    n_pad = -(-n_om // 16) * 16
                    e["om"] = PlanarConv(F.pad(om.weight, (0, 0, 0, 0, 0, 0, 0, n_pad - n_om)), F.pad(om.bias, (0, n_pad - n_om)), om.stride, om.padding,
                                         relu=False, fmt=fmt, group_cout=[n_om], algo_frac=n_om / n_pad)
                    e["n_om"] = n_om
                    # the deformable conv's GEMM as a planar 1x1 convolution over the sampled columns [pixel][tap*C + c]
                    O, Cin = c2.weight.shape[:2]
                    e["dcn_planar"] = (c2.kernel_size == (3, 3) and c2.deformable_groups == 1 and Cin in (128, 256, 512))
                    if e["dcn_planar"]:
                        wk = c2.weight.detach().permute(0, 2, 3, 1).reshape(O, 9 * Cin, 1, 1)
                        e["dcn_conv"] = PlanarConv(wk, c2.bias, 1, 0, relu=True, fmt=fmt)
                        # ... or the whole deformable convolution as one kernel: the ORIGINAL 3x3 weight packed channel-slab outer / tap inner
                        if DCN_FUSED and fmt in (1, 2) and ops.deform_conv_fused_supported(Cin, O, 3, True, fmt):
                            e["dcn_fused"] = PlanarConv(c2.weight, c2.bias, c2.stride, c2.padding, relu=True, fmt=fmt)
                else:
                    e["c2"] = PlanarConv(c2.weight, c2.bias, c2.stride, c2.padding, relu=True, fmt=fmt)
                if blk.downsample is not None:
                    d = blk.downsample[0]
                    e["ds"] = PlanarConv(d.weight, d.bias, d.stride, 0, relu=False, fmt=fmt)
                    if (C3DS_FUSED and fmt in (1, 2) and tuple(d.kernel_size) == (1, 1) and d.stride[0] == d.stride[1] and tuple(d.padding) == (0, 0)
                            and d.bias is not None and c3.bias is not None and d.weight.shape[1] % 32 == 0):
                        # conv3 and the projection shortcut as ONE two-source product (stm_conv2d_planar_dual_f32): W = [W3 | Wds],
                        # bias = b3 + bds; the projection's output tensor is never written
                        wcat = torch.cat([c3.weight.detach(), d.weight.detach()], 1)
                        e["c3ds"] = PlanarConv(wcat, c3.bias.detach() + d.bias.detach(), 1, 0, relu=True, fmt=fmt,
                                               out_fmt=self.graph_fmt if to_graph else fmt)
                        e["ds_stride"] = d.stride[0]
                e["stride"] = _pair(c2.stride)
                if PlanarChain.eligible(c2, c3, blk, fmt, e["c3"].out_fmt):
                    nxt = layer[bi + 1] if bi + 1 < len(layer) else None
                    give_z = bool(CHAIN_MODE & 2) and nxt is not None and not isinstance(nxt.conv2, DCN) and PlanarChain.takes_z(nxt.conv1)
                    e["chain"] = PlanarChain(c2, c3, nxt.conv1 if give_z else None, blk.downsample[0] if blk.downsample is not None else None)
                blks.append(e)
            self.blocks.append(blks)

    def __call__(self, x):
        bb = self.bb
        c1, mp = bb.conv1, bb.maxpool
        if (isinstance(bb.bn1, torch.nn.Identity) and c1.out_channels % 32 == 0 and mp.kernel_size == 3 and mp.stride == 2 and mp.padding == 1
                and not mp.ceil_mode and mp.dilation == 1):
            # the whole stem as one kernel: no row-patch tensor, no fp32 convolution output
            if (STEM_FUSED and self.fmt in (1, 2) and tuple(c1.weight.shape) == (64, 3, 7, 7) and tuple(c1.stride) == (2, 2) and tuple(c1.padding) == (3, 3)
                    and c1.groups == 1 and tuple(c1.dilation) == (1, 1)):
                if self._stem_fused is None:
                    ops.planar_range_flag()
                    self._stem_fused = ops.stem_pack_weights(c1.weight.detach(), self.fmt)
                xp, (H, W) = ops.stem_fused(_nhwc(x), self._stem_fused[0], self._stem_fused[1], c1.bias, self.fmt)
                B, C = x.shape[0], c1.out_channels
                return self._stages(xp, B, C, H, W)
            # stem tail in one pass: folded-BN bias + ReLU + 3x3/2 max-pool of the raw 7x7 convolution output, straight to planes
            if self.stem is not None and STEM_PLANAR:
                # the 7x7 / stride-2 convolution itself as a (7 x 1) planar convolution over the row-patch tensor R
                B = x.shape[0]
                kh, kw = c1.kernel_size
                rp, Wo = ops.stem_rows_planes(_nhwc(x), kw, c1.stride[1], c1.padding[1], self.fmt)
                y = self.stem(rp, ("img", B, x.shape[2], Wo), out="f32")
                Ho = (x.shape[2] + 2 * c1.padding[0] - kh) // c1.stride[0] + 1
                y = y.view(B, Ho, Wo, c1.out_channels)
            else:
                y = _nhwc(F.conv2d(x, c1.weight, None, c1.stride, c1.padding, c1.dilation, c1.groups))
                B = y.shape[0]
            C = c1.out_channels
            xp, (H, W) = ops.bias_relu_maxpool_planes(y, c1.bias, self.fmt)
        else:
            x = bb.maxpool(bb.relu(bb.bn1(bb.conv1(x))))        # conv1 carries the folded BN + ReLU after fuse
            B, C, H, W = x.shape
            xp = _split(_nhwc(x), self.fmt)
        return self._stages(xp, B, C, H, W)

    def _stages(self, xp, B, C, H, W):
        outs, self.out_planes = [], []
        for blks in self.blocks:
            y32 = None
            z_next = None         # the next block's conv1 output, when the previous block's chain kernel already produced it
            for bi, e in enumerate(blks):
                last = bi == len(blks) - 1
                shape = ("img", B, H, W)
                sh, sw = e["stride"]
                Ho, Wo = (H - 1) // sh + 1, (W - 1) // sw + 1
                if "chain" in e and not (last and not self.planes_only) and B * H * W <= CHAIN_MAX_PIXELS:
                    mid1 = z_next if z_next is not None else e["c1"](xp, shape)
                    xp, z_next = e["chain"](mid1, xp, B, H, W)
                    if CHAIN_MODE & 4:          # diagnostics: the kernel computes z, the next block does not use it
                        z_next = None
                    continue
                if "dcn" in e:
                    # conv1 -> fp32 (NCHW copy for the deformable sampler) and planes (offset / mask convolution);
                    # the sampler reads the raw conv_offset_mask output (sigmoid folded in), GEMM adds bias + ReLU
                    d = e["dcn"]
                    if e["dcn_planar"]:
                        # all planar: conv1 -> fp32 NHWC (sampler input) + planes (offset conv input); offset conv -> fp32
                        # [pixels, 27]; sampler -> planar columns; GEMM + bias + ReLU as a planar 1x1 convolution
                        t32, tpl = e["c1"](xp, shape, out="both")
                        om = e["om"](tpl, shape, out="f32")
                        fz = e.get("dcn_fused")
                        if (fz is not None and ops.deform_conv_fused_tiles(B, Ho, Wo, fz.O) >= DCN_FUSED_MIN_TILES
                                and (DCN_FUSED_RULE == 0 or fz.O <= 128 or (fz.O <= 256 and _pair(d.stride)[0] == 2))):
                            mid = fz.deform(t32, B, H, W, om, d.stride, d.padding, d.dilation, has_mask=True)
                        else:
                            cols = ops.dcn_sample_planar(t32.view(B, H, W, -1), om, d.stride, d.padding, d.dilation, fmt=self.fmt)
                            mid = e["dcn_conv"](cols, ("img", B, Ho, Wo))
                    elif B * Ho * Wo >= self.OM_PLANAR_MIN_PIXELS:
                        t32, tpl = e["c1"](xp, shape, out="both")
                        om = e["om"](tpl, shape, out="f32")[:, :e["n_om"]].reshape(B, Ho, Wo, -1).permute(0, 3, 1, 2).contiguous()
                    else:
                        # alternative for tiny maps: the dense-conv library's small-tile kernel (not taken by default)
                        t32 = e["c1"](xp, shape, out="f32")
                        om = d.conv_offset_mask(t32.view(B, H, W, -1).permute(0, 3, 1, 2)).contiguous()
                    if not e["dcn_planar"]:
                        xin = t32.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
                        t = ops.deform_conv(xin, None, None, d.weight, d.bias, d.stride, d.padding, d.dilation,
                                            d.deformable_groups, relu=True, fused_om=om)
                        mid = _split(_nhwc(t), self.fmt)
                else:
                    mid = e["c2"](z_next if z_next is not None else e["c1"](xp, shape), shape)
                    z_next = None
                if "c3ds" in e:
                    x2 = (xp, H, W, e["ds_stride"])
                    H, W = Ho, Wo
                    if last and not self.planes_only:
                        y32, xp = e["c3ds"](mid, ("img", B, H, W), out="both", x2=x2)
                    else:
                        xp = e["c3ds"](mid, ("img", B, H, W), x2=x2)
                    continue
                res = e["ds"](xp, shape) if "ds" in e else xp
                H, W = Ho, Wo
                if last and not self.planes_only:
                    y32, xp = e["c3"](mid, ("img", B, H, W), out="both", residual=res)
                else:
                    xp = e["c3"](mid, ("img", B, H, W), residual=res)
            self.out_planes.append((xp, B, H, W))
            outs.append(y32.view(B, H, W, -1).permute(0, 3, 1, 2) if y32 is not None else None)   # channels_last NCHW view
        return tuple(outs)
