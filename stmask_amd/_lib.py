"""ctypes binding of libstmask_hip.so (include/stmask_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STM_LIBRARY") or os.path.join(_HERE, "libstmask_hip.so")   # STM_LIBRARY: an -DSTM_ABLATE build, for timing runs
_lib = None

c_i, c_l, c_f, c_p, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# every symbol include/stmask_hip.h declares (tests/test_abi.py checks the library exports all of them)
ABI_VERSION = 4   # include/stmask_hip.h STM_ABI_VERSION
ABI_SYMBOLS = [
    "stm_version", "stm_last_error_string", "stm_struct_bytes", "stm_debug_reload_tunables", "stm_debug_launch_count", "stm_conv_kxr_packed_bytes", "stm_conv_pack_weights_kxr_f32", "stm_conv2d_planar_kxr_f32", "stm_conv2d_planar_dual_f32", "stm_conv2d_planar_windows_f32", "stm_conv2d_planar_windows_pool_f32", "stm_temporal_pool_fc_f32", "stm_stem_packed_weight_bytes", "stm_stem_pack_weights_f32", "stm_stem_fused_f32", "stm_chain_tail_weight_bytes", "stm_chain_tail_weight_bytes_proj", "stm_chain_pack_tail_f32", "stm_chain_pack_tail_proj_f32", "stm_bottleneck_chain_f32", "stm_bottleneck_chain_proj_f32", "stm_deform_im2col_f32", "stm_deform_conv_workspace_bytes",
    "stm_deform_conv_fwd_f32", "stm_gemm_bias_f32", "stm_gemm_workspace_bytes", "stm_gemm_bias_ws_f32", "stm_fcb_ali_offsets_f32", "stm_corr_patch_f32", "stm_corr_patch_nhwc_f32",
    "stm_roi_align_avg_f32", "stm_decode_boxes_f32", "stm_generate_candidates_f32", "stm_cc_fast_nms_f32",
    "stm_detect_cc_workspace_bytes", "stm_detect_cc_f32", "stm_detect_cc_logits_f32", "stm_fast_nms_workspace_bytes", "stm_fast_nms_f32",
    "stm_jaccard_f32", "stm_lincomb_sigmoid_crop_f32", "stm_mask_iou_workspace_bytes", "stm_mask_iou_f32",
    "stm_bias_act_f32", "stm_mask_rle_workspace_bytes", "stm_mask_resize_rle_f32",
    "stm_conv_packed_weight_bytes", "stm_conv_pack_weights_f32",
    "stm_split_bf16_planes_f32", "stm_conv2d_planar_f32", "stm_conv_packed_weight_bytes_tiled",
    "stm_conv_pack_weights_tiled_f32", "stm_preprocess_u8_f32", "stm_head_assemble_f32", "stm_conv2d_planar_ws_f32", "stm_dcn_sample_planar_f32", "stm_conv_pack_weights_fmt_f32", "stm_split_planes_fmt_f32", "stm_dcn_sample_planar_fmt_f32", "stm_planar_set_range_flag", "stm_resize_bilinear_planes_f32", "stm_bias_relu_maxpool_planes_f32", "stm_roi_align_planes_f32", "stm_roi_align_planes_nhwc_f32", "stm_deform_sample_planar_f32", "stm_stem_rows_planes_f32", "stm_mask_iou_grouped_f32", "stm_cc_fast_nms_workspace_bytes", "stm_cc_fast_nms_ws_f32",
    "stm_gather_detections_f32", "stm_shift_rois_f32", "stm_shift_apply_f32", "stm_match_scores_f32", "stm_match_scores_embed_f32", "stm_gather_rows2", "stm_pack_tracked_f32", "stm_pack_tracked_bits_f32",
    "stm_lincomb_sigmoid_crop_bits_f32", "stm_mask_iou_bits_f32", "stm_split_planes_f16", "stm_conv_pack_weights_f16", "stm_conv2d_planar_f16", "stm_dcn_sample_planar_f16",
    "stm_deform_conv_fused_planar_supported", "stm_deform_conv_fused_planar_f32", "stm_fast_nms_batched_workspace_bytes", "stm_fast_nms_batched_f32", "stm_rle_strings_host",
]


class StmError(RuntimeError):
    pass


class DeformGeom(ctypes.Structure):
    _fields_ = [(n, c_i) for n in ("B", "C", "H", "W", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "dg", "Ho", "Wo")]


class ConvGeom(ctypes.Structure):
    _fields_ = ([(n, c_i) for n in ("B", "H", "W", "C", "Ho", "Wo", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "x_ld",
                                    "out_ld", "res_ld", "planes", "groups", "n_levels")] +
                [("lvl_start", c_i * 9), ("lvl_h", c_i * 8), ("lvl_w", c_i * 8),
                 ("x_plane_stride", c_l), ("out_plane_stride", c_l), ("res_plane_stride", c_l), ("x_np", c_i), ("out_np", c_i), ("res_np", c_i), ("group_cout", c_i * 8), ("fmt", c_i), ("out_scale", c_f), ("tile_n", c_i), ("out_fmt_plus1", c_i),
                 ("win_h", c_i), ("win_w", c_i), ("win_y0", c_i), ("win_x0", c_i)])


class ConvWindow(ctypes.Structure):
    _fields_ = [(n, c_i) for n in ("kh", "kw", "ph", "pw", "Ho", "Wo", "y0", "x0")]


class HeadLayout(ctypes.Structure):
    _fields_ = ([(n, c_i) for n in ("B", "K", "n_levels", "n_cls", "mask_dim", "embed_dim", "group_pad", "small_ld", "trk_ld")] +
                [("lvl_start", c_i * 8), ("lvl_hw", c_i * 8)])


def build(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StmError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback on the product path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.stm_last_error_string.restype = ctypes.c_char_p
        _lib.stm_version.restype = c_i
        for name in ("stm_deform_conv_workspace_bytes", "stm_gemm_workspace_bytes", "stm_mask_rle_workspace_bytes", "stm_detect_cc_workspace_bytes", "stm_fast_nms_workspace_bytes",
                     "stm_mask_iou_workspace_bytes", "stm_conv_packed_weight_bytes", "stm_conv_packed_weight_bytes_tiled", "stm_cc_fast_nms_workspace_bytes", "stm_conv_kxr_packed_bytes", "stm_stem_packed_weight_bytes", "stm_chain_tail_weight_bytes", "stm_chain_tail_weight_bytes_proj", "stm_fast_nms_batched_workspace_bytes"):
            getattr(_lib, name).restype = c_sz
        _lib.stm_struct_bytes.restype = c_sz
        _lib.stm_debug_reload_tunables.restype = None
        _lib.stm_debug_launch_count.restype = ctypes.c_longlong
        # this binding and the library must describe the same structs (a stale .so would read garbage past a shorter struct)
        if _lib.stm_version() != ABI_VERSION or _lib.stm_struct_bytes(0) != ctypes.sizeof(DeformGeom) or \
                _lib.stm_struct_bytes(1) != ctypes.sizeof(ConvGeom) or _lib.stm_struct_bytes(2) != ctypes.sizeof(ConvWindow) or \
                _lib.stm_struct_bytes(3) != ctypes.sizeof(HeadLayout):
            v = _lib.stm_version()
            _lib = None
            raise StmError(f"{LIB_PATH} has ABI version {v}, this binding was written for {ABI_VERSION} (or a struct size "
                           "differs): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        missing = [n for n in ABI_SYMBOLS if not hasattr(_lib, n)]
        if missing:
            _lib = None
            raise StmError(f"{LIB_PATH} lacks {', '.join(missing)}: rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().stm_last_error_string().decode(errors="replace")
        raise StmError(f"{what} failed with code {rc}: {msg}")
