"""Output stage on the MI355X -- mirror of the reference's layers/output_utils.py:16-133 (``postprocess_ytbvis``), the
first "next" row after the hot path: score filter, un-pad, bilinear up-sampling of the soft masks to
the original frame size, threshold, COCO run-length encoding, box rescaling.

The reference brings every full-resolution mask to the host (``masks[i].cpu()``) and encodes it with pycocotools; here
resize + threshold + run extraction run on the device (``stm_mask_resize_rle_f32``) and only the run lengths cross PCIe.
The 5-bit string packing of COCO RLE (pycocotools maskApi.c ``rleToString``) is a few hundred bytes per mask and stays
on the host.
"""
import torch

from . import ops
from .layers.box_utils import center_size, sanitize_coordinates

_SKIP = ("proto", "bbox_idx", "priors", "embed_vectors", "box_shift")


def rle_counts_to_string(counts):
    """COCO compressed RLE string of a list of run lengths (maskApi.c rleToString)."""
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def encode_masks(masks_soft, crop_h, crop_w, out_h, out_w, thr=0.5, max_runs=4096):
    """[n,mh,mw] soft masks -> list of COCO RLE dicts {'size': [h, w], 'counts': bytes} (device resize + RLE)."""
    n = masks_soft.shape[0]
    counts, n_runs = ops.mask_resize_rle(masks_soft, crop_h, crop_w, out_h, out_w, thr, max_runs)
    nr = n_runs.cpu()
    if n and int(nr.max()) > max_runs:  # a very ragged mask: redo with room for every run
        return encode_masks(masks_soft, crop_h, crop_w, out_h, out_w, thr, int(nr.max()))
    width = int(nr.max()) if n else 0
    if n == 0:
        return []
    host = counts[:, :width].contiguous().cpu()  # the ONLY mask bytes that cross PCIe: run lengths
    strings = rle_strings(host, nr)
    return [{"size": [out_h, out_w], "counts": strings[i]} for i in range(n)]


def rle_strings(counts_host, n_runs_host):
    """COCO RLE strings of the rows of counts_host [n, width] int32 (CPU, contiguous; row i holds n_runs_host[i] runs): the library's host-side
    packer (stm_rle_strings_host) -- the Python form above costs ~0.2 ms per mask, more than the GPU spends on a whole step."""
    import ctypes
    import numpy as np
    from . import _lib
    n, width = counts_host.shape
    c = np.ascontiguousarray(counts_host.numpy().astype(np.uint32, copy=False))
    nr = np.ascontiguousarray(n_runs_host.numpy().astype(np.int32, copy=False))
    out_ld = max(16, 5 * width)                              # a run takes at most 7 characters, typically 1-3; retried below if it does not fit
    while True:
        out = np.empty((n, out_ld), dtype=np.uint8)
        lens = np.empty(n, dtype=np.int32)
        rc = _lib.lib().stm_rle_strings_host(c.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(width), nr.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n),
                                             out.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(out_ld), lens.ctypes.data_as(ctypes.c_void_p))
        if rc == 0:
            break
        if int(lens.max()) <= out_ld:
            _lib.check(rc, "stm_rle_strings_host")
        out_ld = int(lens.max())
    return [out[i, :lens[i]].tobytes() for i in range(n)]


def postprocess_ytbvis(det_output, img_meta, interpolation_mode="bilinear", display_mask=False, score_threshold=0,
                       preserve_aspect_ratio=True):
    """Same contract as the reference: returns the detection dict with 'segm' (list of COCO RLE dicts, or the binary
    masks on the device when display_mask) and integer pixel 'box'.  `preserve_aspect_ratio` is what eval.py sets on the
    global cfg before calling (eval.py:639)."""
    if interpolation_mode != "bilinear":
        raise NotImplementedError("the reference only ever calls this with bilinear interpolation")
    dets = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in det_output["detection"].items()}
    ori_h, ori_w = img_meta["ori_shape"][:2]
    img_h, img_w = img_meta["img_shape"][:2]
    pad_h, pad_w = img_meta["pad_shape"][:2]
    s_w, s_h = img_w / pad_w, img_h / pad_h
    if dets["box"].nelement() == 0:
        dets["segm"] = []
        return dets

    def keep_rows(keep):
        idx = torch.nonzero(keep).view(-1)
        for k in dets:
            if k not in _SKIP and torch.is_tensor(dets[k]) and dets[k].dim() > 0 and dets[k].shape[0] == keep.shape[0]:
                dets[k] = dets[k].index_select(0, idx)

    if score_threshold > 0:
        keep_rows(dets["score"] > score_threshold)
    if preserve_aspect_ratio and dets["score"].nelement() != 0:
        c = center_size(dets["box"])
        keep_rows(((c[:, 0] > s_w).int() + (c[:, 1] > s_h).int()) < 1)
    if dets["score"].size(0) == 0:
        dets["segm"] = []
        return dets

    masks, boxes = dets["mask"], dets["box"]
    crop_h, crop_w = int(s_h * masks.size(1)), int(s_w * masks.size(2))
    out_h, out_w = (ori_h, ori_w) if preserve_aspect_ratio else (img_h, img_w)
    if display_mask:
        up = torch.nn.functional.interpolate(masks[None, :, :crop_h, :crop_w], (out_h, out_w), mode="bilinear",
                                             align_corners=False)[0]
        dets["segm"] = up.gt_(0.5)
    else:
        dets["segm"] = encode_masks(masks, crop_h, crop_w, out_h, out_w)
    boxes = boxes.clone()
    boxes[:, 0::2] = boxes[:, 0::2] / s_w
    boxes[:, 1::2] = boxes[:, 1::2] / s_h
    boxes[:, 0], boxes[:, 2] = sanitize_coordinates(boxes[:, 0], boxes[:, 2], out_w, cast=False)
    boxes[:, 1], boxes[:, 3] = sanitize_coordinates(boxes[:, 1], boxes[:, 3], out_h, cast=False)
    dets["box"] = boxes.long()
    return dets
