"""Host logic of TemporalNet's border classes (stmask_amd.planar.border_windows): emulated on the CPU with torch -- every class as a plain
VALID convolution with its sub-kernel over its window of the map, written in place -- the nine classes rebuild the 3x3 / pad-1
convolution of the reference's TemporalNet layers (track_to_segment_head.py:10-37) exactly, cover every output pixel once, and multiply
361 of the 441 tap-pixels of a 7x7 map.  (The GPU form is tests/test_gpu_conv.py::test_conv_window_launches_equal_the_padded_convolution.)"""
import pytest
import torch
import torch.nn.functional as F

from stmask_amd import _lib
from stmask_amd.planar import _BORDER_CLASSES, border_windows


@pytest.mark.parametrize("hw", [(7, 7), (5, 9), (3, 3), (3, 8), (12, 4)])
def test_border_classes_rebuild_the_padded_convolution(hw):
    h, w = hw
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 6, h, w, generator=g, dtype=torch.float64)
    wt = torch.randn(5, 6, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, wt, padding=1)
    out = torch.full_like(ref, float("nan"))
    count = torch.zeros(h, w, dtype=torch.int64)
    macs = 0
    for ci, (kh, kw, ph, pw, ho, wo, y0, x0) in border_windows(h, w):
        _, _, k0y, k1y, k0x, k1x = _BORDER_CLASSES[ci]
        assert (kh, kw) == (k1y - k0y, k1x - k0x) and ph <= 0 and pw <= 0
        sub = wt[:, :, k0y:k1y, k0x:k1x]
        # output (oy, ox) of the window reads input (oy - ph + ky', ox - pw + kx'): a VALID convolution over the crop that starts at (-ph, -pw)
        crop = x[:, :, -ph:-ph + ho + kh - 1, -pw:-pw + wo + kw - 1]
        assert crop.shape[2:] == (ho + kh - 1, wo + kw - 1)          # every tap of every output of the class lies inside the map
        out[:, :, y0:y0 + ho, x0:x0 + wo] = F.conv2d(crop, sub)
        count[y0:y0 + ho, x0:x0 + wo] += 1
        macs += ho * wo * kh * kw
    assert (count == 1).all()
    assert torch.allclose(out, ref, rtol=0, atol=1e-12)
    valid = sum(1 for y in range(h) for x_ in range(w) for ky in range(3) for kx in range(3) if 0 <= y + ky - 1 < h and 0 <= x_ + kx - 1 < w)
    assert macs == valid
    if hw == (7, 7):
        assert macs == 361 and h * w * 9 == 441


def test_window_struct_matches_the_header():
    import ctypes
    import re
    from conftest import ROOT
    header = open(f"{ROOT}/include/stmask_hip.h").read()
    fields = re.search(r"typedef struct stm_conv_window \{ int ([^;]+); \} stm_conv_window;", header).group(1).replace(" ", "").split(",")
    assert fields == [n for n, _ in _lib.ConvWindow._fields_] and ctypes.sizeof(_lib.ConvWindow) == 4 * len(fields)
    # stm_conv_geom's window fields close the struct (ABI 3)
    assert [n for n, _ in _lib.ConvGeom._fields_][-4:] == ["win_h", "win_w", "win_y0", "win_x0"]
