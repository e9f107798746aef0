"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares,
and the product path never touches the oracle."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT
from stmask_amd import _lib, ops


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "stmask_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _header_symbols() == sorted(_lib.ABI_SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/stmask_hip.h but not exported"
    lib.stm_version.restype = ctypes.c_int
    assert lib.stm_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "stmask_hip.h")).read()
    assert int(re.search(r"#define STM_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION


def test_struct_layouts_of_binding_and_library_agree():
    """A client built against another header revision passes a struct of another size: stm_struct_bytes lets it find out
    (stm_conv_geom grew a trailing field in round 2 while the version stayed 1)."""
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.stm_struct_bytes.restype = ctypes.c_size_t
    assert lib.stm_struct_bytes(0) == ctypes.sizeof(_lib.DeformGeom)
    assert lib.stm_struct_bytes(1) == ctypes.sizeof(_lib.ConvGeom)
    assert lib.stm_struct_bytes(2) == ctypes.sizeof(_lib.ConvWindow) == 32         # passed as an array: the size is the stride
    assert lib.stm_struct_bytes(3) == ctypes.sizeof(_lib.HeadLayout)
    assert lib.stm_struct_bytes(7) == 0


def test_library_targets_gfx950():
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data


def test_argument_errors_are_reported_not_thrown():
    """Null pointers / bad sizes return STM_E* codes with a message (no compute, no GPU needed)."""
    lib = _lib.lib()
    rc = lib.stm_decode_boxes_f32(None, None, None, ctypes.c_int64(5), None)
    assert rc == -2 and b"non-NULL" in lib.stm_last_error_string()
    rc = lib.stm_cc_fast_nms_f32(None, None, None, 10, 41, None, ctypes.c_float(0.5), 4096, 1, None, None, None, None,
                                 None, None)
    assert rc in (-1, -2)
    g = _lib.DeformGeom(1, 6, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, 4, 8, 8)  # C % dg != 0
    rc = lib.stm_deform_im2col_f32(None, None, ctypes.c_int64(0), None, ctypes.c_int64(0), 0, None, ctypes.byref(g), 0,
                                   None)
    assert rc == -1 and b"deform groups" in lib.stm_last_error_string()
    g = _lib.DeformGeom(1, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, 1, 9, 8)  # wrong Ho
    rc = lib.stm_deform_im2col_f32(None, None, ctypes.c_int64(0), None, ctypes.c_int64(0), 0, None, ctypes.byref(g), 0,
                                   None)
    assert rc == -1 and b"conv arithmetic" in lib.stm_last_error_string()


def test_cpu_tensors_fail_loudly():
    """No silent CPU fallback on the product path."""
    with pytest.raises(_lib.StmError):
        ops.decode(torch.zeros(4, 4), torch.zeros(4, 4))
    with pytest.raises(_lib.StmError):
        ops.corr_patch(torch.zeros(1, 4, 8, 8), torch.zeros(1, 4, 8, 8), 11)


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "stmask_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", src, flags=re.M), f"{f} imports the oracle"
                assert "libstm_oracle" not in src, f"{f} references the oracle library"


def test_planar_conv_kernels_do_not_spill():
    """hipcc's resource report for csrc/conv_bf16x.hip: no instantiation of the planar convolution kernels may use scratch
    (a register spill in the MFMA loop costs 2x: seen once when the 128-wide body was duplicated)."""
    import re
    import subprocess
    src = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc", "conv_bf16x.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize",
           "--cuda-device-only", "-c", src, "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600).stderr
    blocks = re.split(r"remark: [^\n]*Function Name: ", out)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "conv_planar_kernel" not in name and "conv_planar_kx_kernel" not in name:
            continue
        seen += 1
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        vgprs = int(re.search(r" VGPRs: (\d+)", b).group(1))
        assert scratch == 0 and vgprs <= 256, (name, scratch, vgprs)
    assert seen >= 8
