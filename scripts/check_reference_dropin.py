#!/usr/bin/env python3
"""Build-container check of the drop-in claim: import the REFERENCE's own STMask.py with
stmask_amd/shims first on sys.path and verify that

  * `import mmcv` still is the installed mmcv (a stand-in package is created in a temp directory to play that role: this
    container has none) -- `mmcv.imread`, `mmcv.parallel.DataContainer`, `mmcv.load` resolve to it -- while `mmcv.ops.
    DeformConv2d / roi_align` are the MI355X implementations;
  * the reference's `STMask()` builds for the four benchmark configs, its DCN / DeformConv2d modules ARE the shim classes,
    and its state-dict keys and shapes equal those of stmask_amd.model.STMask (checkpoint compatibility);
  * the reference's call sites bind against the shim signatures: backbone.py:21-22 (DCN), Featurealign.py:27-31,72
    (DeformConv2d), track_to_segment_head.py:53-59 (spatial_correlation_sample), :86 (roi_align).

Reads /root/reference: never shipped to or run on the GPU box.  No GPU needed (nothing is executed on tensors).
"""
import collections
import collections.abc
import inspect
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("STMASK_REFERENCE", "/root/reference")
CONFIGS = ["STMask_plus_resnet50_config", "STMask_plus_resnet50_ada_config", "STMask_plus_resnet50_ali_config",
           "STMask_plus_base_ali_config"]


def make_installed_mmcv(base):
    """A stand-in for the user's installed mmcv (image / file helpers, mmcv.parallel, and an mmcv.ops that must NOT win)."""
    pkg = os.path.join(base, "mmcv")
    os.makedirs(os.path.join(pkg, "parallel"))
    os.makedirs(os.path.join(pkg, "runner"))
    os.makedirs(os.path.join(pkg, "ops"))
    with open(os.path.join(pkg, "__init__.py"), "w") as f:
        f.write("from .image import imread, imresize, impad_to_multiple, imflip\nfrom .misc import is_str, is_list_of, load, dump\n"
                "__version__ = 'installed-standin'\n")
    with open(os.path.join(pkg, "image.py"), "w") as f:
        f.write("def imread(p, *a, **k): return ('installed-imread', p)\ndef imresize(*a, **k): return 'installed-imresize'\n"
                "def impad_to_multiple(*a, **k): return 'installed-impad'\ndef imflip(*a, **k): return 'installed-imflip'\n")
    with open(os.path.join(pkg, "misc.py"), "w") as f:
        f.write("def is_str(x): return isinstance(x, str)\ndef is_list_of(s, t): return isinstance(s, list) and all(isinstance(i, t) for i in s)\n"
                "def load(p, *a, **k): return ('installed-load', p)\ndef dump(o, p, *a, **k): return 'installed-dump'\n")
    with open(os.path.join(pkg, "parallel", "__init__.py"), "w") as f:
        f.write("class DataContainer:\n    def __init__(self, data, **kw): self.data = data\ndef collate(batch, samples_per_gpu=1): return batch\n")
    with open(os.path.join(pkg, "runner", "__init__.py"), "w") as f:
        f.write("def get_dist_info(): return 0, 1\ndef obj_from_dict(info, parent=None, default_args=None): raise NotImplementedError\n")
    with open(os.path.join(pkg, "ops", "__init__.py"), "w") as f:
        f.write("raise ImportError('installed mmcv.ops: CUDA extension not available (this must be overridden by the shim)')\n")


def stub_absent_third_party():
    """Packages the reference imports that this container lacks and that are NOT part of the deliverable."""
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return None

    def stub(name, **kw):
        m = _Any(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    for n in ["cocoapi", "cocoapi.PythonAPI", "cocoapi.PythonAPI.pycocotools", "pycocotools", "pycocotools.mask", "cv2", "pyximport"]:
        stub(n)
    sys.modules["pyximport"].install = lambda *a, **k: None
    stub("cocoapi.PythonAPI.pycocotools.ytvos", YTVOS=object)
    stub("cocoapi.PythonAPI.pycocotools.ytvoseval", YTVOSeval=object)
    stub("pycocotools.coco", COCO=object)
    stub("pycocotools.cocoeval", COCOeval=object)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    stub("utils.cython_nms", nms=None)
    collections.Sequence = collections.abc.Sequence
    np.int, np.float = int, float
    torch.cuda.current_device = lambda: "cpu"       # STMask.py:15 calls it at import


def main():
    if not os.path.isdir(REF):
        print("reference not present: nothing to check")
        return 0
    tmp = tempfile.mkdtemp(prefix="installed_mmcv_")
    make_installed_mmcv(tmp)
    for m in [k for k in sys.modules if k == "mmcv" or k.startswith("mmcv.") or k in ("dcn_v2", "spatial_correlation_sampler")]:
        del sys.modules[m]
    sys.path[:0] = [os.path.join(ROOT, "stmask_amd", "shims"), ROOT, REF, tmp]   # shims first; tmp = "site-packages"
    stub_absent_third_party()

    import mmcv
    import mmcv.ops
    from mmcv.parallel import DataContainer
    from stmask_amd import dcn_v2 as amd_dcn, mmcv_ops as amd_mmcv, spatial_correlation_sampler as amd_corr
    assert mmcv.__version__ == "installed-standin" and mmcv.imread("x.jpg") == ("installed-imread", "x.jpg"), "mmcv was shadowed"
    assert mmcv.impad_to_multiple() == "installed-impad" and mmcv.load("a.json")[0] == "installed-load"
    assert DataContainer(1).data == 1 and mmcv.STMASK_AMD_OPS
    assert mmcv.ops.DeformConv2d is amd_mmcv.DeformConv2d and mmcv.ops.roi_align is amd_mmcv.roi_align
    print("mmcv: installed package intact, mmcv.ops served by stmask_amd")

    from datasets.config import cfg, set_cfg
    import backbone as ref_backbone
    from layers.modules import Featurealign as ref_fa, track_to_segment_head as ref_t2s
    assert ref_backbone.DCN is amd_dcn.DCN and ref_fa.DeformConv2d is amd_mmcv.DeformConv2d
    assert ref_t2s.roi_align is amd_mmcv.roi_align and ref_t2s.spatial_correlation_sample is amd_corr.spatial_correlation_sample

    # call-site signatures (bind only: no tensors are touched)
    inspect.signature(amd_dcn.DCN.__init__).bind(None, 128, 128, kernel_size=3, stride=2, padding=1, dilation=1, deformable_groups=1)   # backbone.py:21-22
    inspect.signature(amd_dcn.DCN.forward).bind(None, "x")                                                                          # backbone.py:45
    inspect.signature(amd_mmcv.DeformConv2d.__init__).bind(None, 256, 256, kernel_size=(3, 5), padding=(1, 2), deform_groups=1)       # Featurealign.py:27-31
    inspect.signature(amd_mmcv.DeformConv2d.forward).bind(None, "x", "offset")                                                      # Featurealign.py:72
    inspect.signature(amd_mmcv.roi_align).bind("feat", "rois", 7)                                                                   # track_to_segment_head.py:86
    inspect.signature(amd_corr.spatial_correlation_sample).bind("x1", "x2", kernel_size=1, patch_size=11, stride=1, padding=0,
                                                                dilation_patch=1)                                                   # :53-59
    print("call sites bind against the shim signatures")

    import STMask as ref_stmask
    from stmask_amd.config import get_cfg
    from stmask_amd.model import STMask as AmdSTMask
    for name in CONFIGS:
        set_cfg(name)
        ref_net = ref_stmask.STMask()
        ours = AmdSTMask(get_cfg(name))
        rs, os_ = ref_net.state_dict(), ours.state_dict()
        assert list(rs.keys()) == list(os_.keys()), (name, set(rs) ^ set(os_))
        assert all(tuple(rs[k].shape) == tuple(os_[k].shape) for k in rs), name
        n_dcn = sum(isinstance(m, amd_dcn.DCN) for m in ref_net.modules())
        n_fcb = sum(isinstance(m, amd_mmcv.DeformConv2d) for m in ref_net.modules())
        assert n_dcn == sum(isinstance(m, amd_dcn.DCN) for m in ours.modules()) and n_dcn in (7, 11), (name, n_dcn)
        assert n_fcb == (3 if cfg.use_dcn_class else 0), (name, n_fcb)
        print(f"{name}: reference STMask() built on the shims, {len(rs)} state-dict entries identical, {n_dcn} DCN, {n_fcb} DeformConv2d")
    return 0


if __name__ == "__main__":
    sys.exit(main())
