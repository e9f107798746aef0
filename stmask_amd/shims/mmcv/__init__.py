"""`mmcv` import name for the reference's hot path WITHOUT shadowing an installed mmcv.

With this directory's parent first on PYTHONPATH, `import mmcv` lands here.  If another `mmcv` package exists further down
sys.path (the reference needs it for `mmcv.imread / imresize / impad_to_multiple` (eval.py:704-715),
`mmcv.parallel.DataContainer` (datasets/custom.py:5), `mmcv.load / dump`, ...), this module BECOMES that package: its
`__init__.py` is executed in this namespace and `__path__` is [this directory, the real package directory], so every
submodule of the real mmcv imports as before and only `mmcv.ops` -- the CUDA extension the reference uses for
`DeformConv2d` and `roi_align` (Featurealign.py:3, track_to_segment_head.py:6) -- resolves to the MI355X kernels
(stmask_amd/shims/mmcv/ops).  With no mmcv installed the two helpers the reference's eval path needs at import time
(datasets/utils.py, datasets/ytvos.py:64) are provided here, and nothing else is pretended.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIM_ROOT = os.path.dirname(_HERE)


def _find_installed():
    for entry in sys.path:
        base = os.path.abspath(entry or os.getcwd())
        if base == _SHIM_ROOT:
            continue
        init = os.path.join(base, "mmcv", "__init__.py")
        if os.path.isfile(init):
            return init
    return None


_REAL_INIT = _find_installed()

if _REAL_INIT is not None:
    __path__ = [_HERE, os.path.dirname(_REAL_INIT)]    # ops/ from here, everything else from the installed package
    __file__ = _REAL_INIT
    with open(_REAL_INIT, "rb") as _fh:
        exec(compile(_fh.read(), _REAL_INIT, "exec"), globals())
    STMASK_AMD_OPS = True                               # marker: mmcv.ops is served by stmask_amd
else:
    def is_str(x):
        return isinstance(x, str)

    def is_list_of(seq, expected_type):
        return isinstance(seq, list) and all(isinstance(i, expected_type) for i in seq)

    def __getattr__(name):
        raise AttributeError(
            f"mmcv.{name}: no mmcv package is installed; stmask_amd/shims only supplies mmcv.ops.DeformConv2d / roi_align "
            "(and is_str / is_list_of).  Install mmcv for the reference's image / file helpers.")
