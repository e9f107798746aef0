"""The drop-in boundary: the import shims must not shadow an installed mmcv, and -- in the build
container, where /root/reference exists -- the reference's own STMask.py must import and build on them with identical
state-dict keys (scripts/check_reference_dropin.py).  CPU only; the GPU box has no reference and skips the second test."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SHIMS = os.path.join(ROOT, "stmask_amd", "shims")


def _run(code, extra_path=()):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([SHIMS, ROOT, *extra_path]))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_shim_extends_an_installed_mmcv_instead_of_shadowing_it(tmp_path):
    pkg = tmp_path / "mmcv"
    (pkg / "parallel").mkdir(parents=True)
    (pkg / "ops").mkdir()
    (pkg / "__init__.py").write_text("from .image import imread\n__version__ = '9.9'\n")
    (pkg / "image.py").write_text("def imread(p): return 'real:' + p\n")
    (pkg / "parallel" / "__init__.py").write_text("class DataContainer: pass\n")
    (pkg / "ops" / "__init__.py").write_text("raise ImportError('CUDA extension missing')\n")
    p = _run("import mmcv, mmcv.ops\nfrom mmcv.parallel import DataContainer\nfrom mmcv.ops import DeformConv2d, roi_align\n"
             "import stmask_amd.mmcv_ops as m\nassert mmcv.__version__ == '9.9' and mmcv.imread('a') == 'real:a'\n"
             "assert DeformConv2d is m.DeformConv2d and roi_align is m.roi_align\nprint('ok')", [str(tmp_path)])
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-1500:]


def test_shim_without_installed_mmcv_pretends_nothing():
    p = _run("import mmcv\nfrom mmcv.ops import DeformConv2d\nassert mmcv.is_str('a') and mmcv.is_list_of([1], int)\n"
             "try:\n    mmcv.imread\nexcept AttributeError as e:\n    print('ok', 'no mmcv package is installed' in str(e))")
    assert p.returncode == 0 and "ok True" in p.stdout, p.stderr[-1500:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="build container only: needs the reference checkout")
def test_reference_stmask_imports_and_builds_on_the_shims():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_reference_dropin.py")], capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("state-dict entries identical") == 4 and "installed package intact" in p.stdout
