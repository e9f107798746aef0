import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden_postproc():
    return load_golden("postproc.npz")


@pytest.fixture(scope="session")
def golden_priors():
    return load_golden("priors.npz")


@pytest.fixture(scope="session")
def golden_fcb_ali():
    return load_golden("fcb_ali.npz")


def ulp_diff(a, b):
    """Distance in units-in-the-last-place between two fp32 tensors (same sign assumed or tiny)."""
    ia = a.contiguous().view(torch.int32).to(torch.int64)
    ib = b.contiguous().view(torch.int32).to(torch.int64)
    ia = torch.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = torch.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return (ia - ib).abs()


@pytest.fixture
def tunables(monkeypatch):
    """Set STM_* A/B switches for one test.  The library reads them once per process; stm_debug_reload_tunables() makes the
    next launch read them again (and once more when the test's environment is restored)."""
    from stmask_amd import _lib

    def reload():
        _lib.lib().stm_debug_reload_tunables()

    class T:
        def set(self, **env):
            for k, v in env.items():
                monkeypatch.setenv(k, str(v))
            reload()

        def clear(self, *names):
            for k in names:
                monkeypatch.delenv(k, raising=False)
            reload()

    yield T()
    monkeypatch.undo()
    reload()
