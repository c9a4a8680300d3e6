import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--paths-off", default="", help="comma-separated names of causaldiffae_amd.ops.PATH_TOGGLES to switch off for the whole session "
                                                     "(tools/switch_matrix.sh: the predecessor of each fused path must stay green)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    off = [n for n in config.getoption("--paths-off").split(",") if n]
    if off:
        from causaldiffae_amd import ops
        g = vars(ops)
        for n in off:
            g[ops.PATH_TOGGLES[n]] = False


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        return cache[name]

    return load


@pytest.fixture(params=["f16x3", "fp32"])
def precision(request):
    """Run a GPU test under both arithmetic modes of the K-contiguous contractions (include/cdae.h)."""
    import causaldiffae_amd
    old = causaldiffae_amd.get_precision()
    causaldiffae_amd.set_precision(request.param)
    yield request.param
    causaldiffae_amd.set_precision(old)


class ExpectKernels:
    """with expect_kernels(convwin=1, convwin_dgrad=1): ... — the block must have launched at least that many kernels of each named
    profile family (include/cdae.h CDAE_PROF_FAMILIES), i.e. the test proves WHICH kernel produced the numbers it checks."""

    def __init__(self, **minimum):
        self.minimum, self.seen = minimum, None

    def __enter__(self):
        from causaldiffae_amd import _lib
        _lib.prof_enable(True)
        _lib.prof_read()
        return self

    def __exit__(self, et, ev, tb):
        from causaldiffae_amd import _lib
        self.seen = _lib.prof_read()
        _lib.prof_enable(False)
        if et is None:
            for fam, n in self.minimum.items():
                assert self.seen[fam]["launches"] >= n, (fam, {k: v["launches"] for k, v in self.seen.items()})
        return False


@pytest.fixture
def expect_kernels():
    return ExpectKernels
