import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the built libraries are git-ignored: a fresh checkout builds them once (hipcc cross-compiles
    # gfx950 without a GPU; ~20 s), exactly what __graft_entry__.build() does
    import glob
    need = [os.path.join(ROOT, "pnec_amd", "libpnec_hip.so"), os.path.join(ROOT, "pnec_amd", "libpnec_host.so"),
            os.path.join(ROOT, "oracle", "libpnec_oracle.so")]
    if not all(os.path.exists(f) for f in need) or not glob.glob(os.path.join(ROOT, "pnec_amd", "pypnec*.so")):
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing in
    hipSetDevice; on the GPU box nothing is skipped (and `-m gpu` runs exactly those)."""
    from pnec_amd import capi
    if capi.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (pnec_hip_device_count() == 0 here)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libpnec_oracle.so if needed."""
    from oracle import pnec_oracle as po
    po.lib()
    return po
