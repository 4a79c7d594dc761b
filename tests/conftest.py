import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from nhd_amd import _lib
        return _lib.load().nhdfit_device_count() > 0
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest` without `-m "not gpu"` on a box without a gfx950 device: the GPU tests are skipped, not failed."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items and not _gpu_available():
        skip = pytest.mark.skip(reason="no gfx950 device / libnhdfit cannot open one")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference (only present in the build container)."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip("reference tree not present")
    r = ref_loader.load()
    return r
