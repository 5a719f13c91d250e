import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


CONV_KERNELS = {"default": 0, "tile": 1, "wave": 2, "band": 3, "cols": 4, "rounds": 5}


def conv_families():
    """the forward / dX kernel families of the loaded library: tile (one offset per phase), wave-private, the tile kernel's
    two-offset rounds; the row-band and column-owner families exist only in experiment builds
    (make -C sparseeventid_amd/csrc -B EXTRA=-DSEI_EXPERIMENTS): the shipped library refuses to pin them"""
    from sparseeventid_amd._lib import lib
    fams = ["tile", "wave", "rounds"]
    for extra in ("band", "cols"):
        if lib().sei_debug_force_conv_kernel(CONV_KERNELS[extra]) == 0:
            fams.append(extra)
    lib().sei_debug_force_conv_kernel(0)
    return tuple(fams)


@pytest.fixture
def force_conv_kernel():
    """force_conv_kernel("tile" | "wave" | "band" | "default"): pins the forward / dX kernel family
    (sei_debug_force_conv_kernel) for the test and restores the plan's own choice afterwards.  "band" skips the test on
    the shipped library (no row-band kernels there)."""
    from sparseeventid_amd._lib import lib

    def force(kind):
        rc = lib().sei_debug_force_conv_kernel(CONV_KERNELS[kind])
        if rc != 0 and kind in ("band", "cols"):
            pytest.skip("library built without -DSEI_EXPERIMENTS: no row-band / column-owner kernels")
        assert rc == 0
    yield force
    lib().sei_debug_force_conv_kernel(0)
