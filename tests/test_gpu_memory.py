"""GPU: library-owned device memory (hash tables, rulebooks) is released with each batch -- no growth across
training steps (a Metadata <-> Rulebook reference cycle once kept every batch alive until the cyclic GC ran)."""
import ctypes
import gc

import pytest
import torch

from tests.util import blob_cloud

pytestmark = pytest.mark.gpu


def _in_use():
    from sparseeventid_amd import _lib
    held, used = ctypes.c_int64(), ctypes.c_int64()
    _lib.lib().sei_pool_stats(ctypes.byref(held), ctypes.byref(used))
    return used.value


def test_no_device_memory_growth_across_steps():
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import Config, Encoder
    gc.collect()
    gc.disable()                      # the engine must not depend on the cyclic collector
    try:
        enc, head = build_networks(Config(encoder=Encoder(depth=2, n_initial_filters=16, n_output_filters=16,
                                                          blocks_per_layer=1)), [1, 32, 32, 32], {"a": 2})
        enc.cuda(); head.cuda()
        usage = []
        for step in range(6):
            coords, feats = blob_cloud(100 + step, 2, [32, 32, 32], 300 + 10 * step, 1)
            out = head(enc((torch.from_numpy(coords), torch.from_numpy(feats), 2)))
            out["a"].square().sum().backward()
            del out
            torch.cuda.synchronize()
            usage.append(_in_use())
        assert usage[-1] <= usage[1] * 1.5 + (1 << 20), usage   # flat after the first steps
        assert usage[-1] < 64 << 20
    finally:
        gc.enable()
