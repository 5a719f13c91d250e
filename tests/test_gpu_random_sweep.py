"""Seeded random sweep (GPU vs oracle): grid shape, events per batch (some empty), sites per event, duplicates,
channel counts, filter shapes and strides drawn at random -- the combinations the hand-written cases do not list.
Integer work (rows, rule lists, output sites) is compared exactly, fp32 within 1e-4 relative."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import blob_cloud, random_cloud, rel_err

pytestmark = pytest.mark.gpu
CHANNELS = [1, 3, 16, 32, 48, 64, 80, 96, 112, 128]
FILTERS = [[3, 3, 3], [1, 3, 3], [5, 5, 5], [1, 1, 1], [3, 1, 3], [1, 5, 5]]


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    sp = [int(v) for v in rng.integers(5, 28, 3)]
    B = int(rng.integers(1, 5))
    n_per = [int(v) for v in rng.integers(0, 260, B)]
    if sum(n_per) == 0:
        n_per[0] = 17
    n_per = [min(n, sp[0] * sp[1] * sp[2]) for n in n_per]
    if seed % 2:
        coords, _ = random_cloud(seed, B, sp, n_per, 1, dup_frac=0.2 if seed % 4 == 1 else 0.0)
    else:
        # clustered sites; blob_cloud takes one count for all events
        coords, _ = blob_cloud(seed, B, sp, max(8, min(max(n_per), sp[0] * sp[1] * sp[2] // 3)), 1)
    return rng, sp, B, coords


@pytest.mark.parametrize("seed", range(12))
def test_random_submanifold_conv(seed):
    import sparseeventid_amd.scn as scn
    rng, sp, B, coords = _case(seed)
    ci, co = int(rng.choice(CHANNELS)), int(rng.choice(CHANNELS))
    filt = FILTERS[int(rng.integers(0, len(FILTERS)))]
    bias = bool(rng.integers(0, 2))
    feats = rng.standard_normal((len(coords), ci)).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, B))
    conv = scn.SubmanifoldConvolution(3, ci, co, filt, bias).cuda()
    if bias:
        with torch.no_grad():
            conv.bias.normal_()
    out = conv(t)
    fo = torch.from_numpy(feats).requires_grad_(True)
    W = conv.weight.detach().cpu().requires_grad_(True)
    b = conv.bias.detach().cpu().requires_grad_(True) if bias else None
    o = scn_cpu.input_layer(coords, fo, B, sp)
    assert np.array_equal(t.get_spatial_locations().numpy(), o.coords.astype(np.int64))
    rb = t.metadata.subm_rulebook(sp, filt)
    start, ri, ro = rb.export("cuda")
    ostart, ori, oro = scn_cpu.subm_rules(o.coords, sp, filt)
    assert list(start) == [int(v) for v in ostart]
    assert np.array_equal(ri.cpu().numpy(), ori) and np.array_equal(ro.cpu().numpy(), oro)
    oo = scn_cpu.subm_conv(o, W, b, filt)
    g = torch.randn(oo.features.shape, generator=torch.Generator().manual_seed(seed))
    out.features.backward(g.cuda())
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4
    assert rel_err(f.grad, fo.grad) < 1e-4
    if bias:
        assert rel_err(conv.bias.grad, b.grad) < 1e-4


@pytest.mark.parametrize("seed", range(12, 20))
def test_random_strided_conv(seed):
    import sparseeventid_amd.scn as scn
    rng, sp, B, coords = _case(seed)
    ci, co = int(rng.choice(CHANNELS)), int(rng.choice(CHANNELS))
    size = [int(v) for v in rng.choice([1, 2, 2, 3], 3)]
    stride = [1 if s == 1 else int(rng.integers(1, 3)) + (1 if s == 3 and rng.integers(0, 2) else 0) for s in size]
    stride = [min(st, s) if s > 1 else 1 for st, s in zip(stride, size)]
    feats = rng.standard_normal((len(coords), ci)).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, B))
    conv = scn.Convolution(3, ci, co, size, stride, False).cuda()
    out = conv(t)
    fo = torch.from_numpy(feats).requires_grad_(True)
    W = conv.weight.detach().cpu().requires_grad_(True)
    o = scn_cpu.input_layer(coords, fo, B, sp)
    oo = scn_cpu.strided_conv(o, W, None, size, stride)
    assert [int(v) for v in out.spatial_size] == [int(v) for v in oo.spatial_size]
    assert np.array_equal(out.get_spatial_locations().numpy(), oo.coords.astype(np.int64))
    g = torch.randn(oo.features.shape, generator=torch.Generator().manual_seed(seed))
    out.features.backward(g.cuda())
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4
    assert rel_err(f.grad, fo.grad) < 1e-4
