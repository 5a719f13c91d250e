"""GPU parity on the edge cases of the domain: tile-boundary row counts, large coordinates / batch indices,
heavy duplication, 125-offset filters on the MFMA path, single-site and single-event inputs."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _line_cloud(n, spatial, b=0, start=(0, 0, 0)):
    """n distinct sites along a diagonal-ish walk (dense neighbourhoods, exact count)."""
    pts, p = [], np.array(start)
    rng = np.random.default_rng(n)
    seen = set()
    while len(pts) < n:
        p = np.clip(p + rng.integers(0, 2, 3), 0, np.array(spatial) - 1)
        t = tuple(int(v) for v in p)
        if t not in seen:
            seen.add(t); pts.append(t + (b,))
        if (p == np.array(spatial) - 1).all():
            p = rng.integers(0, spatial)
    return np.array(pts, dtype=np.float64)


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257])
def test_row_counts_around_tile_boundaries(n):
    import sparseeventid_amd.scn as scn
    sp = [300, 300, 300]
    coords = _line_cloud(n, sp)
    feats = np.random.default_rng(1).standard_normal((n, 16)).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, 1))
    conv = scn.SubmanifoldConvolution(3, 16, 32, 3, True).cuda()
    down = scn.Convolution(3, 32, 16, 2, 2, False).cuda()
    out = down(conv(t))
    out.features.sum().backward()
    fo = torch.from_numpy(feats).requires_grad_(True)
    o = scn_cpu.input_layer(coords, fo, 1, sp)
    W, b, W2 = (conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True),
                down.weight.detach().cpu().requires_grad_(True))
    oo = scn_cpu.strided_conv(scn_cpu.subm_conv(o, W, b, [3, 3, 3]), W2, None, [2, 2, 2], [2, 2, 2])
    oo.features.sum().backward()
    assert np.array_equal(out.get_spatial_locations().numpy(), oo.coords.astype(np.int64))
    assert rel_err(out.features, oo.features) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4 and rel_err(down.weight.grad, W2.grad) < 1e-4
    assert rel_err(f.grad, fo.grad) < 1e-4


def test_large_coordinates_and_many_small_events():
    import sparseeventid_amd.scn as scn
    sp, B = [65535, 40000, 65535], 300
    rng = np.random.default_rng(3)
    rows = []
    for b in range(B):                              # a few clustered sites per event, some events empty
        if b % 7 == 0:
            continue
        base = np.array([rng.integers(0, 65533), rng.integers(0, 39998), rng.integers(0, 65533)])
        for d in range(rng.integers(1, 6)):
            rows.append((base[0] + d % 2, base[1] + (d // 2) % 2, base[2] + d // 4, b))
    rows.append((65534, 39999, 65534, B - 1))       # the very last cell of the last event
    coords = np.unique(np.array(rows, dtype=np.float64), axis=0)
    coords = coords[np.argsort(coords[:, 3], kind="stable")]
    feats = rng.standard_normal((len(coords), 16)).astype(np.float32)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), torch.from_numpy(feats).cuda(), B))
    conv = scn.SubmanifoldConvolution(3, 16, 16, 3, False).cuda()
    out = conv(t)
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    oo = scn_cpu.subm_conv(o, conv.weight.detach().cpu(), None, [3, 3, 3])
    assert np.array_equal(t.get_spatial_locations().numpy(), o.coords.astype(np.int64))
    assert rel_err(out.features, oo.features) < 1e-4


@pytest.mark.parametrize("mode", [3, 4])
def test_heavy_duplication(mode):
    import sparseeventid_amd.scn as scn
    sp, B = [20, 20, 20], 2
    base = _line_cloud(40, sp)
    coords = np.concatenate([base, base, base[::-1], base[:10]])     # every site 3-4 times
    coords = np.concatenate([coords, np.concatenate([base[:, :3], np.ones((40, 1))], 1)])
    feats = np.random.default_rng(2).standard_normal((len(coords), 3)).astype(np.float32)
    t = scn.InputLayer(3, sp, mode=mode)((torch.from_numpy(coords), torch.from_numpy(feats).cuda(), B))
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp, mode)
    assert t.features.shape[0] == 80
    assert np.array_equal(t.get_spatial_locations().numpy(), o.coords.astype(np.int64))
    assert rel_err(t.features, o.features) < 1e-5                    # 3-4 addends: fp32 order may differ


def test_125_offset_filter_on_the_mfma_path():
    import sparseeventid_amd.scn as scn
    sp, B = [24, 24, 24], 2
    coords = np.concatenate([_line_cloud(300, sp, 0), _line_cloud(280, sp, 1, (5, 9, 2))])
    feats = np.random.default_rng(4).standard_normal((len(coords), 16)).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, B))
    conv = scn.SubmanifoldConvolution(3, 16, 32, 5, True).cuda()
    out = conv(t)
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(g.cuda())
    fo = torch.from_numpy(feats).requires_grad_(True)
    W, b = conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
    oo = scn_cpu.subm_conv(scn_cpu.input_layer(coords, fo, B, sp), W, b, [5, 5, 5])
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4 and rel_err(f.grad, fo.grad) < 1e-4


@pytest.mark.parametrize("ci,co", [(128, 128), (160, 160), (96, 192)])
def test_wide_layers_on_a_large_level(ci, co):
    """>= 100 000 dst rows switch the wide layers (nOut > 96) to 128-row tiles with one workgroup per
    (tile, column block): forward, dX (roles of ci / co swapped) and dW against the oracle."""
    import sparseeventid_amd.scn as scn
    from tests.util import random_cloud
    sp, B = [56, 56, 56], 2
    coords, _ = random_cloud(31, B, sp, 52000, 1)
    n = len(coords)
    assert n >= 100000
    feats = np.random.default_rng(8).standard_normal((n, ci)).astype(np.float32)
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, B))
    assert t.features.shape[0] == n
    conv = scn.SubmanifoldConvolution(3, ci, co, 3, True).cuda()
    with torch.no_grad():
        conv.bias.normal_()
    out = conv(t)
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(g.cuda())
    torch.set_num_threads(16)
    fo = torch.from_numpy(feats).requires_grad_(True)
    W, b = conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
    oo = scn_cpu.subm_conv(scn_cpu.input_layer(coords, fo, B, sp), W, b, [3, 3, 3])
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < 1e-4
    assert rel_err(f.grad, fo.grad) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4 and rel_err(conv.bias.grad, b.grad) < 1e-4
