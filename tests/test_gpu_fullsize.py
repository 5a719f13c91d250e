"""GPU: BASELINE.json's full size (64 events x 50 000 voxels in 900^3) through size-independent
properties -- the oracle would need minutes there, so the checks are algebraic:
unique-site count, rulebook mirror symmetry, linearity, conservation under a ones-filter,
BatchNorm moments, SparseToDense mass, run-to-run determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GRID = [900, 900, 900]
B, NVOX = 64, 50000


@pytest.fixture(scope="module")
def batch():
    from bench import generate_batch
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    arr, _ = generate_batch(0, 0, B, tuple(GRID), NVOX, "tracks", 1)   # no fork: the GPU may be initialised
    return larcvsparse_to_scnsparse_3d(arr)


@pytest.fixture(scope="module")
def level0(batch):
    import sparseeventid_amd.scn as scn
    c, f, b = batch
    return scn.InputLayer(3, GRID)((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))


def test_input_layer_counts(batch, level0):
    c, f, b = batch
    assert level0.features.shape == (B * NVOX, 1)              # generator emits distinct voxels per event
    locs = level0.get_spatial_locations().numpy()
    assert np.array_equal(locs, c.astype(np.int64))            # first-appearance order == input order
    assert torch.equal(level0.features.cpu(), torch.from_numpy(f))


def test_submanifold_rulebook_symmetry(level0):
    rb = level0.metadata.subm_rulebook(GRID, [3, 3, 3])
    start, ri, ro = rb.export("cuda")
    cnt = np.diff(np.asarray(start))
    assert cnt[13] == B * NVOX                                  # centre offset: every site sees itself
    assert np.array_equal(cnt, cnt[::-1])                       # (i -> o, k) <=> (o -> i, K-1-k)
    assert rb.n_rules == int(cnt.sum())
    k = 5
    a = torch.stack([ri[start[k]:start[k + 1]], ro[start[k]:start[k + 1]]], 1)
    m = torch.stack([ro[start[26 - k]:start[27 - k]], ri[start[26 - k]:start[27 - k]]], 1)
    key = lambda t: (t[:, 0].long() * (B * NVOX) + t[:, 1].long()).sort().values
    assert torch.equal(key(a), key(m))


def test_conv_linearity_and_determinism(level0):
    import sparseeventid_amd.scn as scn
    n = level0.features.shape[0]
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, 32, device="cuda", generator=g)
    y = torch.randn(n, 32, device="cuda", generator=g)
    conv = scn.SubmanifoldConvolution(3, 32, 32, 3, False).cuda()

    def run(feat):
        t = scn.SparseConvNetTensor(feat, level0.metadata, level0.spatial_size)
        return conv(t).features.detach()
    cx, cy = run(x), run(y)
    cl = run(2.5 * x - 0.75 * y)
    ref = 2.5 * cx - 0.75 * cy
    assert float((cl - ref).abs().max() / ref.abs().max()) < 1e-4
    assert torch.equal(cx, run(x))                              # bit-identical rerun


def test_strided_conv_conserves_mass(level0):
    import sparseeventid_amd.scn as scn
    n = level0.features.shape[0]
    x = torch.rand(n, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    conv = scn.Convolution(3, 16, 16, 2, 2, False).cuda()
    with torch.no_grad():
        conv.weight.zero_()
        for k in range(8):
            conv.weight[k] = torch.eye(16)
    out = conv(scn.SparseConvNetTensor(x, level0.metadata, level0.spatial_size))
    assert [int(v) for v in out.spatial_size] == [450, 450, 450]
    # identity filter at every offset, 900 even => every input feeds exactly one output: mass is conserved
    assert float((out.features.sum(0) - x.sum(0)).abs().max() / x.sum(0).abs().max()) < 1e-4
    locs = level0.get_spatial_locations()
    parents = torch.unique(torch.cat([locs[:, :3] // 2, locs[:, 3:]], 1), dim=0)
    assert out.features.shape[0] == parents.shape[0]


def test_batchnorm_moments_and_dense_mass(level0):
    import sparseeventid_amd.scn as scn
    n = level0.features.shape[0]
    x = torch.randn(n, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 3 + 2
    bn = scn.BatchNormalization(32).cuda()
    y = bn.forward_features(x)
    assert float(y.mean(0).abs().max()) < 1e-4 and float((y.var(0, unbiased=False) - 1).abs().max()) < 1e-3
    assert float((bn.running_mean - 0.1 * x.mean(0)).abs().max()) < 1e-4
    # SparseToDense on a coarse level: total mass preserved, tanh(0)=0 keeps inactive cells exact zeros
    t = scn.SparseConvNetTensor(torch.rand(n, 16, device="cuda"), level0.metadata, level0.spatial_size)
    for _ in range(5):
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    dense = scn.SparseToDense(3, 16)(t)
    assert dense.shape == (B, 16, 28, 28, 28)
    assert float((dense.sum() - t.features.sum()).abs() / t.features.sum().abs()) < 1e-4
    assert int((dense != 0).sum()) <= t.features.numel()
