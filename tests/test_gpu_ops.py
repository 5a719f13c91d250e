"""GPU parity tests: every engine op (through the scn module surface -> C ABI -> HIP kernels)
against the CPU oracle (oracle/scn_cpu.py) on the same seeded inputs.
Tolerance: 1e-4 relative fp32 (BASELINE.json north_star); integer work is compared exactly.
"""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import blob_cloud, elem_close, random_cloud, rel_err, sort_rows

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine():
    import sparseeventid_amd.scn as scn
    return scn


def _input(scn, coords, feats, B, sp, mode=3, requires_grad=False):
    f = torch.from_numpy(feats).cuda().requires_grad_(requires_grad)
    t = scn.InputLayer(3, sp, mode=mode)((torch.from_numpy(coords), f, B))
    return t, f


def test_library_loaded_is_in_tree():
    from sparseeventid_amd import _lib
    assert _lib.lib().sei_abi_version() == 1
    assert _lib.LIB_PATH.endswith("sparseeventid_amd/libsparseeventid_amd.so")


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_input_layer_rows_and_features(mode):
    scn = _engine()
    sp, B = [20, 24, 18], 3
    coords, feats = random_cloud(11, B, sp, [300, 0, 150], 2, dup_frac=0.0 if mode == 0 else 0.3)
    t, _ = _input(scn, coords, feats, B, sp, mode)
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp, mode)
    # integer part: identical rows in identical (first-appearance) order
    assert np.array_equal(t.get_spatial_locations().numpy(), o.coords.astype(np.int64))
    assert rel_err(t.features, o.features) < 1e-6


def test_input_layer_empty_and_oob():
    scn = _engine()
    t = scn.InputLayer(3, [8, 8, 8])((torch.zeros(0, 4), torch.zeros(0, 1).cuda(), 2))
    assert t.features.shape == (0, 1)
    with pytest.raises(RuntimeError):
        scn.InputLayer(3, [8, 8, 8])((torch.tensor([[1., 2., 8., 0.]]), torch.ones(1, 1).cuda(), 1))
    with pytest.raises(RuntimeError):
        scn.InputLayer(3, [8, 8, 8])((torch.tensor([[1., 2., 3., 1.]]), torch.ones(1, 1).cuda(), 1))


def _rule_set(start, ri, ro, K):
    s = set()
    for k in range(K):
        for a in range(int(start[k]), int(start[k + 1])):
            s.add((k, int(ri[a]), int(ro[a])))
    return s


@pytest.mark.parametrize("filt", [[3, 3, 3], [5, 5, 5], [1, 3, 3], [1, 1, 1]])
def test_subm_rulebook_exact(filt):
    scn = _engine()
    sp, B = [16, 12, 20], 2
    coords, feats = blob_cloud(5, B, sp, 300, 1)
    t, _ = _input(scn, coords, feats, B, sp)
    rb = t.metadata.subm_rulebook(sp, filt)
    start, ri, ro = rb.export("cuda")
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    ostart, ori, oro = scn_cpu.subm_rules(o.coords, sp, filt)
    K = filt[0] * filt[1] * filt[2]
    assert rb.n_rules == len(ori)
    assert list(start) == [int(x) for x in ostart]
    assert _rule_set(start, ri.cpu().numpy(), ro.cpu().numpy(), K) == _rule_set(ostart, ori, oro, K)


@pytest.mark.parametrize("big,small", [([5, 5, 5], [3, 3, 3]), ([5, 5, 5], [1, 3, 3]), ([3, 5, 5], [3, 3, 1]), ([1, 5, 5], [1, 3, 3])])
def test_subm_rulebook_cut_from_a_larger_filters_table(big, small):
    """A submanifold rulebook whose level already holds one with a larger (centred, odd) filter -- the 5^3 stem next to
    the 3^3 blocks of level 0 -- is cut from THAT rulebook's neighbour table instead of probing again: same rules as the
    oracle, for two tile sizes."""
    scn = _engine()
    sp, B = [16, 12, 20], 2
    coords, feats = blob_cloud(6, B, sp, 320, 1)
    t, _ = _input(scn, coords, feats, B, sp)
    rb_big = t.metadata.subm_rulebook(sp, big)
    rb_big.stats(256)                                   # probes and keeps the table
    rb = t.metadata.subm_rulebook(sp, small)
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    ostart, ori, oro = scn_cpu.subm_rules(o.coords, sp, small)
    K = small[0] * small[1] * small[2]
    for tile_rows in (128, 64):
        assert rb.stats(tile_rows)[0] == len(ori)
    start, ri, ro = rb.export("cuda")
    assert list(start) == [int(x) for x in ostart]
    assert _rule_set(start, ri.cpu().numpy(), ro.cpu().numpy(), K) == _rule_set(ostart, ori, oro, K)
    # and the donor's own rules are untouched
    bstart, bri, bro = rb_big.export("cuda")
    obs, obi, obo = scn_cpu.subm_rules(o.coords, sp, big)
    Kb = big[0] * big[1] * big[2]
    assert _rule_set(bstart, bri.cpu().numpy(), bro.cpu().numpy(), Kb) == _rule_set(obs, obi, obo, Kb)


@pytest.mark.parametrize("sp,size,stride", [([16, 12, 20], [2, 2, 2], [2, 2, 2]), ([15, 13, 9], [2, 2, 2], [2, 2, 2]),
                                            ([3, 16, 12], [1, 2, 2], [1, 2, 2]), ([12, 12, 12], [3, 3, 3], [2, 2, 2])])
def test_strided_rulebook_exact(sp, size, stride):
    scn = _engine()
    B = 2
    coords, feats = random_cloud(6, B, sp, 250, 1)
    t, _ = _input(scn, coords, feats, B, sp)
    rb = t.metadata.conv_rulebook(sp, size, stride)
    o = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    oc, osp, ostart, ori, oro = scn_cpu.conv_rules(o.coords, sp, size, stride)
    assert rb.spatial_out == osp
    got = t.metadata.coords(osp, "cuda").cpu().numpy()
    assert np.array_equal(got, oc)          # same sites, same (first-appearance) order
    start, ri, ro = rb.export("cuda")
    K = size[0] * size[1] * size[2]
    assert _rule_set(start, ri.cpu().numpy(), ro.cpu().numpy(), K) == _rule_set(ostart, ori, oro, K)


CONV_CASES = [(1, 32, [5, 5, 5]), (32, 32, [3, 3, 3]), (64, 64, [3, 3, 3]), (96, 96, [3, 3, 3]),
              (128, 128, [3, 3, 3]), (160, 160, [3, 3, 3]), (192, 192, [3, 3, 3]), (192, 128, [1, 1, 1]),
              (16, 48, [1, 3, 3]), (3, 5, [3, 3, 3]), (32, 80, [3, 3, 3]), (224, 224, [3, 3, 3]), (224, 3, [1, 1, 1]), (40, 7, [3, 3, 3]),
              (256, 208, [1, 3, 3])]


@pytest.mark.parametrize("kernel", ["default", "tile", "wave", "band", "rounds"])
@pytest.mark.parametrize("ci,co,filt", CONV_CASES)
def test_subm_conv_forward_backward(ci, co, filt, kernel, force_conv_kernel):
    """`kernel` forces the workgroup-tile, the wave-private or the row-band MFMA kernel, or the tile kernel's round forms
    (two-offset and multi-offset rounds, wherever instantiated: sei_debug_force_conv_kernel) so that every family meets the
    oracle DIRECTLY for every plane count, whatever the default plan picks."""
    if kernel != "default":
        if ci % 16 or co % 16:
            pytest.skip("scalar kernels: no MFMA variant to force")
        force_conv_kernel(kernel)
    scn = _engine()
    sp, B = [14, 12, 16], 2
    coords, feats = blob_cloud(21, B, sp, 260, ci)
    t, f = _input(scn, coords, feats, B, sp, requires_grad=True)
    conv = scn.SubmanifoldConvolution(3, ci, co, filt, True).cuda()
    with torch.no_grad():
        conv.bias.normal_()
    out = conv(t)
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(3))
    out.features.backward(g.cuda())

    fo = torch.from_numpy(feats).requires_grad_(True)
    W = conv.weight.detach().cpu().requires_grad_(True)
    b = conv.bias.detach().cpu().requires_grad_(True)
    o = scn_cpu.input_layer(coords, fo, B, sp)
    oo = scn_cpu.subm_conv(o, W, b, filt)
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < TOL
    assert rel_err(conv.weight.grad, W.grad) < TOL
    assert rel_err(conv.bias.grad, b.grad) < TOL
    assert rel_err(f.grad, fo.grad) < TOL
    # and element by element (round-4 review: rel_err is a max-norm ratio): every output / input-gradient element within
    # 1e-4 of the oracle's, with an absolute floor for elements that are themselves a cancellation of the row's products
    for got, want in ((out.features, oo.features), (f.grad, fo.grad)):
        ok, worst = elem_close(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
        assert ok, worst


@pytest.mark.parametrize("kernel", ["default", "tile", "wave", "band", "rounds"])
@pytest.mark.parametrize("ci,co,size", [(32, 64, [2, 2, 2]), (64, 96, [2, 2, 2]), (160, 192, [2, 2, 2]),
                                        (16, 32, [1, 2, 2]), (5, 7, [2, 2, 2]), (32, 32, [3, 3, 3])])
def test_strided_conv_forward_backward(ci, co, size, kernel, force_conv_kernel):
    if kernel != "default":
        if ci % 16 or co % 16:
            pytest.skip("scalar kernels: no MFMA variant to force")
        force_conv_kernel(kernel)
    scn = _engine()
    sp, B = [15, 12, 18], 2
    if size[0] == 1:
        sp = [3, 14, 18]
    stride = [2 if s > 1 else 1 for s in size]
    coords, feats = blob_cloud(22, B, sp, 240, ci)
    t, f = _input(scn, coords, feats, B, sp, requires_grad=True)
    conv = scn.Convolution(3, ci, co, size, stride, False).cuda()
    out = conv(t)
    fo = torch.from_numpy(feats).requires_grad_(True)
    W = conv.weight.detach().cpu().requires_grad_(True)
    o = scn_cpu.input_layer(coords, fo, B, sp)
    oo = scn_cpu.strided_conv(o, W, None, size, stride)
    assert np.array_equal(out.get_spatial_locations().numpy(), oo.coords.astype(np.int64))
    g = torch.randn(oo.features.shape, generator=torch.Generator().manual_seed(4))
    out.features.backward(g.cuda())
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < TOL
    assert rel_err(conv.weight.grad, W.grad) < TOL
    assert rel_err(f.grad, fo.grad) < TOL
    for got, want in ((out.features, oo.features), (f.grad, fo.grad)):
        ok, worst = elem_close(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
        assert ok, worst


@pytest.mark.parametrize("C,leak,with_res", [(32, 1.0, False), (32, 1 / 3, False), (96, 1 / 3, True), (7, 0.0, False),
                                             (192, 1 / 3, True), (1, 1 / 3, False)])
def test_batchnorm_fused_forward_backward(C, leak, with_res):
    scn = _engine()
    n = 3001
    gen = torch.Generator().manual_seed(9)
    x = (torch.randn(n, C, generator=gen) * 2.5 + 0.7)
    r = torch.randn(n, C, generator=gen) if with_res else None
    g = torch.randn(n, C, generator=gen)
    bn = scn.BatchNormalization(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    xg = x.clone().cuda().requires_grad_(True)
    rg = r.clone().cuda().requires_grad_(True) if with_res else None
    y = bn.forward_features(xg, leak=leak, residual=rg)
    y.backward(g.cuda())

    xo = x.clone().requires_grad_(True)
    ro = r.clone().requires_grad_(True) if with_res else None
    w = bn.weight.detach().cpu().requires_grad_(True)
    b = bn.bias.detach().cpu().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    t = scn_cpu.OracleTensor(xo, np.zeros((n, 4), np.int32), [1, 1, 1], 1)
    z = scn_cpu.batchnorm(t, w, b, rm, rv).features
    if with_res:
        z = z + ro
    yo = torch.where(z > 0, z, z * leak)
    yo.backward(g)
    assert rel_err(y, yo) < TOL
    assert rel_err(xg.grad, xo.grad) < TOL
    assert rel_err(bn.weight.grad, w.grad) < TOL
    assert rel_err(bn.bias.grad, b.grad) < TOL
    if with_res:
        assert rel_err(rg.grad, ro.grad) < TOL
    assert rel_err(bn.running_mean, rm) < 1e-5 and rel_err(bn.running_var, rv) < 1e-5
    # eval mode uses the running statistics
    bn.eval()
    ye = bn.forward_features(x.cuda(), leak=1.0)
    ze = scn_cpu.batchnorm(scn_cpu.OracleTensor(x, None, [1, 1, 1], 1), w.detach(), b.detach(), rm, rv,
                           training=False).features
    assert rel_err(ye, ze) < TOL


@pytest.mark.parametrize("C,n,leak", [(32, 3001, 1 / 3), (96, 70001, 1 / 3), (7, 515, 0.0), (160, 9000, 1.0)])
def test_bn_backward_colsum_and_recomputed_mask(C, n, leak):
    """`sei_bn_backward_colsum` through the C ABI: (i) the column sums of dx it emits equal dx.sum(0) (the conv bias
    gradient), (ii) with y = NULL (no residual) the LeakyReLU mask recomputed from x, gamma, beta gives the SAME dx,
    dgamma, dbeta as with the stored activation -- bit for bit (same expression as the forward)."""
    import ctypes
    from sparseeventid_amd._lib import lib, check
    scn = _engine()
    gen = torch.Generator().manual_seed(21)
    x = (torch.randn(n, C, generator=gen) * 1.7 - 0.3).cuda()
    g = torch.randn(n, C, generator=gen).cuda()
    bn = scn.BatchNormalization(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    y = bn.forward_features(x, leak=leak)
    mean, var = x.double().mean(0), x.double().var(0, unbiased=False)
    save_mean = mean.float().contiguous()
    save_invstd = (1.0 / torch.sqrt(var + bn.eps)).float().contiguous()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for y_arg in (y, None):
        dx, dg, db, cs = torch.empty_like(x), torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        check(lib().sei_bn_backward_colsum(P(x), P(y_arg), P(g), n, C, P(bn.weight), P(bn.bias), P(save_mean), P(save_invstd),
                                           leak, 1, P(dx), None, P(dg), P(db), P(cs), st))
        outs.append((dx, dg, db, cs))
    (dx0, dg0, db0, cs0), (dx1, dg1, db1, cs1) = outs
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1) and torch.equal(cs0, cs1)
    ref = dx0.double().sum(0)
    scale = float(dx0.double().abs().sum(0).max())
    assert float((cs0.double() - ref).abs().max()) < 1e-6 * scale
    # and the gradient itself against autograd of the same expression in fp64
    xd = x.double().requires_grad_(True)
    z = (xd - xd.mean(0)) / torch.sqrt(xd.var(0, unbiased=False) + bn.eps) * bn.weight.double() + bn.bias.double()
    torch.where(z > 0, z, z * leak).backward(g.double())
    assert rel_err(dx0, xd.grad) < TOL


def test_elementwise_and_dense():
    scn = _engine()
    sp, B, C = [6, 5, 7], 3, 8
    coords, feats = random_cloud(31, B, sp, 60, C)
    t, f = _input(scn, coords, feats, B, sp, requires_grad=True)
    a = scn.LeakyReLU()(t)
    s = scn.AddTable()([a, t])
    r = scn.ReLU()(s)
    d = scn.SparseToDense(3, C)(r)
    assert d.shape == (B, C, 6, 5, 7)
    g = torch.randn(d.shape, generator=torch.Generator().manual_seed(1))
    d.backward(g.cuda())
    fo = torch.from_numpy(feats).requires_grad_(True)
    o = scn_cpu.input_layer(coords, fo, B, sp)
    oa = scn_cpu.leaky_relu(o)
    os_ = scn_cpu.add_table([oa, o])
    orr = scn_cpu.leaky_relu(os_, 0.0)
    od = scn_cpu.sparse_to_dense(orr, C)
    od.backward(g)
    assert rel_err(d, od) < 1e-6
    assert rel_err(f.grad, fo.grad) < 1e-6


def test_conv_is_deterministic():
    scn = _engine()
    sp, B = [14, 12, 16], 2
    coords, feats = blob_cloud(40, B, sp, 300, 32)
    conv = scn.SubmanifoldConvolution(3, 32, 32, 3, True).cuda()
    outs = []
    for _ in range(2):
        t, f = _input(scn, coords, feats, B, sp, requires_grad=True)
        o = conv(t)
        o.features.sum().backward()
        outs.append((o.features.detach().clone(), conv.weight.grad.clone(), f.grad.clone()))
        conv.weight.grad = None
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_device_larcv_transform_matches_host_transform():
    import os
    scn = _engine()
    from sparseeventid_amd import data
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, host_fn in (("transform_3d.npz", data.larcvsparse_to_scnsparse_3d),
                          ("transform_2d.npz", data.larcvsparse_to_scnsparse_2d)):
        g = np.load(os.path.join(gold, name))
        c, f, b = scn.larcv_to_scn_device(g["input"])
        assert b == int(g["batch_size"])
        assert np.array_equal(c.cpu().numpy().astype(np.float64), g["coords"])      # reference golden, exact
        assert np.array_equal(f.cpu().numpy(), g["features"])
        hc, hf, hb = host_fn(g["input"])
        assert np.array_equal(c.cpu().numpy(), hc.astype(np.int32))
    # and the tuple feeds InputLayer directly
    arr, _ = data.synthetic_dune3d_batch(0, 0, batch_size=2, n_voxels=3000, grid=(128, 128, 128))
    c, f, b = scn.larcv_to_scn_device(arr)
    t = scn.InputLayer(3, [128, 128, 128])((c, f, b))
    assert t.features.shape == (6000, 1)
