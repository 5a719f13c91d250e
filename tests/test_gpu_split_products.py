"""GPU: the OPT-IN product mode of the forward / dX convolutions (sei_set_product_mode(1), scn.set_product_mode("bf16x3")):
every fp32 operand as an exact sum of three bf16 numbers, six part products on the bf16 matrix pipe, fp32 accumulation.  Same
bar as the default mode -- within 1e-4 of the oracle, element by element -- and, measured against an fp64 run of the oracle,
an error of the size of the fp32 kernels' own.  The default mode stays bit-for-bit what it was."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import blob_cloud, elem_close, random_cloud, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture
def product_mode():
    import sparseeventid_amd.scn as scn
    yield scn.set_product_mode
    scn.set_product_mode("fp32")


def _conv_both_modes(scn, conv, t, x, gy, product_mode):
    res = {}
    for mode in ("fp32", "bf16x3"):
        product_mode(mode)
        xx = x.clone().requires_grad_(True)
        conv.weight.grad = None
        y = conv(scn.SparseConvNetTensor(xx, t.metadata, t.spatial_size)).features
        y.backward(gy)
        res[mode] = (y.detach().clone(), xx.grad.clone(), conv.weight.grad.clone())
    return res


@pytest.mark.parametrize("ci,co,filt", [(64, 64, [3, 3, 3]), (96, 96, [3, 3, 3]), (128, 128, [3, 3, 3]), (160, 160, [3, 3, 3]),
                                        (192, 192, [3, 3, 3]), (96, 128, [3, 3, 3]), (192, 128, [1, 1, 1]), (64, 96, [1, 3, 3])])
def test_split_products_meet_the_oracle(ci, co, filt, force_conv_kernel, product_mode):
    import sparseeventid_amd.scn as scn
    force_conv_kernel("rounds")          # the sixteen-wave round kernels on a small cloud
    sp, B = [26, 22, 24], 3
    coords, _ = blob_cloud(31 + ci, B, sp, 1500, 1)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), torch.ones(len(coords), 1).cuda(), B))
    n = t.features.shape[0]
    gen = torch.Generator(device="cuda").manual_seed(ci * 1000 + co)
    x = torch.randn(n, ci, device="cuda", generator=gen)
    gy = torch.randn(n, co, device="cuda", generator=gen)
    conv = scn.SubmanifoldConvolution(3, ci, co, filt, True).cuda()
    with torch.no_grad():
        conv.bias.normal_()
    res = _conv_both_modes(scn, conv, t, x, gy, product_mode)
    # the opt-in path really ran (its sums differ in the last bits) and only the forward / dX products changed
    assert not torch.equal(res["fp32"][0], res["bf16x3"][0])
    assert torch.equal(res["fp32"][2], res["bf16x3"][2])                      # dW: untouched
    assert rel_err(res["bf16x3"][0], res["fp32"][0]) < 5e-6 and rel_err(res["bf16x3"][1], res["fp32"][1]) < 5e-6
    # against the oracle, fp32 and fp64
    fo = x.cpu().clone().requires_grad_(True)
    W, b = conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
    o = scn_cpu.input_layer(coords, torch.ones(len(coords), 1), B, sp)
    o.features = fo
    oo = scn_cpu.subm_conv(o, W, b, filt)
    oo.features.backward(gy.cpu())
    for got, want in ((res["bf16x3"][0], oo.features), (res["bf16x3"][1], fo.grad)):
        assert rel_err(got, want) < 1e-4
        ok, worst = elem_close(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
        assert ok, worst
    f64 = x.cpu().double()
    o64 = scn_cpu.input_layer(coords, torch.ones(len(coords), 1), B, sp)
    o64.features = f64
    y64 = scn_cpu.subm_conv(o64, W.detach().double(), b.detach().double(), filt).features
    e_split, e_fp32 = rel_err(res["bf16x3"][0], y64), rel_err(res["fp32"][0], y64)
    assert e_split < 3 * e_fp32 + 1e-7, (e_split, e_fp32)                       # an error of the size of fp32's own


def test_split_products_on_a_large_level_default_plan(product_mode):
    """> 100 000 rows: the DEFAULT plan takes the round kernels, and with the opt-in their split form: oracle, both passes."""
    import sparseeventid_amd.scn as scn
    ci = co = 128
    sp, B = [56, 56, 56], 2
    coords, _ = random_cloud(31, B, sp, 52000, 1)
    n = len(coords)
    feats = np.random.default_rng(8).standard_normal((n, ci)).astype(np.float32)
    product_mode("bf16x3")
    f = torch.from_numpy(feats).cuda().requires_grad_(True)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), f, B))
    conv = scn.SubmanifoldConvolution(3, ci, co, 3, True).cuda()
    out = conv(t)
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(g.cuda())
    torch.set_num_threads(16)
    fo = torch.from_numpy(feats).requires_grad_(True)
    W, b = conv.weight.detach().cpu().requires_grad_(True), conv.bias.detach().cpu().requires_grad_(True)
    oo = scn_cpu.subm_conv(scn_cpu.input_layer(coords, fo, B, sp), W, b, [3, 3, 3])
    oo.features.backward(g)
    assert rel_err(out.features, oo.features) < 1e-4 and rel_err(f.grad, fo.grad) < 1e-4
    assert rel_err(conv.weight.grad, W.grad) < 1e-4


def test_strided_layers_with_split_products(force_conv_kernel, product_mode):
    """scn.Convolution 2^3 / stride 2 (8 offsets; dX through the transposed rules) in the opt-in mode against the default."""
    import sparseeventid_amd.scn as scn
    force_conv_kernel("rounds")
    sp, B = [31, 29, 33], 2
    coords, _ = blob_cloud(91, B, sp, 3000, 1)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), torch.ones(len(coords), 1).cuda(), B))
    n = t.features.shape[0]
    for ci, co in ((96, 128), (64, 96), (128, 160)):
        x = torch.randn(n, ci, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        conv = scn.Convolution(3, ci, co, 2, 2, False).cuda()
        res = {}
        for mode in ("fp32", "bf16x3"):
            product_mode(mode)
            xx = x.clone().requires_grad_(True)
            y = conv(scn.SparseConvNetTensor(xx, t.metadata, t.spatial_size)).features
            y.backward(torch.randn(y.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)))
            res[mode] = (y.detach().clone(), xx.grad.clone())
        for a, b in zip(res["fp32"], res["bf16x3"]):
            assert rel_err(b, a) < 5e-6, (ci, co)
