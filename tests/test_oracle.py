"""CPU: pin the oracle (oracle/scn_cpu.py + oracle/scn_rules.c) by dense-conv equivalence
(oracle/dense_ref.py) -- the parity definition of SURVEY.md 8c."""
import numpy as np
import pytest
import torch

from oracle import dense_ref, scn_cpu
from tests.util import blob_cloud, random_cloud, rel_err, sort_rows

TOL = 1e-5


@pytest.mark.parametrize("size", [[3, 3, 3], [5, 5, 5], [1, 3, 3], [1, 5, 5], [1, 1, 1]])
def test_subm_conv_matches_dense(size):
    sp, B, ci, co = [12, 10, 14], 2, 3, 5
    coords, feats = blob_cloud(1, B, sp, 150, ci)
    t = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp, mode=3)
    K = size[0] * size[1] * size[2]
    g = torch.Generator().manual_seed(2)
    W = torch.randn(K, ci, co, generator=g); b = torch.randn(co, generator=g)
    out = scn_cpu.subm_conv(t, W, b, size)
    ref = dense_ref.subm_conv_dense(t.coords, t.features, sp, B, W, b, size)
    assert rel_err(out.features, ref) < TOL


@pytest.mark.parametrize("sp,size", [([12, 10, 14], [2, 2, 2]), ([13, 9, 15], [2, 2, 2]),
                                     ([3, 12, 10], [1, 2, 2]), ([12, 12, 12], [3, 3, 3])])
def test_strided_conv_matches_dense(sp, size):
    B, ci, co = 2, 4, 6
    stride = [2 if s > 1 else 1 for s in size]
    coords, feats = random_cloud(3, B, sp, 120, ci)
    t = scn_cpu.input_layer(coords, torch.from_numpy(feats), B, sp)
    K = size[0] * size[1] * size[2]
    W = torch.randn(K, ci, co, generator=torch.Generator().manual_seed(4))
    out = scn_cpu.strided_conv(t, W, None, size, stride)
    yd, active = dense_ref.strided_conv_dense(t.coords, t.features, sp, B, W, size, stride)
    assert out.spatial_size == list(yd.shape[2:])
    # same active set
    m = dense_ref.active_mask(out.coords, out.spatial_size, B)
    assert torch.equal(m, active)
    assert out.coords.shape[0] == int(active.sum())
    ref = dense_ref.gather_rows(yd, out.coords)
    assert rel_err(out.features, ref) < TOL


def test_odd_size_clipping_900_like():
    # 900 -> 450 -> 225 -> 112: sites at coordinate 224 of the 225-level are dropped
    sp = [9, 9, 9]
    coords = np.array([[8, 8, 8, 0], [0, 0, 0, 0], [7, 6, 8, 0]], dtype=np.float64)
    t = scn_cpu.input_layer(coords, torch.ones(3, 1), 1, sp)
    out = scn_cpu.strided_conv(t, torch.ones(8, 1, 1), None, [2, 2, 2], [2, 2, 2])
    assert out.spatial_size == [4, 4, 4]
    assert out.coords.shape[0] == 1 and out.coords[0].tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_input_layer_modes(mode):
    sp, B = [6, 6, 6], 2
    coords = np.array([[1, 2, 3, 0], [1, 2, 3, 0], [0, 0, 0, 1], [1, 2, 3, 0], [5, 5, 5, 1]], np.float64)
    f = torch.tensor([[1.0], [2.0], [4.0], [8.0], [16.0]])
    t = scn_cpu.input_layer(coords, f, B, sp, mode=mode)
    assert t.coords.tolist() == [[1, 2, 3, 0], [0, 0, 0, 1], [5, 5, 5, 1]]
    want = {1: 8.0, 2: 1.0, 3: 11.0, 4: 11.0 / 3}[mode]
    assert abs(float(t.features[0, 0]) - want) < 1e-6
    assert t.features[1:, 0].tolist() == [4.0, 16.0]


def test_input_layer_out_of_bounds():
    with pytest.raises(RuntimeError):
        scn_cpu.input_layer(np.array([[6, 0, 0, 0]], np.float64), torch.ones(1, 1), 1, [6, 6, 6])


def test_batchnorm_matches_torch():
    x = torch.randn(500, 7, generator=torch.Generator().manual_seed(0)) * 3 + 1
    t = scn_cpu.OracleTensor(x, np.zeros((500, 4), np.int32), [1, 1, 1], 1)
    rm, rv = torch.zeros(7), torch.ones(7)
    y = scn_cpu.batchnorm(t, torch.ones(7) * 1.5, torch.ones(7) * 0.2, rm, rv)
    bn = torch.nn.BatchNorm1d(7, eps=1e-4, momentum=0.1)
    bn.weight.data.fill_(1.5); bn.bias.data.fill_(0.2)
    ref = bn(x)
    assert rel_err(y.features, ref) < 1e-5
    assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5


def _leaf(t):
    return t.detach().clone().requires_grad_(True)


@pytest.mark.parametrize("size", [[3, 3, 3], [5, 5, 5], [1, 3, 3], [1, 5, 5], [1, 1, 1]])
def test_subm_conv_backward_matches_dense_autograd(size):
    """The oracle's hand-written backward (scn_cpu._RuleConv.backward) against torch autograd through conv3d on the
    densified input: dX, dW and dbias for the same upstream gradient."""
    sp, B, ci, co = [12, 10, 14], 2, 3, 5
    coords, feats = blob_cloud(1, B, sp, 150, ci)
    K = size[0] * size[1] * size[2]
    g = torch.Generator().manual_seed(2)
    W0 = torch.randn(K, ci, co, generator=g); b0 = torch.randn(co, generator=g)
    f1, W1, b1 = _leaf(torch.from_numpy(feats)), _leaf(W0), _leaf(b0)
    t = scn_cpu.input_layer(coords, f1, B, sp, mode=3)
    out = scn_cpu.subm_conv(t, W1, b1, size)
    up = torch.randn(out.features.shape, generator=g)
    out.features.backward(up)
    f2, W2, b2 = _leaf(torch.from_numpy(feats)), _leaf(W0), _leaf(b0)
    t2 = scn_cpu.input_layer(coords, f2.detach(), B, sp, mode=3)          # rows / coords only
    rows = torch.from_numpy(scn_cpu.input_rows(coords, sp, B)[0].astype(np.int64))
    x2 = torch.zeros(t2.coords.shape[0], ci).index_add(0, rows, f2)        # InputLayer mode 3, differentiable
    ref = dense_ref.subm_conv_dense(t2.coords, x2, sp, B, W2, b2, size)
    ref.backward(up)
    assert rel_err(out.features, ref) < TOL
    assert rel_err(f1.grad, f2.grad) < TOL
    assert rel_err(W1.grad, W2.grad) < TOL
    assert rel_err(b1.grad, b2.grad) < TOL


@pytest.mark.parametrize("sp,size", [([12, 10, 14], [2, 2, 2]), ([13, 9, 15], [2, 2, 2]),
                                     ([3, 12, 10], [1, 2, 2]), ([12, 12, 12], [3, 3, 3])])
def test_strided_conv_backward_matches_dense_autograd(sp, size):
    B, ci, co = 2, 4, 6
    stride = [2 if s > 1 else 1 for s in size]
    coords, feats = random_cloud(3, B, sp, 120, ci)
    K = size[0] * size[1] * size[2]
    W0 = torch.randn(K, ci, co, generator=torch.Generator().manual_seed(4))
    f1, W1 = _leaf(torch.from_numpy(feats)), _leaf(W0)
    t = scn_cpu.input_layer(coords, f1, B, sp)
    out = scn_cpu.strided_conv(t, W1, None, size, stride)
    up = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(5))
    out.features.backward(up)
    f2, W2 = _leaf(torch.from_numpy(feats)), _leaf(W0)
    rows = torch.from_numpy(scn_cpu.input_rows(coords, sp, B)[0].astype(np.int64))
    x2 = torch.zeros(t.coords.shape[0], ci).index_add(0, rows, f2)
    yd, _ = dense_ref.strided_conv_dense(t.coords, x2, sp, B, W2, size, stride)
    ref = dense_ref.gather_rows(yd, out.coords)
    ref.backward(up)
    assert rel_err(out.features, ref) < TOL
    assert rel_err(f1.grad, f2.grad) < TOL
    assert rel_err(W1.grad, W2.grad) < TOL


def test_encoder_tiny_matches_dense_forward_and_backward():
    """Whole Encoder.forward on a tiny grid equals a dense re-implementation with masks (torch conv3d + autograd):
    output, and the gradient of every parameter and of the input features."""
    cfg = scn_cpu.NetConfig(depth=2, n_initial_filters=4, n_output_filters=6, blocks_per_layer=1)
    sp = [16, 16, 16]
    coords, feats = blob_cloud(7, 2, sp, 200, 1)
    sd0 = scn_cpu.init_params(cfg, seed=1)
    g = torch.Generator().manual_seed(11)
    for k in sd0:                                   # non-trivial BN affine parameters and biases
        if k.endswith("norm.weight"):
            sd0[k] = 0.5 + torch.rand(sd0[k].shape, generator=g)
        elif k.endswith(".bias"):
            sd0[k] = torch.randn(sd0[k].shape, generator=g) * 0.3

    def run(fn):
        sd = {k: (_leaf(v) if "running" not in k else v.clone()) for k, v in sd0.items()}
        f = _leaf(torch.from_numpy(feats))
        out = fn(sd, cfg, (coords, f, 2), [1] + sp)
        return out, sd, f

    out, sd1, f1 = run(scn_cpu.encoder_forward)
    assert out.shape == (2, 6, 4, 4, 4)
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    ref, sd2, f2 = run(dense_ref.encoder_forward_dense)
    assert rel_err(out, ref) < 1e-5
    up = torch.randn(out.shape, generator=g)
    out.backward(up)
    ref.backward(up)
    assert rel_err(f1.grad, f2.grad) < 1e-4
    for k in sd1:
        if "running" in k:
            continue
        assert sd2[k].grad is not None, k
        if k.endswith(".conv1.bias"):
            # a conv bias directly in front of a training-mode BatchNorm has an exactly zero gradient; both sides
            # hold only rounding noise: compare against the scale of the same convolution's weight gradient
            wscale = float(sd2[k[:-4] + "weight"].grad.abs().max())
            assert float(sd1[k].grad.abs().max()) < 1e-4 * wscale and float(sd2[k].grad.abs().max()) < 1e-4 * wscale, k
        else:
            assert rel_err(sd1[k].grad, sd2[k].grad) < 1e-4, k
