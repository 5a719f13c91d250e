"""oracle/dense_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Dense-convolution equivalence: the definition of parity for this project
(SURVEY.md 8c), since the reference's arithmetic is in the absent third-party
``sparseconvnet`` package and the reference holds no golden vectors
(PARITY UNPINNED by the reference itself).

For SCN weights ``W[k, i, o]`` with k row-major over the filter (last dim
fastest), the equivalent torch cross-correlation kernel is
``Wt[o, i, d0, d1, d2] = W[k, i, o]``:

  subm_conv(X)    == mask_active(conv3d(dense(X), Wt, padding=size//2)) + bias on active rows
  strided_conv(X) == conv3d(dense(X), Wt, stride=s) at windows holding >=1 active input

Call sites being modelled: /root/reference/src/networks/resnet.py:30-50,105-110;
sparse_building_blocks.py:29-34,110-117.
"""
import numpy as np
import torch
import torch.nn.functional as F


def densify(coords, features, spatial, batch_size):
    """[N,4] (x0,x1,x2,b) + [N,C] -> [B,C,D0,D1,D2]; duplicates summed."""
    c = torch.as_tensor(np.asarray(coords), dtype=torch.int64)
    d = torch.zeros(batch_size, spatial[0], spatial[1], spatial[2], features.shape[1], dtype=features.dtype)
    d.index_put_((c[:, 3], c[:, 0], c[:, 1], c[:, 2]), features, accumulate=True)
    return d.permute(0, 4, 1, 2, 3).contiguous()


def active_mask(coords, spatial, batch_size):
    c = torch.as_tensor(np.asarray(coords), dtype=torch.int64)
    m = torch.zeros(batch_size, spatial[0], spatial[1], spatial[2], dtype=torch.bool)
    m[c[:, 3], c[:, 0], c[:, 1], c[:, 2]] = True
    return m


def to_torch_kernel(W, size):
    K, ci, co = W.shape
    return W.reshape(size[0], size[1], size[2], ci, co).permute(4, 3, 0, 1, 2).contiguous()


def gather_rows(dense, coords):
    c = torch.as_tensor(np.asarray(coords), dtype=torch.int64)
    return dense.permute(0, 2, 3, 4, 1)[c[:, 3], c[:, 0], c[:, 1], c[:, 2]]


def subm_conv_dense(coords, features, spatial, batch_size, W, bias, size):
    """Rows of the submanifold conv output, in the row order of ``coords`` (unique)."""
    x = densify(coords, features, spatial, batch_size)
    y = F.conv3d(x, to_torch_kernel(W, size), padding=[s // 2 for s in size])
    rows = gather_rows(y, coords)
    return rows + bias if bias is not None else rows


def strided_conv_dense(coords, features, spatial, batch_size, W, size, stride):
    """Returns (out_dense [B,Co,...], out_active_mask [B,...]) of scn.Convolution."""
    x = densify(coords, features, spatial, batch_size)
    y = F.conv3d(x, to_torch_kernel(W, size), stride=stride)
    m = active_mask(coords, spatial, batch_size).float()[:, None]
    cnt = F.conv3d(m, torch.ones(1, 1, *size), stride=stride)
    return y, cnt[:, 0] > 0


# ----------------------------------------------------------------------------- whole encoder, dense + masks
def _masked_bn(y, m, weight, bias, eps=1e-4):
    """scn.BatchNormalization in training mode on a dense tensor: statistics over ACTIVE sites only (biased var)."""
    mf = m[:, None].to(y.dtype)
    n = mf.sum()
    mean = (y * mf).sum(dim=(0, 2, 3, 4)) / n
    var = (((y - mean[None, :, None, None, None]) ** 2) * mf).sum(dim=(0, 2, 3, 4)) / n
    z = (y - mean[None, :, None, None, None]) * torch.rsqrt(var + eps)[None, :, None, None, None]
    return (z * weight[None, :, None, None, None] + bias[None, :, None, None, None]) * mf


def _leaky(y, leak=1.0 / 3.0):
    return torch.where(y > 0, y, y * leak)


def encoder_forward_dense(sd, cfg, x, image_size):
    """Encoder.forward (/root/reference/src/networks/resnet.py:141-161) restated on DENSE tensors with active-site
    masks and torch's own conv3d + autograd: the pin of oracle/scn_cpu.encoder_forward, forward AND backward
    (training-mode BatchNorm, 3-D only).  Running statistics are not updated."""
    coords, features, B = x
    spatial = list(image_size[1:])
    fs = cfg.filter_size
    c = np.asarray(coords).astype(np.int64)
    m = active_mask(c, spatial, B)
    h = densify(c, features, spatial, B)

    def conv(h, m, w, b, size):
        y = F.conv3d(h, to_torch_kernel(w, size), padding=[s // 2 for s in size])
        if b is not None:
            y = y + b[None, :, None, None, None]
        return y * m[:, None].to(y.dtype)

    def series(h, m, prefix):
        for j in range(cfg.blocks_per_layer):
            p = "%s.block_%d" % (prefix, j)
            res = h
            o = conv(h, m, sd[p + ".convolution_1.conv1.weight"], sd.get(p + ".convolution_1.conv1.bias"), [fs] * 3)
            o = _leaky(_masked_bn(o, m, sd[p + ".convolution_1.norm.weight"], sd[p + ".convolution_1.norm.bias"]))
            o = conv(o, m, sd[p + ".convolution_2.conv1.weight"], sd.get(p + ".convolution_2.conv1.bias"), [fs] * 3)
            o = _masked_bn(o, m, sd[p + ".convolution_2.norm.weight"], sd[p + ".convolution_2.norm.bias"])
            h = _leaky(o + res)
        return h

    h = conv(h, m, sd["initial_convolution.weight"], sd.get("initial_convolution.bias"), [5, 5, 5])
    for l in range(cfg.depth):
        h = series(h, m, "network_layers.%d" % (2 * l))
        p = "network_layers.%d" % (2 * l + 1)
        y = F.conv3d(h, to_torch_kernel(sd[p + ".conv.weight"], [2, 2, 2]), stride=2)
        m = F.conv3d(m.float()[:, None], torch.ones(1, 1, 2, 2, 2), stride=2)[:, 0] > 0
        h = _leaky(_masked_bn(y * m[:, None].to(y.dtype), m, sd[p + ".norm.weight"], sd[p + ".norm.bias"]))
    h = series(h, m, "final_layer")
    h = conv(h, m, sd["bottleneck.weight"], sd.get("bottleneck.bias"), [1, 1, 1])
    return torch.tanh(h)
