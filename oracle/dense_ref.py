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
