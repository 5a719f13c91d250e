"""SBND 2-D multiplane shared-weight sparse ResNet (BASELINE configs[4]) restated on the engine with the
module tree / state_dict keys of the reference's legacy model
(/root/reference/src/networks/torch/sparseresnet.py:12-334):

  SparseBlock :12-43, SparseResidualBlock :47-103, SparseConvolutionDownsample :108-137,
  SparseBlockSeries :139-158, ResNet :194-334.

Planes are the first "spatial" dimension ([3, 2048, 1280]); pre-merge layers use [1,3,3] filters (weights
shared across planes, no mixing), post-merge layers [3,3,3] (planes mix); downsampling is [1,2,2]/[1,2,2].
The reference never defines ``args.network.*``; defaults here follow the run names in its
notebooks (nf32, bpl2).
"""
from dataclasses import dataclass

import torch
from torch import nn

from . import scn


@dataclass
class SBNDNetworkConfig:
    n_initial_filters: int = 32
    res_blocks_per_layer: int = 2
    network_depth: int = 6
    depth_pre_merge: int = 3
    batch_norm: bool = True
    leaky_relu: bool = True
    spatial_size: tuple = (3, 2048, 1280)


def _fused(bn, x, leak, residual=None):
    out = scn.SparseConvNetTensor(metadata=x.metadata, spatial_size=x.spatial_size)
    out.features = bn.forward_features(x.features, leak=leak, residual=residual)
    return out


class SparseBlock(nn.Module):
    def __init__(self, inplanes, outplanes, batch_norm, leaky_relu, nplanes=1):
        super().__init__()
        self.batch_norm, self.leaky_relu = batch_norm, leaky_relu
        self.conv1 = scn.SubmanifoldConvolution(dimension=3, nIn=inplanes, nOut=outplanes,
                                                filter_size=[nplanes, 3, 3], bias=False)
        if batch_norm:
            self.bn1 = scn.BatchNormLeakyReLU(outplanes) if leaky_relu else scn.BatchNormReLU(outplanes)
        else:
            self.relu = scn.LeakyReLU() if leaky_relu else scn.ReLU()

    def forward(self, x):
        out = self.conv1(x)
        return self.bn1(out) if self.batch_norm else self.relu(out)


class SparseResidualBlock(nn.Module):
    def __init__(self, inplanes, outplanes, batch_norm, leaky_relu, nplanes=1):
        super().__init__()
        self.batch_norm, self.leaky_relu = batch_norm, leaky_relu
        self.conv1 = scn.SubmanifoldConvolution(dimension=3, nIn=inplanes, nOut=outplanes,
                                                filter_size=[nplanes, 3, 3], bias=False)
        if batch_norm:
            self.bn1 = scn.BatchNormLeakyReLU(outplanes) if leaky_relu else scn.BatchNormReLU(outplanes)
        self.conv2 = scn.SubmanifoldConvolution(dimension=3, nIn=outplanes, nOut=outplanes,
                                                filter_size=[nplanes, 3, 3], bias=False)
        if batch_norm:
            self.bn2 = scn.BatchNormalization(outplanes)
        self.residual = scn.Identity()
        self.relu = scn.LeakyReLU() if leaky_relu else scn.ReLU()
        self.add = scn.AddTable()

    def forward(self, x):
        if (self.batch_norm and self.conv1.nIn == self.conv1.nOut and scn.residual_block_supported(self.conv1.nOut)
                and x.features.shape[0] > 0 and x.features.is_cuda):
            return scn.residual_block(x, self.conv1, self.bn1, self.conv2, self.bn2, self.bn1.leakiness, self.relu.leak)
        residual = self.residual(x)
        out = self.conv1(x)
        out = self.bn1(out) if self.batch_norm else self.relu(out)
        out = self.conv2(out)
        if self.batch_norm:
            # BatchNormalization -> AddTable -> (Leaky)ReLU as one fused pass
            return _fused(self.bn2, out, self.relu.leak, residual.features)
        out = self.add([out, residual])
        return self.relu(out)


class SparseConvolutionDownsample(nn.Module):
    def __init__(self, inplanes, outplanes, batch_norm, leaky_relu, nplanes=1):
        super().__init__()
        self.batch_norm, self.leaky_relu = batch_norm, leaky_relu
        self.conv = scn.Convolution(dimension=3, nIn=inplanes, nOut=outplanes, filter_size=[nplanes, 2, 2],
                                    filter_stride=[1, 2, 2], bias=False)
        if batch_norm:
            self.bn = scn.BatchNormalization(outplanes)
        self.relu = scn.LeakyReLU() if leaky_relu else scn.ReLU()

    def forward(self, x):
        out = self.conv(x)
        if self.batch_norm:
            return _fused(self.bn, out, self.relu.leak)
        return self.relu(out)


class SparseBlockSeries(nn.Module):
    def __init__(self, inplanes, n_blocks, nplanes, batch_norm, leaky_relu, residual=False):
        super().__init__()
        cls = SparseResidualBlock if residual else SparseBlock
        self.blocks = [cls(inplanes, inplanes, batch_norm, leaky_relu, nplanes=nplanes) for _ in range(n_blocks)]
        for i, block in enumerate(self.blocks):
            self.add_module("block_{}".format(i), block)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


def filter_increase(input_filters, initial_filters):
    return input_filters + initial_filters


class ResNet(nn.Module):
    """sparseresnet.py:194-334.  ``output_shape``: {key: [..., n_classes]}; ``args``: SBNDNetworkConfig or any
    object exposing ``args.network.{n_initial_filters, depth_pre_merge, network_depth, res_blocks_per_layer,
    batch_norm, leaky_relu}``."""

    def __init__(self, output_shape, args=None):
        super().__init__()
        net = getattr(args, "network", args) or SBNDNetworkConfig()
        spatial = list(getattr(net, "spatial_size", (3, 2048, 1280)))
        self.input_tensor = scn.InputLayer(dimension=3, spatial_size=spatial)
        nf0 = net.n_initial_filters
        self.initial_convolution = scn.SubmanifoldConvolution(dimension=3, nIn=1, nOut=nf0, filter_size=[1, 5, 5],
                                                              bias=False)
        n_filters = nf0
        bn, lr, bpl = net.batch_norm, net.leaky_relu, net.res_blocks_per_layer
        self.pre_convolutional_layers = nn.ModuleList()
        for _ in range(net.depth_pre_merge):
            out_filters = filter_increase(n_filters, nf0)
            self.pre_convolutional_layers.append(SparseBlockSeries(n_filters, bpl, 1, bn, lr, residual=True))
            self.pre_convolutional_layers.append(SparseConvolutionDownsample(n_filters, out_filters, bn, lr, nplanes=1))
            n_filters = out_filters
        self.post_convolutional_layers = nn.ModuleList()
        for _ in range(net.network_depth - net.depth_pre_merge):
            out_filters = filter_increase(n_filters, nf0)
            self.post_convolutional_layers.append(SparseBlockSeries(n_filters, bpl, 3, bn, lr, residual=True))
            self.post_convolutional_layers.append(SparseConvolutionDownsample(n_filters, out_filters, bn, lr, nplanes=1))
            n_filters = out_filters
        self.final_layer = nn.ModuleDict({
            key: nn.Sequential(
                SparseBlockSeries(n_filters, bpl, 3, bn, lr, residual=True),
                scn.SubmanifoldConvolution(dimension=3, nIn=n_filters, nOut=output_shape[key][-1], filter_size=1,
                                           bias=False),
                scn.SparseToDense(dimension=3, nPlanes=output_shape[key][-1]))
            for key in output_shape})

    def forward(self, x):
        # a prepared input-level tensor (Trainer builds the InputLayer itself to batch the rulebook builds) passes through
        batch_size = x.batch_size() if isinstance(x, scn.SparseConvNetTensor) else x[-1]
        x = self.input_tensor(x)
        x = self.initial_convolution(x)
        for l in self.pre_convolutional_layers:
            x = l(x)
        for l in self.post_convolutional_layers:
            x = l(x)
        output = {}
        for key in self.final_layer:
            dense = self.final_layer[key](x)
            # global AvgPool3d over the whole (plane, H, W) extent == mean over those dims
            output[key] = dense.mean(dim=(2, 3, 4)).view([batch_size, dense.shape[1]])
        return output
