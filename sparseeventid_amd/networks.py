"""The reference's network topology restated on the MI355X engine, with IDENTICAL module tree /
state_dict keys so checkpoints interchange (create_trainer.py:96-103 loads ``encoder.*`` keys):

  Encoder            /root/reference/src/networks/resnet.py:10-161
  Block, ResidualBlock, ConvolutionDownsample, BlockSeries
                     /root/reference/src/networks/sparse_building_blocks.py:18-139,231-264
  multi_head_output, create_final_dense_chain, build_networks
                     /root/reference/src/networks/classification_head.py:7-55

Only the constructible branch of the reference is built (sparse + batch norm + convolutional
downsampling).  Forward passes call the engine's fused BN(+residual)+LeakyReLU
kernels where the reference chains scn.BatchNormalization -> scn.LeakyReLU / scn.AddTable; the
module tree (and therefore the parameters) is unchanged.
"""
import torch
from torch import nn

from . import scn
from .config import DataMode, DownSampling, GrowthRate, Norm


def _enum_name(v):
    return getattr(v, "name", v)


class Block(nn.Module):
    """subm-conv -> BatchNormalization -> activation (sparse_building_blocks.py:18-57)."""

    def __init__(self, *, nIn, nOut, dim, params, activation=scn.LeakyReLU):
        super().__init__()
        kernel = [1, params.filter_size, params.filter_size] if dim == 2 else 3 * [params.filter_size]
        self.conv1 = scn.SubmanifoldConvolution(dimension=3, nIn=nIn, nOut=nOut, filter_size=kernel, bias=params.bias)
        self._do_normalization = False
        if _enum_name(params.normalization) == "batch":
            self._do_normalization = True
            self.norm = scn.BatchNormalization(nOut)
        elif _enum_name(params.normalization) != "none":
            raise NotImplementedError("only Norm.batch is constructible in the reference (SURVEY.md 2.3)")
        self.activation = activation()

    def forward(self, x, residual=None):
        leak = self.activation.leak if isinstance(self.activation, scn.LeakyReLU) else None
        if (self._do_normalization and (leak is not None or isinstance(self.activation, scn.Identity))
                and x.features.is_cuda):
            # conv -> BN + activation as one node (Identity activation -> leak 1); BN statistics from the conv epilogue
            return scn.conv_bn_act(x, self.conv1, self.norm, 1.0 if leak is None else leak)
        out = self.conv1(x)
        if self._do_normalization:
            out = self.norm(out)
        return self.activation(out)


class ResidualBlock(nn.Module):
    """Block -> Block(no act) -> AddTable -> LeakyReLU (sparse_building_blocks.py:61-100)."""

    def __init__(self, *, nIn, nOut, dim, params):
        super().__init__()
        self.convolution_1 = Block(nIn=nIn, nOut=nOut, dim=dim, params=params)
        self.convolution_2 = Block(nIn=nIn, nOut=nOut, dim=dim, activation=scn.Identity, params=params)
        self.residual = scn.Identity()
        self.relu = scn.LeakyReLU()
        self.add = scn.AddTable()

    def forward(self, x):
        c1, c2 = self.convolution_1, self.convolution_2
        if (c1._do_normalization and c2._do_normalization and isinstance(c1.activation, scn.LeakyReLU)
                and c1.norm.affine and c2.norm.affine and c1.conv1.nIn == c1.conv1.nOut == c2.conv1.nOut
                and c1.conv1.filter_size == c2.conv1.filter_size and scn.residual_block_supported(c1.conv1.nOut)
                and x.features.shape[0] > 0 and x.features.is_cuda):
            # whole block as one autograd node (skip gradient folded into conv1's dX kernel)
            return scn.residual_block(x, c1.conv1, c1.norm, c2.conv1, c2.norm, c1.activation.leak, self.relu.leak)
        residual = self.residual(x)
        out = self.convolution_1(x)
        if c2._do_normalization and x.features.is_cuda:
            # conv -> [BN + residual add + LeakyReLU] fused: one read of each operand, one write
            return scn.conv_bn_act(out, c2.conv1, c2.norm, self.relu.leak, residual=residual.features)
        out = c2(out)
        out = self.add([out, residual])
        return self.relu(out)


class ConvolutionDownsample(nn.Module):
    """scn.Convolution(2,s2) -> BN -> LeakyReLU (sparse_building_blocks.py:103-139)."""

    def __init__(self, *, nIn, nOut, dim, params):
        super().__init__()
        filter_size = [1, 2, 2] if dim == 2 else [2, 2, 2]
        self.conv = scn.Convolution(dimension=3, nIn=nIn, nOut=nOut, filter_size=filter_size,
                                    filter_stride=filter_size, bias=False)
        self._do_normalization = False
        if _enum_name(params.normalization) == "batch":
            self._do_normalization = True
            self.norm = scn.BatchNormalization(nOut)
        elif _enum_name(params.normalization) != "none":
            raise NotImplementedError("only Norm.batch is constructible in the reference (SURVEY.md 2.3)")
        self.relu = scn.LeakyReLU()

    def forward(self, x):
        if self._do_normalization and x.features.is_cuda:
            return scn.conv_bn_act(x, self.conv, self.norm, self.relu.leak)
        out = self.conv(x)
        if self._do_normalization:
            return self.relu(self.norm(out))
        return self.relu(out)


class BlockSeries(nn.Module):
    """n_blocks ResidualBlocks sharing one rulebook (sparse_building_blocks.py:231-264)."""

    def __init__(self, *, nIn, n_blocks, dim, params):
        super().__init__()
        cls = ResidualBlock if params.residual else Block
        self.blocks = [cls(nIn=nIn, nOut=nIn, dim=dim, params=params) for _ in range(n_blocks)]
        for i, block in enumerate(self.blocks):
            self.add_module("block_{}".format(i), block)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class Encoder(nn.Module):
    """resnet.py:10-161.  ``image_size``: 3-D -> [1, D0, D1, D2] (uses [1:]), 2-D -> [planes, H, W]."""

    def __init__(self, params, image_size):
        super().__init__()
        dimension = params.data.dimension
        if _enum_name(params.framework.mode) != "sparse":
            raise NotImplementedError("only framework.mode=sparse is constructible in the reference (SURVEY.md 2.3)")
        enc = params.encoder
        current = enc.n_initial_filters
        image_size = [int(i) for i in image_size]
        if dimension == 2:
            self.input_layer = scn.InputLayer(dimension=3, spatial_size=torch.tensor(image_size))
            self.initial_convolution = scn.SubmanifoldConvolution(
                dimension=3, nIn=1, nOut=enc.n_initial_filters, filter_size=[1, 5, 5], bias=enc.bias)
        else:
            self.input_layer = scn.InputLayer(dimension=3, spatial_size=torch.tensor(image_size[1:]))
            self.initial_convolution = scn.SubmanifoldConvolution(
                dimension=3, nIn=1, nOut=enc.n_initial_filters, filter_size=[5, 5, 5], bias=enc.bias)
        if _enum_name(enc.downsampling) != "convolutional":
            raise NotImplementedError("only DownSampling.convolutional is constructible in the reference")
        self.network_layers = nn.ModuleList()
        for _ in range(enc.depth):
            self.network_layers.append(BlockSeries(nIn=current, n_blocks=enc.blocks_per_layer, dim=dimension, params=enc))
            nxt = self.increase_filters(current, enc)
            self.network_layers.append(ConvolutionDownsample(nIn=current, nOut=nxt, dim=dimension, params=enc))
            current = nxt
        self.final_layer = BlockSeries(nIn=current, n_blocks=enc.blocks_per_layer, dim=dimension, params=enc)
        self.bottleneck = scn.SubmanifoldConvolution(dimension=3, nIn=current, nOut=enc.n_output_filters,
                                                     filter_size=1, bias=enc.bias)
        final_shape = [i // 2 ** enc.depth for i in image_size]
        if dimension == 2:
            final_shape[0] = 3
        else:
            final_shape = final_shape[1:]
        self.output_shape = [enc.n_output_filters] + final_shape
        self.pool = nn.Sequential(scn.SparseToDense(dimension=3, nPlanes=self.output_shape[0]))

    def forward(self, x):
        x = self.input_layer(x)
        x = self.initial_convolution(x)
        for l in self.network_layers:
            x = l(x)
        x = self.final_layer(x)
        x = self.bottleneck(x)
        x = self.pool(x)
        return torch.tanh(x)

    def forward_pooled(self, x):
        """Opt-in fused sparse head: returns AvgPool3d_full(forward(x)) as [B, C, 1, 1, 1] without materialising
        the dense tensor (tanh(0) = 0, so only active rows contribute).  ``multi_head_output.forward_pooled``
        consumes it.  The default ``forward`` keeps the reference's dense output."""
        x = self.input_layer(x)
        x = self.initial_convolution(x)
        for l in self.network_layers:
            x = l(x)
        x = self.final_layer(x)
        x = self.bottleneck(x)
        if [int(v) for v in x.spatial_size] != [int(v) for v in self.output_shape[1:]]:
            raise RuntimeError("fused head: final spatial size %r differs from output_shape %r"
                               % (x.spatial_size, self.output_shape))
        return scn.pool_tanh(x)[:, :, None, None, None]

    def increase_filters(self, current_filters, params):
        if _enum_name(params.growth_rate) == "multiplicative":
            return current_filters * 2
        return current_filters + params.n_initial_filters


def create_final_dense_chain(spatial_shape, n_in, n_out):
    """classification_head.py:19-28 (dense torch head; keys .2 and .5 hold the Linears)."""
    return nn.Sequential(
        nn.AvgPool3d(spatial_shape),
        nn.Flatten(start_dim=1, end_dim=-1),
        nn.Linear(in_features=n_in, out_features=256),
        nn.Dropout(),
        nn.LeakyReLU(),
        nn.Linear(in_features=256, out_features=n_out),
    )


class multi_head_output(nn.Module):
    """classification_head.py:7-17."""

    def __init__(self, spatial_shape, n_in, output_shape):
        super().__init__()
        self.classification_head = nn.ModuleDict(
            {key: create_final_dense_chain(spatial_shape, n_in, output_shape[key]) for key in output_shape.keys()})

    def forward(self, x):
        heads = self.classification_head
        first = [heads[k][0] for k in heads.keys()]
        if x.dim() == 5 and all(isinstance(p, nn.AvgPool3d) and _full_window(p, x) for p in first):
            # every head opens with the same global AvgPool3d: pool the dense tensor ONCE (a full-window
            # average is the mean over the spatial dims) and hand it to the rest of each chain.
            pooled = x.mean(dim=(2, 3, 4), keepdim=True)
            return {key: heads[key][1:](pooled) for key in heads.keys()}
        return {key: heads[key](x) for key in heads.keys()}


    def forward_pooled(self, pooled):
        heads = self.classification_head
        return {key: heads[key][1:](pooled) for key in heads.keys()}


def _full_window(pool, x):
    ks = pool.kernel_size
    ks = (ks, ks, ks) if isinstance(ks, int) else tuple(ks)
    return tuple(int(k) for k in ks) == tuple(x.shape[2:]) and pool.padding in (0, (0, 0, 0))


def build_networks(params, input_shape, output_shape):
    """classification_head.py:30-55."""
    encoder = Encoder(params, input_shape)
    n_filters = encoder.output_shape[0]
    spatial_shape = encoder.output_shape[1:]
    if isinstance(output_shape, dict):
        head = multi_head_output(spatial_shape, n_filters, output_shape)
    else:
        head = create_final_dense_chain(spatial_shape, n_filters, output_shape)
    return encoder, head
