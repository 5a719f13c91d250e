"""GPU: the inference path (eval-mode BatchNorm, nothing to differentiate -- the reference's `trainer.validate`,
bin/exec.py:206-224): conv -> BN (+ residual) -> LeakyReLU is ONE kernel (sei_conv_bn_inference: the normalisation, the
residual add and the activation are applied while the convolution kernel stores its finished tile).  It evaluates the
same expression as the two-pass eval path (convolution, then the BatchNorm apply kernel), which the oracle tests cover:
results must be bit-identical -- per block for every kernel family the plans use (wave 32 / 64 planes, tile 96-192
planes, the small-nIn stem kernel, the strided convolution), for the two-pass fallback of plans without the epilogue,
and for the whole network."""
import pytest
import torch

from tests.util import blob_cloud

pytestmark = pytest.mark.gpu


def _params():
    from sparseeventid_amd.config import Encoder
    return Encoder()


def _cloud(seed, B, sp, n, planes):
    import sparseeventid_amd.scn as scn
    coords, feats = blob_cloud(seed, B, sp, n, 1)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), torch.from_numpy(feats).cuda(), B))
    g = torch.Generator(device="cuda").manual_seed(seed)
    t.features = torch.randn(t.features.shape[0], planes, device="cuda", generator=g)
    return t


def _randomise_bn(module, seed):
    import sparseeventid_amd.scn as scn
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, scn.BatchNormalization):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, device="cuda", generator=g) * 0.3)
                m.running_var.copy_(torch.rand(m.running_var.shape, device="cuda", generator=g) + 0.5)
                if m.weight is not None:
                    m.weight.copy_(torch.rand(m.weight.shape, device="cuda", generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.bias.shape, device="cuda", generator=g) * 0.2)


def _both_ways(module, t):
    """eval mode: the differentiable path (two passes: conv, BN apply) and the inference path (one kernel)."""
    import sparseeventid_amd.scn as scn
    module.eval()
    x = t.features
    two = module(scn.SparseConvNetTensor(x.clone().requires_grad_(True), t.metadata, t.spatial_size))
    with torch.no_grad():
        one = module(scn.SparseConvNetTensor(x.clone(), t.metadata, t.spatial_size))
    assert two.features.requires_grad and not one.features.requires_grad
    return two.features.detach(), one.features


@pytest.mark.parametrize("C", [32, 64, 96, 128, 160, 192])
def test_residual_block_inference_is_bit_identical_to_the_two_pass_eval_path(C):
    from sparseeventid_amd.networks import ResidualBlock
    t = _cloud(3 + C, 3, [28, 26, 30], 2200, C)
    assert t.features.shape[0] % 64 != 0
    torch.manual_seed(C)
    blk = ResidualBlock(nIn=C, nOut=C, dim=3, params=_params()).cuda()
    _randomise_bn(blk, C)
    two, one = _both_ways(blk, t)
    assert torch.equal(two, one)
    assert float(one.abs().max()) > 0 and bool((one < 0).any())      # LeakyReLU really applied to both signs


@pytest.mark.parametrize("C", [64, 128, 160, 192])
def test_residual_block_inference_through_the_round_kernel(C, force_conv_kernel):
    """The two-offset-round form of the tile kernel applies the same epilogue; small clouds only reach it when it is pinned
    (the default takes it for 96 columns at any size, for 128 / 160 from 100 000 rows on)."""
    from sparseeventid_amd.networks import ResidualBlock
    t = _cloud(5 + C, 3, [28, 26, 30], 2200, C)
    torch.manual_seed(C)
    blk = ResidualBlock(nIn=C, nOut=C, dim=3, params=_params()).cuda()
    _randomise_bn(blk, C)
    force_conv_kernel("tile")
    two_tile, one_tile = _both_ways(blk, t)
    force_conv_kernel("rounds")
    two, one = _both_ways(blk, t)
    assert torch.equal(two, one) and torch.equal(one, one_tile) and torch.equal(two, two_tile)


def test_stem_block_and_downsample_inference():
    """1 -> 32 planes 5^3 (the small-nIn kernel) and the strided 2^3 convolution 64 -> 96 (tile kernel, K = 8)."""
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.config import Encoder
    from sparseeventid_amd.networks import Block, ConvolutionDownsample
    t = _cloud(11, 2, [30, 30, 30], 2500, 1)
    torch.manual_seed(1)
    p = Encoder(filter_size=5)
    stem = Block(nIn=1, nOut=32, dim=3, params=p).cuda()
    _randomise_bn(stem, 5)
    two, one = _both_ways(stem, t)
    assert torch.equal(two, one)
    t2 = _cloud(12, 2, [30, 30, 30], 2500, 64)
    down = ConvolutionDownsample(nIn=64, nOut=96, dim=3, params=_params()).cuda()
    _randomise_bn(down, 6)
    two, one = _both_ways(down, t2)
    assert torch.equal(two, one) and one.shape[1] == 96 and one.shape[0] < t2.features.shape[0]


def test_inference_falls_back_to_two_passes_for_plans_without_the_epilogue(force_conv_kernel):
    """Odd plane counts run the scalar kernels and a pinned row-band family has no epilogue: sei_conv_bn_inference then runs
    the convolution into scratch and the BatchNorm apply kernel behind it -- same results."""
    import sparseeventid_amd.scn as scn
    t = _cloud(21, 2, [24, 24, 24], 1500, 20)
    conv = scn.SubmanifoldConvolution(3, 20, 12, 3, True).cuda()
    bn = scn.BatchNormalization(12).cuda()
    _randomise_bn(bn, 9)
    bn.eval()
    xin = scn.SparseConvNetTensor(t.features.clone().requires_grad_(True), t.metadata, t.spatial_size)
    two = scn.conv_bn_act(xin, conv, bn, 0.25).features.detach()
    with torch.no_grad():
        one = scn.conv_bn_act(scn.SparseConvNetTensor(t.features.clone(), t.metadata, t.spatial_size), conv, bn, 0.25).features
    assert torch.equal(two, one)
    from tests.conftest import conv_families
    if "band" not in conv_families():
        return   # the row-band family (experiment builds only) is the other plan without the epilogue
    t96 = _cloud(22, 2, [24, 24, 24], 1500, 96)
    conv = scn.SubmanifoldConvolution(3, 96, 96, 3, False).cuda()
    bn = scn.BatchNormalization(96).cuda()
    _randomise_bn(bn, 10)
    bn.eval()
    res = {}
    for fam in ("tile", "band"):
        force_conv_kernel(fam)
        with torch.no_grad():
            res[fam] = scn.conv_bn_act(scn.SparseConvNetTensor(t96.features.clone(), t96.metadata, t96.spatial_size), conv, bn,
                                       0.333, residual=t96.features).features
    assert torch.equal(res["tile"], res["band"])


def test_whole_network_inference_equals_the_differentiable_eval_forward():
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.networks import build_networks
    cfg = Config(encoder=Encoder(depth=3, n_initial_filters=32, n_output_filters=32, blocks_per_layer=2))
    torch.manual_seed(4)
    enc, head = build_networks(cfg, [1, 64, 64, 64], {"labelneutID": 3, "labelprotID": 3})
    enc.cuda().eval(); head.cuda().eval()
    _randomise_bn(enc, 13)
    coords, feats = blob_cloud(31, 3, [64, 64, 64], 3000, 1)
    image = (torch.from_numpy(coords), torch.from_numpy(feats).cuda(), 3)
    f2 = torch.from_numpy(feats).cuda().requires_grad_(True)
    two = head(enc((torch.from_numpy(coords), f2, 3)))
    with torch.no_grad():
        one = head(enc(image))
    for k in two:
        assert torch.equal(two[k].detach(), one[k]), k


def test_inference_forward_matches_the_oracle_in_eval_mode():
    """The one-kernel inference path against the CPU restatement run with training=False (running statistics): dense
    encoder output and logits within the forward tolerance of every other oracle test (1e-4, max-norm relative)."""
    from oracle import scn_cpu
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.networks import build_networks
    from tests.util import rel_err
    heads = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
    kw = dict(depth=2, n_initial_filters=32, n_output_filters=32, blocks_per_layer=2)
    image_size = [1, 36, 28, 44]
    ocfg = scn_cpu.NetConfig(dimension=3, **kw)
    sd = scn_cpu.init_params(ocfg, seed=5, heads=heads)
    g = torch.Generator().manual_seed(6)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("norm.bias") or k.endswith("conv1.bias") or k == "initial_convolution.bias":
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("running_mean"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
    enc, head = build_networks(Config(encoder=Encoder(**kw)), image_size, heads)
    enc.cuda(); head.cuda()
    enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
    head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
    enc.eval(); head.eval()
    B = 3
    coords, feats = blob_cloud(78, B, image_size[1:], 500, 1)
    with torch.no_grad():
        dense = enc((torch.from_numpy(coords), torch.from_numpy(feats).cuda(), B))
        logits = head(dense)
        odense = scn_cpu.encoder_forward(sd, ocfg, (coords, torch.from_numpy(feats), B), image_size, training=False)
        ologits = scn_cpu.head_forward(sd, odense, heads)
    assert rel_err(dense, odense) < 1e-4
    for k in heads:
        assert rel_err(logits[k], ologits[k]) < 1e-4
