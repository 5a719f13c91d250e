"""GPU parity of the reference's OWN (unfused) call sequence on the engine's scn module surface.

INTEGRATION.md 1 promises that swapping ``import sparseconvnet as scn`` for ``sparseeventid_amd.scn`` lets the
reference's network code run unchanged.  The product's networks.py always routes through fused kernels
(BN+LeakyReLU, BN+AddTable+LeakyReLU, the one-node residual block); this test runs what the import swap would
actually execute: ``Block.forward`` = conv1 -> norm -> activation, ``ResidualBlock.forward`` = Block -> Block ->
AddTable([out, residual]) -> LeakyReLU, ``ConvolutionDownsample.forward`` = conv -> norm -> relu
(/root/reference/src/networks/sparse_building_blocks.py:47-57, 85-100, 130-139), as separate scn modules with their
own autograd nodes, over a whole depth-2 encoder with backward, against the oracle.
"""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import elem_close, blob_cloud, rel_err

pytestmark = pytest.mark.gpu
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def _reference_style_classes():
    from sparseeventid_amd import networks

    class RefBlock(networks.Block):
        def forward(self, x):
            out = self.conv1(x)
            if self._do_normalization:
                out = self.norm(out)            # scn.BatchNormalization module (leakiness 1)
            return self.activation(out)         # scn.LeakyReLU / scn.Identity module

    class RefResidualBlock(networks.ResidualBlock):
        def forward(self, x):
            residual = self.residual(x)
            out = self.convolution_1(x)
            out = self.convolution_2(out)
            out = self.add([out, residual])     # scn.AddTable
            return self.relu(out)               # scn.LeakyReLU

    class RefConvolutionDownsample(networks.ConvolutionDownsample):
        def forward(self, x):
            out = self.conv(x)
            if self._do_normalization:
                out = self.norm(out)
            return self.relu(out)

    return {networks.Block: RefBlock, networks.ResidualBlock: RefResidualBlock,
            networks.ConvolutionDownsample: RefConvolutionDownsample}


def _build(kw, image_size, unfused):
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import Config, Encoder
    enc, head = build_networks(Config(encoder=Encoder(**kw)), image_size, HEADS)
    if unfused:
        swap = _reference_style_classes()
        n = 0
        for m in enc.modules():
            if type(m) in swap:
                m.__class__ = swap[type(m)]
                n += 1
        assert n > 0
    return enc.cuda(), head.cuda()


def test_reference_style_unfused_encoder_matches_oracle_and_fused_path():
    from sparseeventid_amd.loss import focal_loss
    kw = dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=2)
    image_size = [1, 32, 28, 36]
    ocfg = scn_cpu.NetConfig(**kw)
    sd = scn_cpu.init_params(ocfg, seed=8, heads=HEADS)
    g = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("norm.bias") or k.endswith("conv1.bias") or k == "initial_convolution.bias":
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    B = 3
    coords, feats = blob_cloud(78, B, image_size[1:], 400, 1)
    labels = {k: torch.randint(0, n, (B,), generator=g) for k, n in HEADS.items()}

    results = {}
    for unfused in (True, False):
        enc, head = _build(kw, image_size, unfused)
        enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
        head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
        enc.train(); head.eval()
        dense = enc((torch.from_numpy(coords), torch.from_numpy(feats), B))
        logits = head(dense)
        loss = focal_loss(logits, {k: v.cuda() for k, v in labels.items()})
        loss.backward()
        named = dict(enc.named_parameters())
        named.update(dict(head.named_parameters()))
        results[unfused] = (dense.detach(), {k: v.detach() for k, v in logits.items()}, float(loss.detach()),
                            {k: p.grad.detach().clone() for k, p in named.items()},
                            {k: b.detach().clone() for k, b in enc.named_buffers()})

    osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
           for k, v in sd.items()}
    odense = scn_cpu.encoder_forward(osd, ocfg, (coords, torch.from_numpy(feats), B), image_size)
    ologits = scn_cpu.head_forward(osd, odense, HEADS)
    oloss = scn_cpu.focal_loss(ologits, labels)
    oloss.backward()

    dense, logits, loss, grads, bufs = results[True]
    assert rel_err(dense, odense) < 1e-4
    for k in HEADS:
        assert rel_err(logits[k], ologits[k]) < 1e-4
        assert elem_close(logits[k], ologits[k])[0], (k, elem_close(logits[k], ologits[k])[1])   # every logit
    assert abs(loss - float(oloss.detach())) < 1e-4 * max(1.0, abs(float(oloss.detach())))
    gmax = max(float(osd[k].grad.abs().max()) for k in grads)
    for k, gk in grads.items():
        og = osd[k].grad
        if float(og.abs().max()) < 1e-5 * gmax:      # conv bias in front of BatchNorm: analytically zero
            assert float(gk.abs().max()) < 1e-4 * gmax, k
            continue
        assert rel_err(gk, og) < 1e-4, (k, rel_err(gk, og))
    for k, b in bufs.items():
        assert rel_err(b, osd[k]) < 1e-4, k
    # and the fused product path computes the same function (different kernels, same arithmetic per element)
    fdense, flogits, floss, fgrads, _ = results[False]
    assert rel_err(dense, fdense) < 1e-5
    for k, gk in grads.items():
        if float(osd[k].grad.abs().max()) >= 1e-5 * gmax:
            assert rel_err(gk, fgrads[k]) < 1e-5, k
