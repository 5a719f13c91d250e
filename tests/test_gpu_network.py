"""GPU parity: the whole hot path (Encoder.forward -> head -> focal loss -> backward) on the engine
vs the CPU oracle, same weights, same seeded input.  Tolerance 1e-4 relative fp32 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import elem_close, blob_cloud, rel_err

pytestmark = pytest.mark.gpu
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def _build(cfg_kw, image_size, dimension=3):
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import Config, Data, Encoder
    cfg = Config(encoder=Encoder(**cfg_kw), data=Data(dimension=dimension))
    enc, head = build_networks(cfg, image_size, HEADS)
    return enc.cuda(), head.cuda()


@pytest.mark.parametrize("dimension,image_size,kw", [
    (3, [1, 32, 32, 32], dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=2)),
    (3, [1, 36, 28, 44], dict(depth=2, n_initial_filters=32, n_output_filters=32, blocks_per_layer=1)),
    (2, [3, 32, 48], dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1)),
])
def test_encoder_head_loss_parity(dimension, image_size, kw):
    torch.manual_seed(0)
    ocfg = scn_cpu.NetConfig(dimension=dimension, **kw)
    sd = scn_cpu.init_params(ocfg, seed=5, heads=HEADS)
    # perturb BN affine so gamma/beta gradients are exercised away from the init point
    g = torch.Generator().manual_seed(6)
    for k in sd:
        if k.endswith("norm.weight"):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("norm.bias") or k.endswith("conv1.bias") or k == "initial_convolution.bias":
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    enc, head = _build(kw, image_size, dimension)
    enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
    head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
    enc.train(); head.eval()

    B = 3
    spatial = image_size[1:] if dimension == 3 else image_size
    coords, feats = blob_cloud(77, B, spatial, 400, 1)
    labels = {k: torch.randint(0, n, (B,), generator=g) for k, n in HEADS.items()}

    dense = enc((torch.from_numpy(coords), torch.from_numpy(feats), B))
    logits = head(dense)
    from sparseeventid_amd.loss import focal_loss
    loss = focal_loss(logits, {k: v.cuda() for k, v in labels.items()})
    loss.backward()

    osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
           for k, v in sd.items()}
    odense = scn_cpu.encoder_forward(osd, ocfg, (coords, torch.from_numpy(feats), B), image_size)
    ologits = scn_cpu.head_forward(osd, odense, HEADS)
    oloss = scn_cpu.focal_loss(ologits, labels)
    oloss.backward()

    assert dense.shape == odense.shape
    assert rel_err(dense, odense) < 1e-4
    for k in HEADS:
        assert rel_err(logits[k], ologits[k]) < 1e-4
        assert elem_close(logits[k], ologits[k])[0], (k, elem_close(logits[k], ologits[k])[1])   # every logit
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-4 * max(1.0, abs(float(oloss.detach())))
    named = dict(enc.named_parameters())
    named.update({"classification_head." + k[len("classification_head."):] if k.startswith("classification_head.") else k: v
                  for k, v in head.named_parameters()})
    # a conv bias feeding BatchNorm has an analytically ZERO gradient (BN removes the mean); both sides
    # hold rounding noise there, so such tensors are checked against the global gradient scale instead.
    gmax = max(float(osd[k].grad.abs().max()) for k in named)
    for k, p in named.items():
        og = osd[k].grad
        assert p.grad is not None and og is not None, k
        if float(og.abs().max()) < 1e-5 * gmax:
            assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            continue
        e = rel_err(p.grad, og)
        assert e < 1e-4, (k, e)
    # running statistics (BN buffers) agree too
    for k, b in enc.named_buffers():
        assert rel_err(b, osd[k]) < 1e-4, k


def test_fused_sparse_head_equals_dense_path_and_oracle():
    """Opt-in fused head: sum_active tanh(x)/V == AvgPool3d(tanh(SparseToDense(x))) -- against the
    engine's own dense path AND against the oracle (pooled activations, logits, bottleneck gradient), and
    bit-identical between two runs (fixed-order reduction, no float atomics)."""
    torch.manual_seed(1)
    kw = dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1)
    enc, head = _build(kw, [1, 32, 32, 32])
    enc.train(); head.eval()
    coords, feats = blob_cloud(9, 3, [32, 32, 32], 300, 1)
    x = (torch.from_numpy(coords), torch.from_numpy(feats), 3)
    # oracle leg: same weights, dense oracle encoder -> spatial mean -> oracle head
    sd = {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}
    sd.update({k: v.detach().cpu().clone() for k, v in head.state_dict().items()})
    osd = {k: (v.requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v) for k, v in sd.items()}
    ocfg = scn_cpu.NetConfig(**kw)
    odense = scn_cpu.encoder_forward(osd, ocfg, (coords, torch.from_numpy(feats), 3), [1, 32, 32, 32])
    ologits = scn_cpu.head_forward(osd, odense, HEADS)
    sum(v.square().sum() for v in ologits.values()).backward()
    with torch.no_grad():
        p1 = enc.forward_pooled(x)
        p2 = enc.forward_pooled(x)
    # BN running statistics moved between the two calls but training-mode outputs do not depend on them
    assert torch.equal(p1, p2)
    assert rel_err(p1[:, :, 0, 0, 0], odense.mean(dim=(2, 3, 4))) < 1e-4
    enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})  # reset BN buffers
    dense_logits = head(enc(x))
    loss_d = sum(v.square().sum() for v in dense_logits.values())
    loss_d.backward()
    g_dense = enc.bottleneck.weight.grad.clone()
    enc.zero_grad(); head.zero_grad()
    fused_logits = head.forward_pooled(enc.forward_pooled(x))
    loss_f = sum(v.square().sum() for v in fused_logits.values())
    loss_f.backward()
    for k in HEADS:
        assert rel_err(fused_logits[k], dense_logits[k]) < 1e-5
        assert rel_err(fused_logits[k], ologits[k]) < 1e-4
        assert elem_close(fused_logits[k], ologits[k])[0], (k, elem_close(fused_logits[k], ologits[k])[1])
    assert rel_err(enc.bottleneck.weight.grad, g_dense) < 1e-4
    assert rel_err(enc.bottleneck.weight.grad, osd["bottleneck.weight"].grad) < 1e-4
