"""GPU parity: SBND 2-D multiplane shared-weight ResNet (BASELINE configs[4];
/root/reference/src/networks/torch/sparseresnet.py:194-334) engine vs oracle."""
import numpy as np
import pytest
import torch

from oracle import scn_cpu
from tests.util import elem_close, rel_err

pytestmark = pytest.mark.gpu
OUT = {"label_cpi": 2, "label_prot": 3}


def _planes_cloud(seed, B, spatial, n_per_plane):
    """(plane, y, x, batch) coordinates, plane-major like larcvsparse_to_scnsparse_2d."""
    rng = np.random.default_rng(seed)
    rows = []
    for p in range(spatial[0]):
        for b in range(B):
            pts = set()
            q = rng.integers(0, spatial[1:])
            while len(pts) < n_per_plane:
                q = np.clip(q + rng.integers(-1, 2, 2), 0, np.array(spatial[1:]) - 1)
                pts.add((int(q[0]), int(q[1])))
            for (y, x) in sorted(pts):
                rows.append((p, y, x, b))
    c = np.array(rows, dtype=np.float64)
    f = rng.standard_normal((len(c), 1)).astype(np.float32)
    return c, f


def test_sbnd_network_parity():
    from sparseeventid_amd.networks_sbnd import ResNet, SBNDNetworkConfig
    spatial = (3, 64, 48)
    kw = dict(n_initial_filters=16, res_blocks_per_layer=1, network_depth=3, depth_pre_merge=1)
    ocfg = scn_cpu.SBNDConfig(spatial_size=spatial, **kw)
    sd = scn_cpu.sbnd_init_params(ocfg, OUT, seed=4)
    g = torch.Generator().manual_seed(8)
    for k in sd:
        if k.endswith(("bn1.weight", "bn2.weight", "bn.weight")):
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith(("bn1.bias", "bn2.bias", "bn.bias")):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    net = ResNet({k: [v] for k, v in OUT.items()}, SBNDNetworkConfig(spatial_size=spatial, **kw)).cuda()
    net.load_state_dict(sd)
    net.train()
    B = 3
    coords, feats = _planes_cloud(3, B, spatial, 150)
    out = net((torch.from_numpy(coords), torch.from_numpy(feats), B))
    loss = sum((v ** 2).sum() for v in out.values())
    loss.backward()

    osd = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in sd.items()}
    oout = scn_cpu.sbnd_forward(osd, ocfg, (coords, torch.from_numpy(feats), B), OUT)
    oloss = sum((v ** 2).sum() for v in oout.values())
    oloss.backward()
    for k in OUT:
        assert out[k].shape == (B, OUT[k])
        assert rel_err(out[k], oout[k]) < 1e-4
        assert elem_close(out[k], oout[k])[0], (k, elem_close(out[k], oout[k])[1])   # every class score
    named = dict(net.named_parameters())
    gmax = max(float(osd[k].grad.abs().max()) for k in named)
    for k, p in named.items():
        og = osd[k].grad
        if float(og.abs().max()) < 1e-5 * gmax:
            assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            continue
        assert rel_err(p.grad, og) < 1e-4, k
    for k, b in net.named_buffers():
        assert rel_err(b, osd[k]) < 1e-4, k
