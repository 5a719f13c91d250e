"""GPU: the engine against the committed golden fixture (tests/golden/encoder_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

from tests.util import elem_close, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def test_encoder_matches_golden_fixture():
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.loss import focal_loss
    g = np.load(os.path.join(GOLD, "encoder_tiny.npz"))
    kw = dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1)
    enc, head = build_networks(Config(encoder=Encoder(**kw)), [1, 32, 32, 32], HEADS)
    sd = {k[len("param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param_")}
    enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
    head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
    enc.cuda().train(); head.cuda().eval()
    dense = enc((torch.from_numpy(g["coords"]), torch.from_numpy(g["features"]), 2))
    logits = head(dense)
    assert rel_err(dense, g["dense"]) < 1e-4                     # north_star tolerance, fp32
    for k in HEADS:
        assert rel_err(logits[k], g["logits_" + k]) < 1e-4
        assert elem_close(logits[k], g["logits_" + k])[0], k
    loss = focal_loss(logits, {k: torch.from_numpy(g["label_" + k]).cuda() for k in HEADS})
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
