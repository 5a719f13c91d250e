"""CPU: host logic -- transforms against the reference-generated golden vectors, the oracle against its
own committed fixture, synthetic generator, LR schedule, network construction (state_dict contract)."""
import math
import os

import numpy as np
import torch

from oracle import scn_cpu
from sparseeventid_amd import data
from tests.util import rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def test_transform_3d_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "transform_3d.npz"))
    c, f, b = data.larcvsparse_to_scnsparse_3d(g["input"])
    assert c.dtype == g["coords"].dtype == np.float64 and f.dtype == g["features"].dtype == np.float32
    assert np.array_equal(c, g["coords"]) and np.array_equal(f, g["features"]) and b == int(g["batch_size"])
    assert c.shape == (58, 4) and set(c[:, 3].astype(int)) == {0, 1, 3}   # event 2 is empty


def test_transform_2d_matches_reference_golden():
    g = np.load(os.path.join(GOLD, "transform_2d.npz"))
    c, f, b = data.larcvsparse_to_scnsparse_2d(g["input"])
    assert np.array_equal(c, g["coords"]) and np.array_equal(f, g["features"]) and b == int(g["batch_size"])
    assert c.dtype == np.float64 and np.all(np.diff(c[:, 0]) >= 0)          # plane-major


def test_oracle_reproduces_its_golden_fixture():
    g = np.load(os.path.join(GOLD, "encoder_tiny.npz"))
    cfg = scn_cpu.NetConfig(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1)
    sd = {k[len("param_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param_")}
    dense = scn_cpu.encoder_forward(sd, cfg, (g["coords"], torch.from_numpy(g["features"]), 2), [1, 32, 32, 32])
    logits = scn_cpu.head_forward(sd, dense, HEADS)
    assert rel_err(dense, g["dense"]) < 1e-5
    for k in HEADS:
        assert rel_err(logits[k], g["logits_" + k]) < 1e-5
    loss = scn_cpu.focal_loss(logits, {k: torch.from_numpy(g["label_" + k]) for k in HEADS})
    assert abs(float(loss) - float(g["loss"])) < 1e-5


def test_synthetic_event_contract():
    ev = data.synthetic_event(123, (900, 900, 900), 50000)
    assert ev.shape == (50000, 4) and ev.dtype == np.float32
    xyz = ev[:, :3].astype(np.int64)
    assert xyz.min() >= 0 and xyz.max() < 900
    assert len(np.unique(xyz, axis=0)) == 50000                      # exactly 50k distinct voxels
    assert np.array_equal(ev, data.synthetic_event(123, (900, 900, 900), 50000))   # seeded
    u = data.uniform_event(5, (64, 64, 64), 1000)
    assert len(np.unique(u[:, :3], axis=0)) == 1000
    arr, labels = data.synthetic_dune3d_batch(0, 0, batch_size=2, n_voxels=2000, grid=(128, 128, 128))
    assert arr.shape == (2, 1, 2000, 4) and set(labels) == set(HEADS)
    c, f, b = data.larcvsparse_to_scnsparse_3d(arr)
    assert c.shape == (4000, 4) and b == 2


def test_lr_schedule_matches_reference_fixture():
    """tests/golden/lr_schedule.npz holds values produced by the REFERENCE's WarmupFlatDecay
    (/root/reference/src/utils/learning_rate_scheduler.py:92-126, imported by tests/golden/make_golden.py)."""
    from sparseeventid_amd.train import WarmupFlatDecay
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedule.npz"))
    n_sets = sum(1 for k in g.files if k.startswith("params_"))
    assert n_sets >= 4
    for i in range(n_sets):
        peak, floor, el, de, te = g["params_%d" % i]
        s = WarmupFlatDecay(peak_learning_rate=float(peak), decay_floor=float(floor), epoch_length=int(el),
                            decay_epochs=int(de), total_epochs=None if te < 0 else int(te))
        want = g["values_%d" % i]
        got = np.array([s[j] for j in range(len(want))])
        assert np.allclose(got, want, rtol=1e-12, atol=1e-15), i


def test_lr_schedule_matches_reference_formula():
    from sparseeventid_amd.train import WarmupFlatDecay
    s = WarmupFlatDecay(peak_learning_rate=3e-3, decay_floor=1e-3, epoch_length=10, decay_epochs=5, total_epochs=25)
    assert abs(s[0] - 1e-5) < 1e-12 and abs(s[5] - (1e-5 + 5 * (3e-3 - 1e-5) / 10)) < 1e-12
    assert s[10] == 3e-3 and s[199] == 3e-3                        # flat: (25-5-1)*10 steps
    assert abs(s[200 + 7] - ((3e-3 - 1e-3) * math.exp(-0.07) + 1e-3)) < 1e-12
    assert s[250] == 0.0


def test_network_state_dict_contract():
    from sparseeventid_amd import build_networks
    from sparseeventid_amd.config import IMAGE_SIZE, OUTPUT_SHAPE, Config
    enc, head = build_networks(Config(), IMAGE_SIZE["dune3d"], OUTPUT_SHAPE)
    sd = enc.state_dict()
    shapes = scn_cpu.param_shapes(scn_cpu.NetConfig())
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    n = sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in head.parameters())
    assert n == 20881994
    assert enc.output_shape == [128, 32, 16, 40]
    assert "classification_head.labelneutID.2.weight" in head.state_dict()
    enc2, _ = build_networks(Config(), IMAGE_SIZE["dune3d_900"], OUTPUT_SHAPE)
    assert enc2.output_shape == [128, 28, 28, 28]


def test_sbnd_state_dict_contract():
    from sparseeventid_amd.networks_sbnd import ResNet, SBNDNetworkConfig
    net = ResNet({"label": [3]}, SBNDNetworkConfig())
    shapes = scn_cpu.sbnd_param_shapes(scn_cpu.SBNDConfig(), {"label": 3})
    sd = net.state_dict()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert "pre_convolutional_layers.0.block_0.conv1.weight" in sd and "final_layer.label.1.weight" in sd


def test_checkpoint_roundtrip_and_encoder_only_restore(tmp_path):
    """Lightning-layout checkpoints (create_trainer.py:83-115) on CPU-constructed modules."""
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.train import Trainer, format_log_message
    cfg = Config(encoder=Encoder(depth=1, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1))
    a = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=1)
    b = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=2)
    assert not torch.equal(a.arena.flat, b.arena.flat)
    a.global_step = 17
    path = str(tmp_path / "checkpoints" / "step17.ckpt")
    a.save_checkpoint(path)
    ck = torch.load(path)
    assert "encoder.initial_convolution.weight" in ck["state_dict"] and ck["global_step"] == 17
    assert any(k.startswith("head.classification_head.labelneutID") for k in ck["state_dict"])
    b.load_checkpoint(path)
    assert torch.equal(a.arena.flat, b.arena.flat) and b.global_step == 17
    # parameters stay views of the flat arena after loading
    assert b.encoder.initial_convolution.weight.data_ptr() == b.arena.flat.data_ptr()
    c = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=3)
    head_before = c.head.state_dict()["classification_head.labelneutID.2.weight"].clone()
    c.restore_encoder_only(path)
    assert torch.equal(c.encoder.initial_convolution.weight, a.encoder.initial_convolution.weight)
    assert torch.equal(c.head.state_dict()["classification_head.labelneutID.2.weight"], head_before)
    assert all(not p.requires_grad for p in c.encoder.parameters())
    # frozen means FROZEN: an optimizer step (weight decay 1e-6, eps 1e-6) leaves every encoder tensor bit-identical
    # and still moves the head (advisor finding, round 1: Adam on the whole arena shrank the "frozen" encoder)
    enc_before = {k: v.clone() for k, v in c.encoder.state_dict().items()}
    c.arena.grad.zero_()
    for p in c.head.parameters():
        p.grad.normal_()
    for g in c.opt.param_groups:
        g["lr"] = 3e-3
    c.opt.step()
    assert all(torch.equal(v, enc_before[k]) for k, v in c.encoder.state_dict().items())
    assert not torch.equal(c.head.state_dict()["classification_head.labelneutID.2.weight"], head_before)
    assert "Img/s" in format_log_message({"loss/loss": 1.25}, 64, 3, dt=0.5, mode="train")
    # a shape-mismatched tensor is an error, not a silent broadcast; a Lightning-style file with an EMPTY optimizer state warns
    import pytest
    bad = dict(ck)
    bad["state_dict"] = dict(ck["state_dict"])
    bad["state_dict"]["encoder.initial_convolution.bias"] = torch.zeros(1)
    torch.save(bad, str(tmp_path / "bad.ckpt"))
    with pytest.raises(ValueError):
        b.load_checkpoint(str(tmp_path / "bad.ckpt"))
    # optimizer state of another arena layout (here: the frozen-encoder trainer optimises arena[start:]) is refused with a
    # clear error instead of corrupting fused Adam at the first step (advisor finding, round 2)
    for p in a.head.parameters():
        p.grad.normal_()
    a.opt.step()                                   # gives Adam exp_avg / exp_avg_sq of the whole arena
    a.save_checkpoint(str(tmp_path / "stepped.ckpt"))
    assert torch.load(str(tmp_path / "stepped.ckpt"))["opt_start"] == 0
    with pytest.raises(ValueError, match="optimizer state"):
        c.load_checkpoint(str(tmp_path / "stepped.ckpt"))
    c.load_checkpoint(str(tmp_path / "stepped.ckpt"), resume=False)      # weights only: fine
    b.load_checkpoint(str(tmp_path / "stepped.ckpt"))                    # same layout: fine
    lit = {"state_dict": ck["state_dict"], "global_step": 5, "optimizer_states": [{}], "lr_schedulers": []}
    torch.save(lit, str(tmp_path / "lit.ckpt"))
    with pytest.warns(UserWarning):
        b.load_checkpoint(str(tmp_path / "lit.ckpt"))
    assert b.global_step == 5


def test_flat_arena_gathered_gradients_equal_accumulated_gradients():
    """FlatArena.detach_grads / gather_grads (autograd assigns the gradients, one multi-tensor copy files them into the
    flat arena) gives the arena the autograd accumulation into a zeroed arena gives -- also for a parameter that
    receives no gradient and for one that is used twice."""
    from sparseeventid_amd.distributed import FlatArena
    torch.manual_seed(4)
    lin, unused = torch.nn.Linear(5, 4), torch.nn.Linear(3, 2)
    net = torch.nn.ModuleList([lin, unused])
    arena = FlatArena([net])
    x = torch.randn(7, 5)

    def loss():
        return (lin(x) ** 2).sum() + lin(x[:3]).sum()      # lin used twice, `unused` not at all

    arena.zero_grad()
    loss().backward()                                       # accumulation into the zeroed arena views
    want = arena.grad.clone()
    assert float(want.abs().sum()) > 0
    arena.grad.fill_(7.0)                                   # stale content
    arena.zero_grad()
    arena.detach_grads()
    assert all(p.grad is None for p in net.parameters())
    loss().backward()
    arena.gather_grads()
    assert torch.equal(arena.grad, want)
    for p in net.parameters():                              # .grad is a view of the arena again
        assert p.grad is not None and p.grad.data_ptr() >= arena.grad.data_ptr()
        assert p.grad.data_ptr() < arena.grad.data_ptr() + arena.grad.numel() * 4
    assert float(unused.weight.grad.abs().sum()) == 0.0


def test_lightning_layout_optimizer_state_matches_per_parameter_adam(tmp_path):
    """`save_checkpoint(lightning_layout=True)` writes `optimizer_states[0]` in the layout of
    `torch.optim.Adam(model.parameters()).state_dict()` (what Lightning stores for the reference's optimizer,
    supervised_eventID.py:198-207 / create_trainer.py:110-115): a plain per-parameter Adam that loads it takes the same
    next step as the Trainer's flat fused-arena Adam, and a Trainer that loads a per-parameter state (a reference-run
    checkpoint) continues exactly like the run that wrote it."""
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.train import Trainer
    cfg = Config(encoder=Encoder(depth=1, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1))
    a = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=1)
    g = torch.Generator().manual_seed(3)

    def fake_grad(tr):       # gradients of the parameters only: the alignment padding of the arena stays zero
        out = torch.zeros_like(tr.arena.grad)
        for off, cnt in tr.arena.spans:
            out[off:off + cnt] = torch.randn(cnt, generator=g) * 1e-2
        return out

    def fake_step(tr, lr):
        tr.arena.grad.copy_(fake_grad(tr))
        for gr in tr.opt.param_groups:
            gr["lr"] = lr
        tr.opt.step()
        tr.global_step += 1

    for lr in (1e-3, 2e-3, 3e-3):
        fake_step(a, lr)
    path = str(tmp_path / "lightning.ckpt")
    a.save_checkpoint(path, lightning_layout=True)
    ck = torch.load(path)
    assert ck["epoch"] == 0 and ck["global_step"] == 3 and ck["lr_schedulers"][0]["last_epoch"] == 3
    per = ck["optimizer_states"][0]
    n = len(a.arena.params)
    assert sorted(per["state"]) == list(range(n)) and per["param_groups"][0]["params"] == list(range(n))
    assert all(tuple(per["state"][i]["exp_avg"].shape) == tuple(p.shape) for i, p in enumerate(a.arena.params))

    # (1) a reference-style per-parameter Adam over copies of the parameters, loaded from the file
    ref_params = [torch.nn.Parameter(p.detach().clone()) for p in a.arena.params]
    ref = torch.optim.Adam(ref_params, lr=1.0, eps=1e-6, betas=(0.8, 0.9), weight_decay=a.weight_decay)
    ref.load_state_dict(per)
    # (2) another Trainer resumed from the file through the per-parameter path only
    del ck["optimizer"]
    torch.save(ck, path)
    b = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=2)
    b.load_checkpoint(path)
    assert b.global_step == 3 and torch.equal(a.arena.flat, b.arena.flat)
    sa, sb = a.opt.state_dict()["state"][0], b.opt.state_dict()["state"][0]
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    assert float(sa["step"]) == float(sb["step"]) == 3.0

    # the same fourth step on all three
    grad = fake_grad(a)
    for tr in (a, b):
        tr.arena.grad.copy_(grad)
        for gr in tr.opt.param_groups:
            gr["lr"] = 2.5e-3
        tr.opt.step()
    for p, (off, cnt) in zip(ref_params, a.arena.spans):
        p.grad = grad[off:off + cnt].view(p.shape).clone()
    for gr in ref.param_groups:
        gr["lr"] = 2.5e-3
    ref.step()
    assert torch.equal(a.arena.flat, b.arena.flat)
    for p, q in zip(a.arena.params, ref_params):
        assert torch.allclose(p, q, rtol=0, atol=1e-7), float((p - q).abs().max())

    # state of a frozen parameter is refused
    import pytest
    c = Trainer(cfg=cfg, image_size=[1, 32, 32, 32], device="cpu", seed=4)
    c.restore_encoder_only(path)
    with pytest.raises(ValueError, match="frozen"):
        c.load_optimizer_state_per_parameter(per)


def test_three_way_bf16_split_is_exact():
    """The arithmetic claim behind the opt-in product mode (conv_fwd.hip: split_pair, conv.hip: weights_split_kernel),
    restated in numpy: rounding an fp32 number to the nearest bf16 three times in a row, each time on the exact fp32 residual,
    writes it as an EXACT sum of three bf16 numbers (for |x| >= 2^-100; below, the residuals are subnormal) -- so the nine part
    products reproduce a*b exactly, and the three that the kernel drops are bounded by 2^-23 |a b|."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(400000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 400000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.3e38, -3.3e38, 1e-30, 65504.0, 1 + 2 ** -23, 1 - 2 ** -24], np.float32)])

    def rne16(v):   # round-to-nearest-even bf16, kept as fp32 (what v_cvt_pk_bf16_f32 produces)
        u = v.view(np.uint32).astype(np.uint64)
        return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)

    x1 = rne16(x)
    r1 = x - x1                       # exact in fp32
    x2 = rne16(r1)
    r2 = r1 - x2
    x3 = rne16(r2)
    normal = (np.abs(x) >= 2.0 ** -100) | (x == 0)
    assert np.array_equal(x3[normal], r2[normal])                            # the third part is the residual itself: nothing is left
    total = x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64)
    assert np.array_equal(total[normal], x.astype(np.float64)[normal])       # exact
    assert np.all(np.abs(x2[normal]) <= 2.0 ** -8 * np.abs(x[normal])) and np.all(np.abs(x3[normal]) <= 2.0 ** -16 * np.abs(x[normal]))
    # the dropped part products: a2 b3 + a3 b2 + a3 b3
    a, b = x[:200000], x[200000:400000]
    a1 = rne16(a); a2 = rne16(a - a1); a3 = (a - a1) - a2
    b1 = rne16(b); b2 = rne16(b - b1); b3 = (b - b1) - b2
    dropped = np.abs(a2.astype(np.float64) * b3 + a3.astype(np.float64) * b2 + a3.astype(np.float64) * b3)
    exact = np.abs(a.astype(np.float64) * b.astype(np.float64))
    ok = (np.abs(a) >= 2.0 ** -100) & (np.abs(b) >= 2.0 ** -100)
    assert np.all(dropped[ok] <= 2.0 ** -23 * exact[ok])
