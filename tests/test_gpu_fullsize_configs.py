"""GPU: BASELINE configs[4] (SBND 2-D multiplane net, 128 events x 3 planes x 15 000 pixels in 3 x 2048 x 1280;
/root/reference/src/networks/torch/sparseresnet.py:194-334) and configs[1] (default DUNE-3D net, forward only, 32
events; /root/reference/bin/exec.py:207-224) at FULL size through size-independent properties -- the oracle would need
minutes there: rulebook mirror symmetry of the [1,3,3] / [3,3,3] filters, plane separation, mass conservation of the
[1,2,2] / stride [1,2,2] down-sampling, and bit-identical reruns of the whole networks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SBND = [3, 2048, 1280]
B_SBND, NPIX = 128, 15000


@pytest.fixture(scope="module")
def sbnd_batch():
    from bench import generate_batch
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_2d
    arr, _ = generate_batch(0, 0, B_SBND, tuple(SBND), NPIX, "sbnd", 1)   # no fork: the GPU may be initialised
    return larcvsparse_to_scnsparse_2d(arr)


@pytest.fixture(scope="module")
def sbnd_level0(sbnd_batch):
    import sparseeventid_amd.scn as scn
    c, f, b = sbnd_batch
    return scn.InputLayer(3, SBND)((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))


def _pairs(ri, ro, n):
    return (ri.long() * n + ro.long()).sort().values


def test_sbnd_input_and_rulebook_symmetry(sbnd_batch, sbnd_level0):
    c, f, b = sbnd_batch
    n = B_SBND * 3 * NPIX
    assert sbnd_level0.features.shape == (n, 1)                 # distinct pixels per (event, plane)
    locs = sbnd_level0.get_spatial_locations().numpy()
    assert np.array_equal(locs, c.astype(np.int64))             # rows keep the reference's plane-major input order
    md = sbnd_level0.metadata
    for filt, K in (([1, 3, 3], 9), ([3, 3, 3], 27), ([1, 5, 5], 25)):
        rb = md.subm_rulebook(SBND, filt)
        start, ri, ro = rb.export("cuda")
        cnt = np.diff(np.asarray(start))
        assert len(cnt) == K and cnt[K // 2] == n                # centre offset: every site sees itself
        assert np.array_equal(cnt, cnt[::-1])                    # (i -> o, k) <=> (o -> i, K-1-k)
        assert rb.n_rules == int(cnt.sum())
        k = 1
        a = _pairs(ri[start[k]:start[k + 1]], ro[start[k]:start[k + 1]], n)
        m = _pairs(ro[start[K - 1 - k]:start[K - k]], ri[start[K - 1 - k]:start[K - k]], n)
        assert torch.equal(a, m)
        plane = torch.from_numpy(locs[:, 0]).cuda()
        same = plane[ri.long()] == plane[ro.long()]
        if filt[0] == 1:
            assert bool(same.all())                              # [1,k,k] filters never mix wire planes
        else:
            assert not bool(same.all())                          # [3,3,3] does (post-merge layers)


def test_sbnd_downsample_conserves_mass_per_plane(sbnd_level0):
    import sparseeventid_amd.scn as scn
    n = sbnd_level0.features.shape[0]
    x = torch.rand(n, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    conv = scn.Convolution(3, 16, 16, [1, 2, 2], [1, 2, 2], False).cuda()
    with torch.no_grad():
        conv.weight.zero_()
        for k in range(4):
            conv.weight[k] = torch.eye(16)
    out = conv(scn.SparseConvNetTensor(x, sbnd_level0.metadata, sbnd_level0.spatial_size))
    assert [int(v) for v in out.spatial_size] == [3, 1024, 640]
    assert float((out.features.sum(0) - x.sum(0)).abs().max() / x.sum(0).abs().max()) < 1e-4
    locs = sbnd_level0.get_spatial_locations()
    parents = torch.unique(torch.stack([locs[:, 0], locs[:, 1] // 2, locs[:, 2] // 2, locs[:, 3]], 1), dim=0)
    assert out.features.shape[0] == parents.shape[0]
    # per plane too: the planes stay apart through a [1,2,2] convolution
    op = out.get_spatial_locations()[:, 0].cuda()
    ip = locs[:, 0].cuda()
    for p in range(3):
        a, b = out.features[op == p].sum(), x[ip == p].sum()
        assert float((a - b).abs() / b.abs()) < 1e-4


def test_sbnd_network_full_size_step_is_finite_and_deterministic(sbnd_batch):
    """The whole default SBND network (depth 6, pre-merge 3, 32 filters, 2 res-blocks per level), forward + backward on
    128 events: finite, every parameter receives a gradient, and a second run is bit-identical."""
    from sparseeventid_amd.networks_sbnd import ResNet, SBNDNetworkConfig
    c, f, b = sbnd_batch
    out_shape = {"label_cpi": [2], "label_prot": [3]}
    torch.manual_seed(0)
    net = ResNet(out_shape, SBNDNetworkConfig()).cuda()
    net.train()
    coords = torch.from_numpy(c).cuda().to(torch.int32)
    feats = torch.from_numpy(f).cuda()

    def run():
        for p in net.parameters():
            p.grad = None
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        out = net((coords, feats, b))
        loss = sum((v ** 2).sum() for v in out.values())
        loss.backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        net.load_state_dict(sd)   # BatchNorm running statistics back to where they were
        return {k: v.detach().clone() for k, v in out.items()}, grads

    o1, g1 = run()
    o2, g2 = run()
    for k, v in o1.items():
        assert v.shape == (B_SBND, out_shape[k][0]) and bool(torch.isfinite(v).all())
        assert torch.equal(v, o2[k]), k
    for k, g in g1.items():
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, k
        assert torch.equal(g, g2[k]), k


@pytest.fixture(scope="module")
def dune_batch64():
    """bench.py's headline batch (rank 0, batch 0: seeds 0..63); its first 32 events are the 32-event batch of configs[1]"""
    from bench import generate_batch
    return generate_batch(0, 0, 64, (900, 900, 900), 50000, "tracks", 1)   # no fork: the GPU may be initialised


def test_dune3d_default_net_training_step_64_events(dune_batch64):
    """configs[2] AS A WHOLE (/root/reference/src/networks/resnet.py:141-161 under src/utils/create_trainer.py:50-58's
    training loop): the default DUNE-3D network, 64 full-size events from bench.generate_batch, two `train_step`s (the second
    one builds its rulebooks from the first one's template) -- finite loss and logits, every parameter receives a finite
    gradient, a rerun from the same state is bit-identical, and the overlapped default of bench.py (dW on a side stream
    joined once per backward pass, the next step's rulebooks built beside the running one) equals the serial step bit for
    bit.  The oracle needs minutes at this size: it checks this network on 2 events (test_gpu_fullnet.py)."""
    import gc
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.train import Trainer
    arr, labels = dune_batch64
    runs = {}
    for name in ("serial", "serial again", "overlapped"):
        tr = Trainer(image_size=[1, 900, 900, 900], device="cuda", seed=0)
        image, lab = tr.stage(tr.to_scn_input(arr), labels)
        ready = torch.cuda.Event()
        ready.record()
        over = name == "overlapped"
        scn.set_backward_streams(2 if over else 0)
        try:
            loss1, logits1 = tr.train_step(tr.prefetch(image, after=ready) if over else image, lab)
            grad1 = tr.arena.grad.clone()
            nxt = tr.prefetch(image, after=ready) if over else image
            loss2, _ = tr.train_step(nxt, lab)
            torch.cuda.synchronize()
        finally:
            scn.set_backward_streams(0)
        assert bool(torch.isfinite(loss1)) and bool(torch.isfinite(loss2)) and float(loss1) > 0
        for k, v in logits1.items():
            assert v.shape[0] == 64 and bool(torch.isfinite(v).all()), k
        names = [n for m in (tr.encoder, tr.head) for n, _ in m.named_parameters()]
        assert len(names) == len(tr.arena.spans)
        for pname, (off, n) in zip(names, tr.arena.spans):
            gp = grad1[off:off + n]
            assert bool(torch.isfinite(gp).all()), pname
            # (a bias in front of a BatchNorm has a mathematically zero gradient: rounding noise, possibly exact zeros)
            if not pname.endswith("conv1.bias"):
                assert float(gp.abs().max()) > 0, pname
        runs[name] = (loss1.clone(), loss2.clone(), grad1, tr.arena.flat.clone(), {k: v.detach().clone() for k, v in logits1.items()})
        # memory ceiling of one 64-event rank (round-4 review): the library's pool held 36.7 GB and torch reserved 44 GB in
        # round 4; with the host's run-ahead bounded the pool stays below 16 GB.  288 GB per GPU is not a licence to grow.
        import ctypes
        from sparseeventid_amd._lib import lib
        held, used = ctypes.c_int64(), ctypes.c_int64()
        lib().sei_pool_stats(ctypes.byref(held), ctypes.byref(used))
        assert held.value < 40e9, held.value / 1e9
        assert torch.cuda.max_memory_reserved() < 80e9, torch.cuda.max_memory_reserved() / 1e9
        del tr, image, lab, nxt, logits1
        gc.collect()
        torch.cuda.empty_cache()
    ref = runs["serial"]
    for name in ("serial again", "overlapped"):
        got = runs[name]
        assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), name      # losses of both steps
        assert torch.equal(ref[2], got[2]), name                                      # every gradient of step 1
        assert torch.equal(ref[3], got[3]), name                                      # parameters after two Adam steps
        for k in ref[4]:
            assert torch.equal(ref[4][k], got[4][k]), (name, k)


def test_dune3d_default_net_forward_only_32_events(dune_batch64):
    """configs[1]: the default DUNE-3D network, eval-free forward (train-mode BatchNorm as bin/exec.py's iotest/forward
    path runs it under no_grad) on 32 full-size events: dense output shape, finite logits, tanh range, inactive cells
    exact zeros, bit-identical rerun, and the fused sparse head equals the dense head."""
    from sparseeventid_amd.train import Trainer
    grid = (900, 900, 900)
    arr, labels = dune_batch64[0][:32], {k: v[:32] for k, v in dune_batch64[1].items()}
    tr = Trainer(image_size=[1] + list(grid), device="cuda", seed=0)
    image, _ = tr.stage(tr.to_scn_input(arr), labels)
    with torch.no_grad():
        sd = {k: v.clone() for k, v in tr.encoder.state_dict().items()}
        dense = tr.encoder(image)
        assert dense.shape == (32, 128, 28, 28, 28)
        assert bool(torch.isfinite(dense).all()) and float(dense.abs().max()) <= 1.0
        # level 5 holds far fewer sites than cells: most of the dense tensor is exact zeros
        assert int((dense != 0).sum()) < dense.numel() // 4
        tr.head.eval()
        logits = tr.head(dense)
        tr.encoder.load_state_dict(sd)
        dense2 = tr.encoder(image)
        assert torch.equal(dense, dense2)
        for k, v in logits.items():
            assert v.shape[0] == 32 and bool(torch.isfinite(v).all())
        # fused sparse head against the dense path on the same batch
        tr.encoder.load_state_dict(sd)
        pooled = tr.encoder.forward_pooled(tr.encoder.input_layer(image))
        flog = tr.head.forward_pooled(pooled)
        for k in logits:
            assert float((flog[k] - logits[k]).abs().max()) <= 1e-4 * float(logits[k].abs().max()), k
