"""GPU: rulebooks built ahead of time on a side stream (Trainer.prefetch / sei_metadata_prebuild_like) are the SAME
rulebooks the forward pass builds lazily (resnet.py:141-161 -> SCN Metadata), and a training loop that prefetches
every batch ends bit-identical to one that does not (every kernel of the engine is deterministic)."""
import numpy as np
import pytest
import torch

from tests.util import blob_cloud

pytestmark = pytest.mark.gpu
HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}


def _trainer(seed=3):
    from sparseeventid_amd.config import Config, Encoder
    from sparseeventid_amd.train import Trainer
    cfg = Config(encoder=Encoder(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1))
    return Trainer(cfg=cfg, image_size=[1, 32, 32, 32], output_shape=HEADS, device="cuda", seed=seed)


def _batches(tr):
    out = []
    for seed, n in ((11, 350), (12, 500), (13, 280)):
        coords, feats = blob_cloud(seed, 3, [32, 32, 32], n, 1)
        g = torch.Generator().manual_seed(seed)
        labels = {k: torch.randint(0, v, (3,), generator=g).numpy() for k, v in HEADS.items()}
        out.append(tr.stage((coords, feats, 3), labels))
    return out


def test_prebuilt_rulebooks_equal_lazy_rulebooks():
    import sparseeventid_amd.scn as scn
    sp = [24, 20, 28]
    ca, fa = blob_cloud(5, 2, sp, 400, 1)
    cb, fb = blob_cloud(6, 2, sp, 520, 1)
    layer = scn.InputLayer(3, sp)

    def walk(t):  # the rulebooks a small network would ask for, built lazily
        out = []
        rb = t.metadata.subm_rulebook(sp, [3, 3, 3]); rb.stats(128); out.append(rb)
        rb = t.metadata.subm_rulebook(sp, [5, 5, 5]); rb.stats(256); out.append(rb)
        conv = scn.Convolution(3, 16, 16, 2, 2, False).cuda()
        t.features = torch.randn(t.features.shape[0], 16, device="cuda", requires_grad=True)
        d = conv(t)
        d.features.sum().backward()   # builds the transposed variant too
        sp2 = [int(v) for v in d.spatial_size]
        rb2 = d.metadata.subm_rulebook(sp2, [3, 3, 3]); rb2.stats(64); out.append(rb2)
        return out, sp2

    ta = layer((torch.from_numpy(ca), torch.from_numpy(fa).cuda(), 2))
    walk(ta)
    # batch B: lazily ...
    tb_lazy = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
    lazy, sp2 = walk(tb_lazy)
    # ... and prebuilt from A's metadata on a side stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tb_pre = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
        tb_pre.metadata.prebuild_like(ta.metadata)
    torch.cuda.current_stream().wait_stream(side)
    pre = [tb_pre.metadata.subm_rulebook(sp, [3, 3, 3]), tb_pre.metadata.subm_rulebook(sp, [5, 5, 5]),
           tb_pre.metadata.subm_rulebook(sp2, [3, 3, 3])]
    import ctypes
    from sparseeventid_amd._lib import lib
    for a, b, tr_rows in zip(lazy, pre, (128, 256, 64)):
        info = (ctypes.c_int64 * 8)()
        assert lib().sei_rulebook_info(b.h, info) == 0 and info[7] == tr_rows   # built already, with the template's tile size
        assert a.n_rules == b.n_rules and a.n_rules > 0
        sa, ia, oa = a.export("cuda")
        sb, ib, ob = b.export("cuda")
        assert list(sa) == list(sb) and torch.equal(ia, ib) and torch.equal(oa, ob)


@pytest.mark.parametrize("overlapped", [False, True])
def test_training_with_prefetch_is_bit_identical_to_lazy_training(overlapped):
    """overlapped: the side stream waits only for the event behind which the inputs were staged, so batch j+1's hash and
    rulebooks are built WHILE step j computes (pool blocks released by step j-1 are fenced by the library)."""
    lazy, pre = _trainer(), _trainer()
    assert torch.equal(lazy.arena.flat, pre.arena.flat)
    batches = _batches(lazy)
    ready = None
    if overlapped:
        ready = torch.cuda.Event()
        ready.record()
    order = [0, 1, 2, 0, 1]
    losses_lazy = []
    torch.manual_seed(123)   # the heads hold nn.Dropout: both runs must draw the same masks
    for i in order:
        loss, _ = lazy.train_step(*batches[i])
        losses_lazy.append(float(loss))
    from sparseeventid_amd.train import Prefetched
    losses_pre = []
    torch.manual_seed(123)
    staged = pre.prefetch(batches[order[0]][0], after=ready)
    assert isinstance(staged, Prefetched)
    for j, i in enumerate(order):
        loss, _ = pre.train_step(staged, batches[i][1])
        if j + 1 < len(order):
            staged = pre.prefetch(batches[order[j + 1]][0], after=ready)   # while step j is still running
        losses_pre.append(float(loss))
    torch.cuda.synchronize()
    assert losses_lazy == losses_pre
    assert torch.equal(lazy.arena.flat, pre.arena.flat)
    for (k, a), (_, b) in zip(lazy.encoder.named_buffers(), pre.encoder.named_buffers()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("mode", [1, 2])
def test_two_stream_backward_is_bit_identical(mode):
    """scn.set_backward_streams: dW of every layer on the library's side stream next to dX (sei_conv_backward), joined per
    layer (1) or once per backward pass (2: Trainer.train_step joins; operands held by the python layer until the join, not record_stream'ed) -- same kernels,
    same results; also together with the overlapped prefetch."""
    import sparseeventid_amd.scn as scn
    base, over = _trainer(), _trainer()
    batches = _batches(base)
    order = [0, 1, 2, 1, 0, 2]
    torch.manual_seed(77)
    for i in order:
        base.train_step(*batches[i])
    ready = torch.cuda.Event()
    ready.record()
    scn.set_backward_streams(mode)
    try:
        torch.manual_seed(77)
        staged = over.prefetch(batches[order[0]][0], after=ready)
        for j, i in enumerate(order):
            over.train_step(staged, batches[i][1])
            if j + 1 < len(order):
                staged = over.prefetch(batches[order[j + 1]][0], after=ready)
        torch.cuda.synchronize()
    finally:
        scn.set_backward_streams(0)
    assert torch.equal(base.arena.flat, over.arena.flat)



def test_overlapped_step_is_bit_identical_at_full_size():
    """The default network on full-size events (kernels of milliseconds: the dW side stream really lags behind the dX
    chain, the rulebook stream really runs beside the step): six steps on changing batches, overlapped (dW joined once
    per backward pass + rulebooks one batch ahead) against serial -- final parameters bit-identical.  (The 32^3 cases
    above finish each kernel before the next is queued and cannot see an ordering hole; tools/soak.py found one.)"""
    import sparseeventid_amd.scn as scn
    from bench import generate_batch
    from sparseeventid_amd.train import Trainer
    data = [generate_batch(0, i, 4, (900, 900, 900), 50000 - 5000 * i, "tracks", 1) for i in range(3)]   # no fork: the GPU is initialised
    runs = {}
    for name in ("serial", "overlapped"):
        torch.manual_seed(99)
        tr = Trainer(image_size=[1, 900, 900, 900], device="cuda", seed=5)
        staged = [tr.stage(tr.to_scn_input(arr), labels) for arr, labels in data]
        order = [0, 1, 2, 1, 0, 2]
        if name == "serial":
            for i in order:
                tr.train_step(*staged[i])
        else:
            ready = torch.cuda.Event()
            ready.record()
            scn.set_backward_streams(2)
            try:
                nxt = tr.prefetch(staged[order[0]][0], after=ready)
                for j, i in enumerate(order):
                    tr.train_step(nxt, staged[i][1])
                    if j + 1 < len(order):
                        nxt = tr.prefetch(staged[order[j + 1]][0], after=ready)
                torch.cuda.synchronize()
            finally:
                scn.set_backward_streams(0)
        torch.cuda.synchronize()
        runs[name] = tr.arena.flat.clone()
        del tr, staged
    assert torch.equal(runs["serial"], runs["overlapped"])


def test_deferred_join_detects_a_gradient_autograd_copied():
    """Mode 2 relies on autograd TAKING OVER the dW tensor as param.grad (no kernel).  If something else holds the tensor
    (here: a tensor hook that keeps it), AccumulateGrad clones it on the main stream before the side stream has written
    it -- the join must refuse that instead of training on garbage."""
    import sparseeventid_amd.scn as scn
    tr = _trainer()
    batches = _batches(tr)
    kept = []
    w = next(p for n, p in tr.encoder.named_parameters() if n.endswith("convolution_2.conv1.weight"))
    h = w.register_hook(lambda g: kept.append(g))
    scn.set_backward_streams(2)
    try:
        with pytest.raises(RuntimeError, match="copied by autograd"):
            tr.train_step(*batches[0])
    finally:
        h.remove()
        torch.cuda.synchronize()
        scn.set_backward_streams(0)


def _sync_count(reset=False):
    import ctypes
    from sparseeventid_amd._lib import lib
    n = ctypes.c_int64()
    assert lib().sei_debug_sync_count(ctypes.byref(n), 1 if reset else 0) == 0
    return n.value


def test_batched_rulebook_build_needs_one_readback_and_changes_nothing():
    """From the second step on the Trainer builds every rulebook variant of the batch -- forward, transposed and dW's --
    at the top of the forward pass from the previous batch's list (sei_metadata_prebuild_like on the main stream): ONE
    count read-back for all of them, plus ONE for the InputLayer and all strided levels together (round 4: they are
    discovered as a chain whose row counts stay on the device), and none in the backward pass; the lazy path
    synchronises once per level and per variant.  Training is bit-identical either way."""
    lazy, batched = _trainer(), _trainer()
    lazy.batch_rulebooks = False
    data = _batches(lazy)
    order = [0, 1, 2, 1, 0]
    counts = {"lazy": [], "batched": []}
    for name, tr in (("lazy", lazy), ("batched", batched)):
        torch.manual_seed(31)
        for i in order:
            torch.cuda.synchronize()
            _sync_count(reset=True)
            tr.train_step(*data[i])
            torch.cuda.synchronize()
            counts[name].append(_sync_count())
    assert torch.equal(lazy.arena.flat, batched.arena.flat)
    # ONE for the InputLayer and all strided levels (discovered as a chain with device-side row counts: scn.InputLayer.template
    # -> sei_input_build_like), ONE for all tiled variants
    assert counts["batched"][0] == counts["lazy"][0]              # first step: nothing to copy the list from
    assert all(c == 2 for c in counts["batched"][1:]), counts
    assert all(c > 3 for c in counts["lazy"]), counts


def test_forward_only_prebuild_skips_the_backward_variants():
    """sei_metadata_prebuild_forward_like builds what a forward pass of the template walked and nothing else: a backward
    pass behind it still finds its variants missing (it builds them lazily: stream synchronisations), one behind the
    full prebuild finds everything built (none).  Results are the same either way."""
    import sparseeventid_amd.scn as scn
    sp = [24, 20, 28]
    ca, fa = blob_cloud(5, 2, sp, 400, 1)
    cb, fb = blob_cloud(6, 2, sp, 520, 1)
    layer = scn.InputLayer(3, sp)
    torch.manual_seed(5)
    sub = scn.SubmanifoldConvolution(3, 16, 32, 3, False).cuda()
    down = scn.Convolution(3, 32, 48, 2, 2, False).cuda()

    def run(t, x):
        t.features = x.clone().requires_grad_(True)
        y = down(sub(t))
        y.features.square().sum().backward()
        return y.features.detach().clone(), t.features.grad.clone(), sub.weight.grad.clone(), down.weight.grad.clone()

    ta = layer((torch.from_numpy(ca), torch.from_numpy(fa).cuda(), 2))
    run(ta, torch.randn(ta.features.shape[0], 16, device="cuda"))       # the template: forward + backward variants
    res, syncs = {}, {}
    xb = None
    for name, fwd_only in (("full", False), ("forward", True)):
        tb = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
        if xb is None:
            xb = torch.randn(tb.features.shape[0], 16, device="cuda")
        tb.metadata.prebuild_like(ta.metadata, forward_only=fwd_only)
        sub.weight.grad = None; down.weight.grad = None
        torch.cuda.synchronize()
        _sync_count(reset=True)
        with torch.no_grad():
            down(sub(scn.SparseConvNetTensor(xb, tb.metadata, tb.spatial_size)))
        torch.cuda.synchronize()
        assert _sync_count() == 0, name          # the forward pass finds its variants built in both cases
        res[name] = run(tb, xb)
        torch.cuda.synchronize()
        syncs[name] = _sync_count()
    assert syncs["full"] == 0 and syncs["forward"] > 0, syncs
    for a, b in zip(res["full"], res["forward"]):
        assert torch.equal(a, b)


def test_eval_step_between_training_steps_keeps_the_training_template():
    """Trainer.eval_step (no_grad) prebuilds the forward variants only and does not replace the template of the training
    steps: the step after it still needs ONE read-back for its rulebooks; logits equal the lazy path's."""
    lazy, batched = _trainer(), _trainer()
    lazy.batch_rulebooks = False
    data = _batches(lazy)
    out = {}
    for name, tr in (("lazy", lazy), ("batched", batched)):
        torch.manual_seed(31)
        tr.train_step(*data[0])
        tr.train_step(*data[1])
        torch.cuda.synchronize()
        _sync_count(reset=True)
        logits = tr.eval_step(data[2][0])
        torch.cuda.synchronize()
        n_eval = _sync_count(reset=True)
        tr.train_step(*data[0])
        torch.cuda.synchronize()
        out[name] = (logits, n_eval, _sync_count())
    for k in out["lazy"][0]:
        assert torch.equal(out["lazy"][0][k], out["batched"][0][k]), k
    assert out["batched"][1] == 2 and out["batched"][2] == 2, out["batched"][1:]
    assert torch.equal(lazy.arena.flat, batched.arena.flat)


def test_deferred_join_from_another_stream_waits_for_the_side_stream():
    """Mode 2 (deferred join): the backward pass runs under stream A, the caller then switches to stream B and joins THERE
    (an optimizer on its own stream).  sei_backward_join joins every side stream of the device with un-joined work into the
    joining stream, so the weight gradients read under B are the finished ones: equal to the serial run's, bit for bit.
    (It used to look up only the side stream paired with the joining stream, find none, and return without waiting.)"""
    import sparseeventid_amd.scn as scn
    sp, B, C = [40, 36, 38], 2, 96
    coords, feats = blob_cloud(91, B, sp, 6000, 1)
    t = scn.InputLayer(3, sp)((torch.from_numpy(coords), torch.from_numpy(feats).cuda(), B))
    n = t.features.shape[0]
    torch.manual_seed(4)
    convs = [scn.SubmanifoldConvolution(3, C, C, 3, False).cuda() for _ in range(6)]
    x = torch.randn(n, C, device="cuda")
    gy = torch.randn(n, C, device="cuda")

    def run():
        for c in convs:
            c.weight.grad = None
        h = scn.SparseConvNetTensor(x.clone().requires_grad_(True), t.metadata, t.spatial_size)
        for c in convs:
            h = c(h)
        h.features.backward(gy)

    run()
    torch.cuda.synchronize()
    want = [c.weight.grad.clone() for c in convs]
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    scn.set_backward_streams(2)
    try:
        a.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(a):
            run()
        b.wait_stream(a)                       # orders B behind the dX chain, NOT behind the library's dW side stream
        with torch.cuda.stream(b):
            scn.backward_join([c.weight for c in convs])
            got = [c.weight.grad.clone() for c in convs]   # read under B right behind the join
        torch.cuda.synchronize()
    finally:
        scn.set_backward_streams(0)
    for w, g in zip(want, got):
        assert torch.equal(w, g)


def test_lazy_build_waits_for_a_batch_build_in_flight():
    """Two host threads, two streams: one runs sei_metadata_prebuild_like (begins every variant, releases the rulebooks'
    mutexes around its one read-back, finishes them), the other asks for the same variants lazily at the same time (the
    autograd thread does when a prefetch overlaps a backward pass).  A begun variant is marked and the lazy call waits for
    it instead of starting a second build into the same slot: same rule counts as an undisturbed build, and the device pool
    holds no more afterwards than after one (a second build used to leak its arrays)."""
    import ctypes
    import threading
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd._lib import lib
    sp = [28, 26, 24]
    ca, fa = blob_cloud(15, 2, sp, 900, 1)
    layer = scn.InputLayer(3, sp)
    ta = layer((torch.from_numpy(ca), torch.from_numpy(fa).cuda(), 2))
    variants = (64, 128, 256)
    for f in ([3, 3, 3], [5, 5, 5]):
        for v in variants:
            ta.metadata.subm_rulebook(sp, f).stats(v)
    torch.cuda.synchronize()

    def pool_in_use():
        held, used = ctypes.c_int64(), ctypes.c_int64()
        assert lib().sei_pool_stats(ctypes.byref(held), ctypes.byref(used)) == 0
        return used.value

    want = None
    for it in range(6):
        cb, fb = blob_cloud(40 + it, 2, sp, 700 + 50 * it, 1)
        tb = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
        rbs = [tb.metadata.subm_rulebook(sp, f) for f in ([3, 3, 3], [5, 5, 5])]
        errs = []
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        go = threading.Barrier(2)

        def batch():
            try:
                with torch.cuda.stream(s1):
                    go.wait()
                    tb.metadata.prebuild_like(ta.metadata)
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        def lazy():
            try:
                with torch.cuda.stream(s2):
                    go.wait()
                    for rb in rbs:
                        for v in variants:
                            check_rc = lib().sei_rulebook_build(rb.h, v, ctypes.c_void_p(s2.cuda_stream))
                            assert check_rc == 0
            except Exception as e:   # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=batch), threading.Thread(target=lazy)]
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join(60)
            assert not t_.is_alive(), "a build is stuck"
        torch.cuda.synchronize()
        assert not errs, errs
        got = [rb.stats(64)[:3] for rb in rbs]       # (rules, sub-chunks, tiles) of the 64-row variant
        # the same batch built undisturbed
        tc = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
        ref = [tc.metadata.subm_rulebook(sp, f).stats(64)[:3] for f in ([3, 3, 3], [5, 5, 5])]
        assert got == ref, (it, got, ref)
        del tb, tc, rbs
        torch.cuda.synchronize()
        used = pool_in_use()
        if want is None:
            want = used
        assert used <= want + (1 << 20), (it, used, want)   # nothing accumulates from one round to the next


def test_input_layer_with_a_template_discovers_the_levels_in_one_readback():
    """scn.InputLayer.template (sei_input_build_like): the input level and the chain of strided levels the template holds
    are discovered back to back with their row counts on the device -- ONE stream synchronisation -- and are the SAME levels
    the lazy path finds one read-back at a time: rows, coordinates and the rulebooks cut from them; duplicate input points
    (summed in input order) and an out-of-range coordinate behave as without a template."""
    import sparseeventid_amd.scn as scn
    sp = [40, 36, 44]

    def net(t):   # three levels deep; returns row counts, level coordinates and rule counts
        out = []
        f = torch.ones(t.features.shape[0], 16, device="cuda")
        t = scn.SparseConvNetTensor(f, t.metadata, t.spatial_size)
        for _ in range(3):
            s3 = [int(v) for v in t.spatial_size]
            rb = t.metadata.subm_rulebook(s3, [3, 3, 3])
            out.append((t.features.shape[0], rb.stats(128)[:3], t.metadata.coords(s3, "cuda").cpu().clone()))
            conv = scn.Convolution(3, 16, 16, 2, 2, False).cuda()
            with torch.no_grad():
                t = conv(t)
        out.append((t.features.shape[0],))
        return out

    layer = scn.InputLayer(3, sp)
    ca, fa = blob_cloud(3, 2, sp, 1500, 2)
    ta = layer((torch.from_numpy(ca), torch.from_numpy(fa).cuda(), 2))
    net(ta)                                                      # the template: three strided levels
    rng = np.random.default_rng(5)
    cb, fb = blob_cloud(4, 3, sp, 1100, 2)
    dup = rng.integers(0, len(cb), 400)                         # duplicate points: modes 3 sums them in input order
    cb = np.concatenate([cb, cb[dup]]); fb = np.concatenate([fb, rng.standard_normal((400, 2)).astype(np.float32)])
    plain = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 3))
    want = net(plain)
    torch.cuda.synchronize()
    tl = scn.InputLayer(3, sp)
    tl.template = ta.metadata
    _sync_count(reset=True)
    fast = tl((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 3))
    torch.cuda.synchronize()
    assert _sync_count() == 1
    assert torch.equal(fast.features, plain.features)
    _sync_count(reset=True)
    got = net(fast)
    for a, b in zip(want, got):
        assert a[0] == b[0]
        if len(a) > 1:
            assert a[1] == b[1] and torch.equal(a[2], b[2])
    # an out-of-range coordinate is reported, not swallowed by the level chain behind it
    bad = cb.copy(); bad[7, 1] = sp[1] + 3
    with pytest.raises(RuntimeError, match="outside spatial_size"):
        tl((torch.from_numpy(bad), torch.from_numpy(fb).cuda(), 3))
    # and an empty batch goes through
    e = tl((torch.zeros(0, 4), torch.zeros(0, 2).cuda(), 2))
    assert e.features.shape[0] == 0


def test_overlapping_strided_filters_do_not_compound_their_bounds_along_a_chain():
    """scn.Convolution(3, .., 3, 2) -- an OVERLAPPING filter: an input site may open up to 8 output sites -- chained four
    levels deep through the template paths (sei_input_build_like, sei_metadata_prebuild_like).  A level begun behind an
    unfinished one is sized by that one's UPPER bound; with 8 candidates per row the bound would compound 8x per level
    (round-4 advisor: multi-GB tables, then 'too many candidate output sites').  The chain is broken there instead: one
    synchronisation, exact sizes -- same levels, coordinates and rules as the lazy path, and a pool that stays small."""
    import ctypes
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd._lib import lib
    sp = [95, 87, 91]   # (s - 3) // 2 + 1: 47/43/45 -> 23/21/22 -> 11/10/10 -> 5/4/4

    def net(t):
        out = []
        f = torch.ones(t.features.shape[0], 16, device="cuda")
        t = scn.SparseConvNetTensor(f, t.metadata, t.spatial_size)
        for _ in range(4):
            s3 = [int(v) for v in t.spatial_size]
            conv = scn.Convolution(3, 16, 16, 3, 2, False).cuda()
            with torch.no_grad():
                conv.weight.fill_(1.0)
                t = conv(t)
            s3o = [int(v) for v in t.spatial_size]
            out.append((t.features.shape[0], t.metadata.coords(s3o, "cuda").cpu().clone(), t.features.sum(1).cpu().clone()))
        return out

    def held():
        h, u = ctypes.c_int64(), ctypes.c_int64()
        lib().sei_pool_stats(ctypes.byref(h), ctypes.byref(u))
        return h.value

    layer = scn.InputLayer(3, sp)
    ca, fa = blob_cloud(13, 2, sp, 20000, 1)
    ta = layer((torch.from_numpy(ca), torch.from_numpy(fa).cuda(), 2))
    want_a = net(ta)                                             # the template (lazy path: exact sizes)
    cb, fb = blob_cloud(14, 2, sp, 18000, 1)
    plain = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
    want = net(plain)
    torch.cuda.synchronize()
    base = held()
    tl = scn.InputLayer(3, sp)
    tl.template = ta.metadata
    fast = tl((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))      # sei_input_build_like: input level + chain
    fast.metadata.prebuild_like(ta.metadata)                               # sei_metadata_prebuild_like: nothing left to discover
    torch.cuda.synchronize()
    # 18 000 points x 8 x 8 x 8 x 8 candidates would be 74 M x ~30 B = 2 GB of tables; exact sizing needs a few MB
    assert held() - base < 256 << 20, (held() - base) / 1e6
    got = net(fast)
    for a, b in zip(want, got):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # and through prebuild_like alone (input level built the plain way, levels discovered by the prebuild)
    plain2 = layer((torch.from_numpy(cb), torch.from_numpy(fb).cuda(), 2))
    plain2.metadata.prebuild_like(ta.metadata)
    got2 = net(plain2)
    for a, b in zip(want, got2):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert want_a[0][0] != want[0][0]                                      # the two batches differ
