#!/usr/bin/env python3
"""Kernel experiments: forward time of one submanifold conv layer for workgroup-tile plans
(SEI_FWD_PLAN="maxs,cs,slj,nOut,blocks" is read by the library per call).
  python tools/sweep_tile.py --events 64 --level 2 --c 96 --plans default 2,1,2,96,2 4,1,2,96,2"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=64)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--c", type=int, default=96)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--plans", nargs="*", default=["default"])
    a = ap.parse_args()
    from bench import generate_batch
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", min(16, os.cpu_count() or 8))
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    t.features = torch.randn(t.features.shape[0], 16, device="cuda")
    os.environ["SEI_CONV_KERNEL"] = "tile"
    for _ in range(a.level):
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    C = a.c
    x = torch.randn(t.features.shape[0], C, device="cuda")
    tt = scn.SparseConvNetTensor(x, t.metadata, t.spatial_size)
    conv = scn.SubmanifoldConvolution(3, C, C, 3, True).cuda()
    rb = t.metadata.subm_rulebook([int(v) for v in t.spatial_size], [3, 3, 3])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ref = None
    for plan in a.plans:
        if plan == "default":
            os.environ.pop("SEI_FWD_PLAN", None)
        else:
            os.environ["SEI_FWD_PLAN"] = plan
        try:
            out = conv(tt); torch.cuda.synchronize()
        except RuntimeError as e:
            print("plan %-14s ERROR %s" % (plan, e), flush=True)
            continue
        if ref is None:
            ref = out.features.clone(); err = 0.0
        else:
            err = float((out.features - ref).abs().max() / ref.abs().max())
        ev0.record()
        for _ in range(a.iters):
            conv(tt)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.iters
        R = rb.n_rules
        print("L%d C=%d N=%d plan %-14s fwd %.3f ms  %.1f TFLOP/s  (max diff %.1e)" % (
            a.level, C, x.shape[0], plan, ms, 2.0 * R * C * C / ms / 1e9, err), flush=True)


if __name__ == "__main__":
    main()
