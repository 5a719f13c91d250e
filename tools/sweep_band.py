#!/usr/bin/env python3
"""Kernel experiments (library built with EXTRA=-DSEI_EXPERIMENTS): forward (or forward + backward) time of one 3^3
submanifold conv layer per level of the synthetic DUNE-3D batch for the workgroup-tile kernel, the wave-private kernel
and row-band plans SEI_BAND_PLAN=nOut:NT,NCW,D, with a bit-exactness cross-check against the first plan.
  python tools/sweep_band.py --events 64 [--levels 2 3 4 5] [--bwd] [--plans tile 3,4,8 ...]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

CH = {0: 32, 1: 64, 2: 96, 3: 128, 4: 160, 5: 192}
PLANS = {   # NT,NCW,D[,S]: S = 1 shared-tile ring kernel, 0 / absent private 64-row bands
    32: ["tile", "wave", "2,8,8,1", "2,4,8,1", "2,8,16,1", "2,8,16", "2,4,16"],
    64: ["tile", "wave", "4,8,8,1", "4,8,6,1", "4,4,8,1", "2,8,8,1", "4,8,6", "4,4,8", "4,4,16"],
    96: ["tile", "wave", "3,8,8,1", "3,8,6,1", "3,8,12,1", "3,8,16,1", "3,4,8,1", "6,8,6,1", "6,8,8,1", "2,8,8,1", "3,8,8"],
    128: ["tile", "wave", "4,8,8,1", "4,8,6,1", "4,8,12,1", "4,4,8,1", "2,8,8,1", "4,8,6"],
    160: ["tile", "wave", "5,8,8,1", "5,8,6,1", "5,4,8,1", "2,8,8,1", "5,4,8"],
    192: ["tile", "wave", "4,8,8,1", "4,8,6,1", "6,8,6,1", "3,8,8,1", "6,8,8,1", "4,8,6"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=64)
    ap.add_argument("--levels", type=int, nargs="*", default=[2, 3, 4, 5])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2, help="plans are interleaved over this many rounds (clock drift)")
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--plans", nargs="*", default=None)
    ap.add_argument("--workers", type=int, default=int(os.environ.get("SEI_GEN_WORKERS", "0")) or min(16, os.cpu_count() or 8))
    a = ap.parse_args()
    from bench import generate_batch
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", a.workers)
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    t.features = torch.randn(t.features.shape[0], 16, device="cuda")
    level = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def select(C, plan):
        os.environ.pop("SEI_BAND_PLAN", None)
        if plan in ("tile", "wave"):
            os.environ["SEI_CONV_KERNEL"] = plan
        else:
            os.environ["SEI_CONV_KERNEL"] = "band"
            os.environ["SEI_BAND_PLAN"] = "%d:%s" % (C, plan)

    while level <= max(a.levels):
        if level in a.levels:
            C = CH[level]
            sp = [int(x) for x in t.spatial_size]
            n = t.features.shape[0]
            x = torch.randn(n, C, device="cuda", requires_grad=a.bwd)
            tt = scn.SparseConvNetTensor(x, t.metadata, t.spatial_size)
            conv = scn.SubmanifoldConvolution(3, C, C, 3, True).cuda()
            rb = t.metadata.subm_rulebook(sp, [3, 3, 3])
            R = rb.n_rules
            ref = None
            plans = [p for p in PLANS[C] if a.plans is None or p in a.plans]
            best = {}
            errs = {}
            for rnd in range(a.rounds):
                for plan in plans:
                    select(C, plan)
                    try:
                        out = conv(tt)
                        torch.cuda.synchronize()
                    except RuntimeError as e:
                        errs[plan] = "ERROR %s" % e
                        continue
                    if ref is None:
                        ref = out.features.detach().clone()
                    errs[plan] = "bit-identical" if torch.equal(out.features.detach(), ref) else "DIFFERS %.1e" % float(
                        (out.features.detach() - ref).abs().max() / ref.abs().max())
                    g = torch.randn_like(out.features)
                    fn = (lambda: conv(tt).features.backward(g)) if a.bwd else (lambda: conv(tt))
                    fn(); torch.cuda.synchronize()
                    ev0.record()
                    for _ in range(a.iters):
                        fn()
                    ev1.record(); torch.cuda.synchronize()
                    ms = ev0.elapsed_time(ev1) / a.iters
                    best[plan] = min(best.get(plan, 1e9), ms)
            for plan in plans:
                if plan in best:
                    fl = 2.0 * R * C * C * (3 if a.bwd else 1)
                    print("L%d C=%d N=%d R/N=%.2f plan %-8s %s %.3f ms  %.1f TFLOP/s  %s" % (
                        level, C, n, R / n, plan, "fwd+bwd" if a.bwd else "fwd", best[plan], fl / best[plan] / 1e9, errs[plan]), flush=True)
                else:
                    print("L%d C=%d plan %-8s %s" % (level, C, plan, errs.get(plan)), flush=True)
            del x, tt, conv
        os.environ["SEI_CONV_KERNEL"] = "tile"
        os.environ.pop("SEI_BAND_PLAN", None)
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
        level += 1


if __name__ == "__main__":
    main()
