#!/usr/bin/env python3
"""Micro-benchmark of ONE sparse conv layer at a chosen level of the synthetic DUNE-3D batch
(tools for kernel work; product code is under sparseeventid_amd/).
  python tools/conv_micro.py --level 2 --cin 96 --cout 96 --events 16 [--bwd] [--stats]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--cin", type=int, default=96)
    ap.add_argument("--cout", type=int, default=96)
    ap.add_argument("--events", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--kind", default="tracks")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("SEI_GEN_WORKERS", "0")) or min(16, os.cpu_count() or 8))
    a = ap.parse_args()
    from bench import generate_batch
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, a.kind, a.workers)
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    t.features = torch.randn(t.features.shape[0], 16, device="cuda")
    for _ in range(a.level):
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    sp = [int(x) for x in t.spatial_size]
    n = t.features.shape[0]
    x = torch.randn(n, a.cin, device="cuda", requires_grad=a.bwd)
    t.features = x
    conv = scn.SubmanifoldConvolution(3, a.cin, a.cout, 3, True).cuda()
    rb = t.metadata.subm_rulebook(sp, [3, 3, 3])
    out = conv(t)
    torch.cuda.synchronize()
    R = rb.n_rules
    print("level %d spatial %s N=%d R=%d (R/N=%.2f)" % (a.level, sp, n, R, R / n), flush=True)
    if a.stats:
        start, ri, ro = rb.export("cuda")
        ro = ro.cpu().numpy().astype(np.int64)
        for TR, U in ((256, 4), (128, 2), (128, 4), (64, 4)):
            tot_sub = tot_max = 0
            for k in range(27):
                rows = ro[start[k]:start[k + 1]]
                cnt = np.bincount(rows // TR, minlength=(n + TR - 1) // TR)
                subs = (cnt + 15) // 16
                tot_sub += subs.sum()
                tot_max += ((subs + U - 1) // U).sum()
            print("  TR=%d U=%d: sub-chunks %d pad-eff %.3f balance %.3f -> combined %.3f" % (
                TR, U, tot_sub, R / (16 * tot_sub), tot_sub / (U * tot_max), R / (16 * U * tot_max)))
    g = torch.randn_like(out.features)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, fn in (("fwd", lambda: conv(t)),) + ((("fwd+bwd", lambda: conv(t).features.backward(g)),) if a.bwd else ()):
        fn(); torch.cuda.synchronize()
        ev0.record()
        for _ in range(a.iters):
            fn()
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.iters
        fl = 2.0 * R * a.cin * a.cout * (3 if name != "fwd" else 1)
        print("%s: %.3f ms  %.1f TFLOP/s (algorithmic)" % (name, ms, fl / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
