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
    ap.add_argument("--order", default="input", choices=["input", "morton", "block8", "shuffle"],
                    help="row order experiment: re-order each event's voxels before the InputLayer (first-appearance row order)")
    ap.add_argument("--zeros", action="store_true", help="all-zero features and weights: same instruction stream, minimal switching power (clock probe)")
    ap.add_argument("--dense", type=int, default=0, help="dense D^3 cube per event instead of the DUNE batch (kernel ceiling: no padding, perfect balance)")
    ap.add_argument("--workers", type=int, default=int(os.environ.get("SEI_GEN_WORKERS", "0")) or min(16, os.cpu_count() or 8))
    a = ap.parse_args()
    import torch
    import sparseeventid_amd.scn as scn
    if a.dense:
        D = a.dense
        ax = np.arange(D)
        g3 = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
        c = np.concatenate([np.concatenate([g3, np.full((len(g3), 1), e)], 1) for e in range(a.events)]).astype(np.int64)
        f = np.ones((len(c), 1), np.float32)
        t = scn.InputLayer(3, [D, D, D])((torch.from_numpy(c), torch.from_numpy(f).cuda(), a.events))
    else:
        from bench import generate_batch
        arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, a.kind, a.workers)
        from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
        c, f, b = larcvsparse_to_scnsparse_3d(arr)
        if a.order != "input":
            ci = c.astype(np.int64)
            if a.order == "morton":
                def spread(v):
                    v = v & 0x3FF
                    v = (v | (v << 16)) & 0x030000FF
                    v = (v | (v << 8)) & 0x0300F00F
                    v = (v | (v << 4)) & 0x030C30C3
                    v = (v | (v << 2)) & 0x09249249
                    return v
                key = (spread(ci[:, 0]) << 2) | (spread(ci[:, 1]) << 1) | spread(ci[:, 2])
            elif a.order == "block8":
                key = (((ci[:, 0] >> 3) * 128 + (ci[:, 1] >> 3)) * 128 + (ci[:, 2] >> 3)) * 512 + ((ci[:, 0] & 7) * 64 + (ci[:, 1] & 7) * 8 + (ci[:, 2] & 7))
            else:
                key = np.random.default_rng(0).permutation(len(ci))
            o = np.lexsort((key, ci[:, 3]))
            c, f = c[o], f[o]
        t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
        t.features = torch.randn(t.features.shape[0], 16, device="cuda")
        for _ in range(a.level):
            t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    sp = [int(x) for x in t.spatial_size]
    n = t.features.shape[0]
    x = torch.randn(n, a.cin, device="cuda", requires_grad=a.bwd)
    t.features = x
    conv = scn.SubmanifoldConvolution(3, a.cin, a.cout, 3, True).cuda()
    if a.zeros:
        with torch.no_grad():
            conv.weight.zero_(); conv.bias.zero_(); x.zero_()
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
        # what-if: remainder sub-chunks of an offset split by columns over the 4 waves (halves / quarters)
        for TR in (128, 64):
            cost = {"whole": 0.0, "halves": 0.0, "quarters": 0.0}
            tot_sub = 0
            hist = np.zeros(17, np.int64)
            for k in range(27):
                rows = ro[start[k]:start[k + 1]]
                cnt = np.bincount(rows // TR, minlength=(n + TR - 1) // TR)
                subs = (cnt + 15) // 16
                hist += np.bincount(subs, minlength=17)[:17]
                q, r = subs // 4, subs % 4
                tot_sub += subs.sum()
                cost["whole"] += (q + (r > 0)).sum()
                cost["halves"] += (q + np.array([0, .5, .5, 1.])[r]).sum()
                cost["quarters"] += (q + r / 4.0).sum()
            print("  TR=%d ns histogram %s" % (TR, hist[:10].tolist()))
            print("  TR=%d balance: " % TR + "  ".join("%s %.3f" % (kk, tot_sub / (4 * v)) for kk, v in cost.items()))
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
