#!/usr/bin/env python3
"""Kernel experiments: forward time of one submanifold conv layer per level of the synthetic DUNE-3D batch, for the
workgroup-tile kernel and for wave-kernel plans (SEI_CONV_KERNEL / SEI_WAVE_PLAN are read by the library per call).
  python tools/sweep_wave.py --events 16 [--levels 0 1 2 3 4 5] [--bwd]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

CH = {0: 32, 1: 64, 2: 96, 3: 128, 4: 160, 5: 192}
PLANS = {
    32: ["tile", "2,64,2", "2,128,2", "2,128,3", "2,128,4", "2,192,2", "2,192,3", "2,256,4", "1,128,4"],
    64: ["tile", "4,64,2", "4,64,3", "4,64,4", "4,128,4", "2,128,4", "2,128,3"],
    96: ["tile", "6,64,2", "6,64,3", "3,128,4", "3,64,2", "6,128,2"],
    128: ["tile", "8,64,2", "4,64,2", "4,128,4", "4,128,2", "8,64,3"],
    160: ["tile", "5,64,2", "5,128,2", "5,64,3"],
    192: ["tile", "6,64,2", "6,64,3", "3,128,4", "4,64,2", "4,128,4"],
}
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=16)
    ap.add_argument("--levels", type=int, nargs="*", default=[0, 1, 2, 3, 4, 5])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--only", nargs="*", default=None, help="restrict to these plan names (e.g. tile)")
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
    while level <= max(a.levels):
        if level in a.levels:
            C = CH[level]
            sp = [int(x) for x in t.spatial_size]
            n = t.features.shape[0]
            x = torch.randn(n, C, device="cuda", requires_grad=a.bwd)
            tt = scn.SparseConvNetTensor(x, t.metadata, t.spatial_size)
            conv = scn.SubmanifoldConvolution(3, C, C, 3, True).cuda()
            rb = t.metadata.subm_rulebook(sp, [3, 3, 3])
            ref = None
            for plan in PLANS[C]:
                if a.only is not None and plan not in a.only:
                    continue
                if plan == "tile":
                    os.environ["SEI_CONV_KERNEL"] = "tile"
                elif plan == "default":
                    os.environ.pop("SEI_CONV_KERNEL", None)
                else:
                    os.environ["SEI_CONV_KERNEL"] = "wave"
                    os.environ["SEI_WAVE_PLAN"] = "%d:%s" % (C, plan)
                try:
                    out = conv(tt)
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print("L%d C=%d plan %-10s ERROR %s" % (level, C, plan, e), flush=True)
                    continue
                R = rb.n_rules
                if ref is None:
                    ref = out.features.detach().clone()
                    err = 0.0
                else:
                    err = float((out.features.detach() - ref).abs().max() / ref.abs().max())
                g = torch.randn_like(out.features)
                fn = (lambda: conv(tt).features.backward(g)) if a.bwd else (lambda: conv(tt))
                fn(); torch.cuda.synchronize()
                ev0.record()
                for _ in range(a.iters):
                    fn()
                ev1.record(); torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / a.iters
                fl = 2.0 * R * C * C * (3 if a.bwd else 1)
                print("L%d C=%d N=%d R/N=%.2f plan %-10s %s %.3f ms  %.1f TFLOP/s  (max diff vs tile %.1e)" % (
                    level, C, n, R / n, plan, "fwd+bwd" if a.bwd else "fwd", ms, fl / ms / 1e9, err), flush=True)
            del x, tt, conv
        os.environ["SEI_CONV_KERNEL"] = "tile"
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
        level += 1


if __name__ == "__main__":
    main()
