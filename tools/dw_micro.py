#!/usr/bin/env python3
"""Kernel experiments: dW time of one submanifold conv layer per level of the synthetic DUNE-3D batch, register-operand
kernel vs LDS-shared kernel (SEI_DW_KERNEL / SEI_DW_SPS are read by the library per call).
  python tools/dw_micro.py --events 16 [--levels 0 1 2 3 4 5]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
CH = {0: 32, 1: 64, 2: 96, 3: 128, 4: 160, 5: 192}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=16)
    ap.add_argument("--levels", type=int, nargs="*", default=[0, 1, 2, 3, 4, 5])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", nargs="*", default=["default"],
                    help="each variant = comma list of ENV=VALUE the library reads per call (SEI_DW_MT, SEI_DW_FIT_KB), or 'default'")
    a = ap.parse_args()
    from bench import generate_batch
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", min(16, os.cpu_count() or 8))
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd._lib import lib, check
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
    t.features = torch.randn(t.features.shape[0], 16, device="cuda")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    level = 0
    while level <= max(a.levels):
        if level in a.levels:
            C = CH[level]
            sp = [int(x) for x in t.spatial_size]
            n = t.features.shape[0]
            x = torch.randn(n, C, device="cuda"); dy = torch.randn(n, C, device="cuda")
            rb = t.metadata.subm_rulebook(sp, [3, 3, 3])
            conv = scn.SubmanifoldConvolution(3, C, C, 3, True).cuda()
            conv(scn.SparseConvNetTensor(x, t.metadata, t.spatial_size))   # builds the tile variant the forward uses
            R = rb.n_rules
            dw = torch.empty(27, C, C, device="cuda"); ref = None
            for v in a.variants:
                for kv in ("SEI_DW_MT", "SEI_DW_FIT_KB"):
                    os.environ.pop(kv, None)
                lib().sei_debug_force_dw_kernel({"reg": 1, "ring": 2}.get(v, 0))
                if v not in ("default", "reg", "ring"):
                    for kv in v.split(","):
                        k_, _, val = kv.partition("=")
                        os.environ[k_] = val
                fn = lambda: check(lib().sei_conv_backward_weight(rb.h, p(x), p(dy), p(dw), None, C, C, st))
                fn(); torch.cuda.synchronize()
                if ref is None:
                    ref = dw.clone(); err = 0.0
                else:
                    err = float((dw - ref).abs().max() / ref.abs().max())
                ev0.record()
                for _ in range(a.iters):
                    fn()
                ev1.record(); torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1) / a.iters
                print("L%d C=%d N=%d R/N=%.2f dW %-32s %.3f ms  %.1f TFLOP/s  (max diff vs first %.1e)" % (
                    level, C, n, R / n, v, ms, 2.0 * R * C * C / ms / 1e9, err), flush=True)
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
        level += 1


if __name__ == "__main__":
    main()
