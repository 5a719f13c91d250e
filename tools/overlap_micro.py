#!/usr/bin/env python3
"""Kernel experiment: do the dX and dW kernels of one layer finish sooner when they run CONCURRENTLY (two streams) than back to
back?  Each alone keeps the MFMA pipes ~60 % busy (its phases are latency chains), so co-resident workgroups of the other
kernel could fill the gaps.   python tools/overlap_micro.py --events 64 --level 2 --c 96"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=64)
    ap.add_argument("--level", type=int, default=2)
    ap.add_argument("--c", type=int, default=96)
    ap.add_argument("--iters", type=int, default=10)
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
    for _ in range(a.level):
        t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
    C = a.c
    n = t.features.shape[0]
    x = torch.randn(n, C, device="cuda")
    dy = torch.randn(n, C, device="cuda")
    w = torch.randn(27, C, C, device="cuda") * 0.05
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    rb = t.metadata.subm_rulebook([int(v) for v in t.spatial_size], [3, 3, 3])
    L = lib()
    P = lambda tt: ctypes.c_void_p(tt.data_ptr())
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    S = lambda s: ctypes.c_void_p(s.cuda_stream)

    def f_dx(s): check(L.sei_conv_backward_input(rb.h, P(dy), P(w), P(dx), C, C, S(s)))
    def f_dw(s): check(L.sei_conv_backward_weight(rb.h, P(x), P(dy), P(dw), None, C, C, S(s)))
    def f_fwd(s): check(L.sei_conv_forward(rb.h, P(x), P(w), None, P(dx), C, C, S(s)))

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(a.iters):
            fn()
        e1.record(main_s); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    def both_seq():
        f_dx(main_s); f_dw(main_s)

    def both_par():
        ev = torch.cuda.Event(); ev.record(main_s); side.wait_event(ev)
        f_dw(side); f_dx(main_s)
        main_s.wait_stream(side)

    for rnd in range(3):
        print("L%d C=%d N=%d round %d: dX %.3f ms | dW %.3f ms | back to back %.3f ms | two streams %.3f ms" % (
            a.level, C, n, rnd, timed(lambda: f_dx(main_s)), timed(lambda: f_dw(main_s)), timed(both_seq), timed(both_par)), flush=True)


if __name__ == "__main__":
    main()
