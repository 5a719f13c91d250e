#!/usr/bin/env python3
"""Do two conv kernels that run TOGETHER on the GPU finish sooner than one after the other?  Two half batches (A, B) of the
synthetic DUNE-3D batch at one level; kernels forced per launch (sei_debug_force_conv_kernel); pairs launched on one
stream (serial) and on two streams (together).   python tools/corun_probe.py --events 32 --levels 2 3
"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CH = {0: 32, 1: 64, 2: 96, 3: 128, 4: 160, 5: 192}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=32, help="events per half")
    ap.add_argument("--levels", type=int, nargs="*", default=[2, 3])
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--workers", type=int, default=int(os.environ.get("SEI_GEN_WORKERS", "0")) or min(16, os.cpu_count() or 8))
    a = ap.parse_args()
    from bench import generate_batch
    arrs = [generate_batch(0, s, a.events, (900, 900, 900), 50000, "tracks", a.workers)[0] for s in (0, 1)]
    import torch
    import sparseeventid_amd.scn as scn
    from sparseeventid_amd._lib import lib, check
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    L = lib()
    ts = []
    for arr in arrs:
        c, f, b = larcvsparse_to_scnsparse_3d(arr)
        t = scn.InputLayer(3, [900, 900, 900])((torch.from_numpy(c), torch.from_numpy(f).cuda(), b))
        t.features = torch.randn(t.features.shape[0], 16, device="cuda")
        ts.append(t)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    level = 0
    while level <= max(a.levels):
        if level in a.levels:
            C = CH[level]
            work = []
            for t in ts:
                sp = [int(x) for x in t.spatial_size]
                n = t.features.shape[0]
                rb = t.metadata.subm_rulebook(sp, [3, 3, 3])
                work.append(dict(rb=rb, n=n, x=torch.randn(n, C, device="cuda"), y=torch.empty(n, C, device="cuda"),
                                 dy=torch.randn(n, C, device="cuda"), dw=torch.empty(27, C, C, device="cuda"),
                                 w=torch.randn(27, C, C, device="cuda") * 0.05))

            def fwd(kind):
                def run(wk, st):
                    L.sei_debug_force_conv_kernel(kind)
                    check(L.sei_conv_forward(wk["rb"].h, P(wk["x"]), P(wk["w"]), None, P(wk["y"]), C, C, ctypes.c_void_p(st.cuda_stream)))
                    L.sei_debug_force_conv_kernel(0)
                return run

            def dw(wk, st):
                check(L.sei_conv_backward_weight(wk["rb"].h, P(wk["x"]), P(wk["dy"]), P(wk["dw"]), None, C, C, ctypes.c_void_p(st.cuda_stream)))

            ops = {"tile": fwd(1), "wave": fwd(2), "dW": dw}

            def timed(fa, fb, together):
                best = 1e9
                for it in range(a.iters + 2):
                    torch.cuda.synchronize()
                    e0.record(s1)
                    fa(work[0], s1)
                    if together:
                        s2.wait_event(e0)
                        fb(work[1], s2)
                        s1.wait_stream(s2)
                    else:
                        fb(work[1], s1)
                    e1.record(s1)
                    torch.cuda.synchronize()
                    if it >= 2:
                        best = min(best, e0.elapsed_time(e1))
                return best

            R = sum(wk["rb"].n_rules for wk in work)
            print("level %d  C=%d  rows %d + %d  rules %d" % (level, C, work[0]["n"], work[1]["n"], R), flush=True)
            for na, nb in (("tile", "tile"), ("tile", "wave"), ("wave", "wave"), ("tile", "dW"), ("wave", "dW"), ("dW", "dW")):
                ser = timed(ops[na], ops[nb], False)
                par = timed(ops[na], ops[nb], True)
                tf = 2.0 * R * C * C / 1e12
                print("   %-4s + %-4s  serial %7.3f ms (%5.1f TF)   together %7.3f ms (%5.1f TF)   %+5.1f %%"
                      % (na, nb, ser, tf / ser * 1e3, par, tf / par * 1e3, (ser / par - 1) * 100), flush=True)
        if level == max(a.levels):
            break
        # next level: the network's own strided convolution (sites only matter)
        nts = []
        for t in ts:
            conv = scn.Convolution(3, t.features.shape[1], 16, 2, 2, False).cuda()
            with torch.no_grad():
                nts.append(conv(t))
        ts = nts
        level += 1


if __name__ == "__main__":
    main()
