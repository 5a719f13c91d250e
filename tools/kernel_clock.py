#!/usr/bin/env python3
"""The shader clock the conv kernels HOLD while they run (MI355X_MICROARCH.md, 'DVFS give-back' item 6): diagnostic build
(make -C sparseeventid_amd/csrc -B EXTRA="-DSEI_EXPERIMENTS -DSEI_CLOCK") -- wave 0 of every workgroup adds the s_memtime
(shader cycles) and s_memrealtime (100 MHz) ticks of its lifetime to a buffer of its own.  One layer (forward + dX + dW) is
launched back to back for SECONDS seconds on random data, then the same for another second with the stamps read.
   python tools/kernel_clock.py LEVEL C [EVENTS] [SECONDS]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c = int(sys.argv[1]), int(sys.argv[2])
events = int(sys.argv[3]) if len(sys.argv) > 3 else 64
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
arr, _ = generate_batch(0, 0, events, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
n = t.features.shape[0]
x = torch.randn(n, c, device="cuda", requires_grad=True)
gy = torch.randn(n, c, device="cuda")
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
tt = scn.SparseConvNetTensor(x, t.metadata, t.spatial_size)
l = ctypes.CDLL(_lib.LIB_PATH)
fam = {0: "conv_round_mfma_kernel (forward + dX)", 1: "conv_mround_mfma_kernel (forward + dX)", 2: "conv_dw_mfma_kernel"}


def run(seconds, fwd_only):
    t0 = time.time(); k = 0
    while time.time() - t0 < seconds:
        for _ in range(10):
            if fwd_only:
                with torch.no_grad():
                    conv(tt)
            else:
                conv.weight.grad = None; x.grad = None
                conv(tt).features.backward(gy)
        torch.cuda.synchronize(); k += 10
    return k


for mode, fwd_only in (("forward only", True), ("forward + dX + dW", False)):
    run(secs, fwd_only)
    buf, bufd = (ctypes.c_ulonglong * 16)(), (ctypes.c_ulonglong * 16)()
    assert l.sei_debug_clock(buf, 1) == 0 and l.sei_debug_clock_dw(bufd, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k = run(1.0, fwd_only); e1.record(); torch.cuda.synchronize()
    assert l.sei_debug_clock(buf, 1) == 0 and l.sei_debug_clock_dw(bufd, 1) == 0
    v = np.array(list(buf), dtype=np.float64).reshape(8, 2)
    v[2] = np.array(list(bufd), dtype=np.float64).reshape(8, 2)[2]
    print("level %d C=%d N=%d, %d events, %s: %d iterations in %.0f ms" % (lvl, c, n, events, mode, k, e0.elapsed_time(e1)))
    for i, name in fam.items():
        if v[i, 1] > 0:
            print("   %-42s in-kernel clock %.3f GHz  (sum memtime %.3e / sum memrealtime %.3e x 100 MHz)" % (
                name, v[i, 0] / v[i, 1] * 0.1, v[i, 0], v[i, 1]))
