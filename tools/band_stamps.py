#!/usr/bin/env python3
"""In-kernel s_memtime shares of the row-band conv kernel (diagnostic build: make -C sparseeventid_amd/csrc -B
EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS").   python tools/band_stamps.py LEVEL C PLAN [EVENTS]   (PLAN = NT,NCW,D)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c, plan = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nev = int(sys.argv[4]) if len(sys.argv) > 4 else 64
arr, _ = generate_batch(0, 0, nev, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
os.environ["SEI_CONV_KERNEL"] = "tile"
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
t.features = torch.randn(t.features.shape[0], c, device="cuda")
os.environ["SEI_CONV_KERNEL"] = "band"
os.environ["SEI_BAND_PLAN"] = "%d:%s" % (c, plan)
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
conv(t); torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
l.sei_debug_band_stamps(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); conv(t); e1.record(); torch.cuda.synchronize()
l.sei_debug_band_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
nt = int(plan.split(",")[0])
names = {0: "prologue (band init, first loads)", 1: "slice: issue next slice's gathers", 2: "slice: wait for the ring slot",
         3: "slice: B reads + MFMAs (+ operand waits)", 4: "slice: release + A rotate", 5: "band read-modify-write",
         6: "offset bookkeeping + index loads", 7: "epilogue (band -> Y)"}
tot = sum(v[i] for i in names)
print("level %d C=%d plan %s (%.3f ms with stamps): compute waves, s_memtime ticks per band (%d bands), share" % (lvl, c, plan, e0.elapsed_time(e1), v[12]))
for i, n in names.items():
    print("  %-44s %10.1f  %5.1f%%" % (n, v[i] / max(v[12], 1), 100 * v[i] / tot))
mf = v[9] * 4 * nt
print("  slices / band %.1f; sub-chunks / slice %.2f; MFMAs / band %.0f; ticks per MFMA inside the MFMA segments %.2f (pipe rate 32)" % (
    v[8] / max(v[12], 1), v[9] / max(v[8], 1), mf / max(v[12], 1), v[3] / max(mf, 1)))
lt = v[10] + v[11] + v[13]
print("loader wave: ticks per slice filled (%d fills): wait for a free slot %.1f, DMA issue %.1f, landing wait + publish %.1f" % (
    v[14], v[10] / max(v[14], 1), v[11] / max(v[14], 1), v[13] / max(v[14], 1)))
