#!/usr/bin/env python3
"""In-kernel s_memtime shares of the sixteen-wave round kernel (conv_mround_mfma_kernel), every wave stamped
(diagnostic build: make -C sparseeventid_amd/csrc -B EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS").
   python tools/mround_stamps.py LEVEL C [EVENTS]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c = int(sys.argv[1]), int(sys.argv[2])
events = int(sys.argv[3]) if len(sys.argv) > 3 else 64
os.environ.setdefault("SEI_CONV_KERNEL", "rounds")
arr, _ = generate_batch(0, 0, events, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
t.features = torch.randn(t.features.shape[0], c, device="cuda")
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
for _ in range(3):
    conv(t)
torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 256)()
assert l.sei_debug_mround_stamps(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); conv(t); e1.record(); torch.cuda.synchronize()
assert l.sei_debug_mround_stamps(buf, 1) == 0
v = np.array(list(buf), dtype=np.float64).reshape(16, 16)
names = {5: "prologue", 0: "wait + barrier", 1: "issue (slices, gathers, indices, plan)", 2: "deferred write-back + clear",
         3: "MFMA block (incl. idle phases)", 4: "end of round: write-back, rotate", 6: "drain + final barrier", 7: "store (+ statistics)"}
tiles = v[0, 11]
print("level %d C=%d, %d events: kernel %.3f ms, %d tiles; s_memtime ticks per tile and wave, share of the wave's time" % (
    lvl, c, events, e0.elapsed_time(e1), tiles))
tot = sum(v[:, i] for i in names)
print("  %-42s %s" % ("", " ".join("w%-5d" % w for w in (0, 1, 2, 3, 7, 8, 12, 15))) + "   all waves")
for i, n in names.items():
    print("  %-42s %s   %8.0f %5.1f%%" % (n, " ".join("%5.1f%%" % (100 * v[w, i] / tot[w]) for w in (0, 1, 2, 3, 7, 8, 12, 15)),
                                          v[:, i].sum() / 16 / tiles, 100 * v[:, i].sum() / tot.sum()))
print("  ticks per tile (mean over waves) %.0f; rounds / tile %.1f; busy phases per wave and tile %.1f, idle %.1f" % (
    tot.sum() / 16 / tiles, v[0, 10] / tiles, v[:, 8].sum() / 16 / tiles, v[:, 9].sum() / 16 / tiles))
busy = v[:, 8].sum()
print("  MFMA block: %.0f ticks per busy phase" % (v[:, 3].sum() / max(busy, 1)))
