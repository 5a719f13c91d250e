#!/usr/bin/env python3
"""In-kernel s_memtime shares of the column-owner conv kernel (diagnostic build:
make -C sparseeventid_amd/csrc EXTRA="-DSEI_EXPERIMENTS -DSEI_STAMPS").
  python tools/cols_stamps.py LEVEL C [EVENTS]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c = int(sys.argv[1]), int(sys.argv[2])
nev = int(sys.argv[3]) if len(sys.argv) > 3 else 64
arr, _ = generate_batch(0, 0, nev, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
os.environ["SEI_CONV_KERNEL"] = "tile"
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
t.features = torch.randn(t.features.shape[0], c, device="cuda")
os.environ["SEI_CONV_KERNEL"] = "cols"
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
conv(t); torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
l.sei_debug_cols_stamps(buf, 1)
conv(t); torch.cuda.synchronize()
l.sei_debug_cols_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = {0: "prologue (tile init, first loads)", 1: "wait: own A piece landed", 2: "barrier", 3: "issue: B batch, index words, A piece",
         4: "issue: LDS reads of the next sub-chunk", 5: "MFMA chain + tile read-modify-write", 6: "offset switch: B copy (+ hipcc's wait)",
         7: "epilogue (tile -> Y, statistics)"}
tot = sum(v[i] for i in names)
print("level %d C=%d: all waves, s_memtime ticks per wave and tile (%d wave-tiles), share" % (lvl, c, v[10]))
for i, n in names.items():
    print("  %-44s %10.1f  %5.1f%%" % (n, v[i] / max(v[10], 1), 100 * v[i] / tot))
print("  steps / tile %.1f; switches / tile %.1f; ticks per step %.1f (MFMA + RMW part %.1f; 24 MFMAs at pipe rate = 768 cycles)" % (
    v[8] / v[10], v[9] / v[10], (tot - v[0] - v[7]) / max(v[8], 1), v[5] / max(v[8], 1)))
