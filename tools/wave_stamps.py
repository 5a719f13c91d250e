#!/usr/bin/env python3
"""In-kernel s_memtime shares of the wave conv kernel (diagnostic build: make -C sparseeventid_amd/csrc EXTRA=-DSEI_STAMPS).
  SEI_WAVE_PLAN=96:6,64,2 python tools/wave_stamps.py LEVEL C [EVENTS]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c = int(sys.argv[1]), int(sys.argv[2])
nev = int(sys.argv[3]) if len(sys.argv) > 3 else 16
arr, _ = generate_batch(0, 0, nev, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
t.features = torch.randn(t.features.shape[0], c, device="cuda")
os.environ["SEI_CONV_KERNEL"] = "wave"
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
conv(t); torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
l.sei_debug_wave_stamps(buf, 1)
conv(t); torch.cuda.synchronize()
l.sei_debug_wave_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = {0: "prologue (tile init, first loads)", 1: "slice: issue next slice's loads", 2: "slice: MFMAs (+ waits for operands)",
         3: "tile read-modify-write", 4: "group bookkeeping + index loads", 5: "epilogue (tile -> Y)"}
tot = sum(v[i] for i in names)
print("level %d C=%d plan %s: all waves, s_memtime ticks per tile (%d tiles), share" % (lvl, c, os.environ.get("SEI_WAVE_PLAN", "default"), v[7]))
for i, n in names.items():
    print("  %-40s %10.1f  %5.1f%%" % (n, v[i] / max(v[7], 1), 100 * v[i] / tot))
print("  groups / tile %.1f; sub-chunks / group %.2f; MFMAs / tile %.0f; ticks per MFMA inside the MFMA segments %.2f (pipe rate 32)" % (
    v[9] / v[7], v[10] / max(v[9], 1), v[8] / v[7], v[2] / max(v[8], 1)))
