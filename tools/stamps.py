import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch
lvl, c = int(sys.argv[1]), int(sys.argv[2])
arr, _ = generate_batch(0, 0, 16, (900, 900, 900), 50000, "tracks", 16)
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
conv(t); torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)   # diagnostic build: make -C sparseeventid_amd/csrc EXTRA=-DSEI_STAMPS
buf = (ctypes.c_ulonglong * 16)()
l.sei_debug_stamps(buf, 1)
conv(t); torch.cuda.synchronize()
l.sei_debug_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = {0: "wait loads + barrier", 1: "accumulators <- LDS tile", 2: "issue W slice DMA + gathers", 3: "MFMA phase",
         4: "A register rotate", 5: "accumulators -> LDS tile", 6: "next offset bookkeeping", 10: "prologue", 11: "epilogue"}
tot = sum(v[i] for i in names)
print("level %d C=%d: wave 1, s_memtime ticks per tile (%d tiles), share" % (lvl, c, v[7]))
for i, n in names.items():
    print("  %-30s %10.1f  %5.1f%%" % (n, v[i] / max(v[7], 1), 100 * v[i] / tot))
print("  phases / tile %.1f; MFMA issued / phase %.1f; ticks per MFMA in the MFMA phase %.3f" % (
    v[8] / v[7], v[9] / max(v[8], 1), v[3] / max(v[9], 1)))
