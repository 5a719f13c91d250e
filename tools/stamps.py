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
l = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
l.sei_debug_stamps(buf, 1)
conv(t); torch.cuda.synchronize()
l.sei_debug_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = ["other", "barrier wait", "issue stage+gathers", "MFMA phase", "wait A + copy", "scatter", "tail", "tiles"]
tot = v[:7].sum()
print("level %d C=%d: per-tile wave-1 cycles (s_memtime ticks), share" % (lvl, c))
for n, x in zip(names, v):
    print("  %-22s %12.0f  %5.1f%%" % (n, x / max(v[7], 1), 100 * x / tot if n != "tiles" else 0))
