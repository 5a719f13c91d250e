#!/usr/bin/env python3
"""Does the library's device pool stop growing when DISTINCT batches repeat?  bench.py's loop (overlapped default: dW on a side
stream, the next batch's rulebooks built beside the running step) over NB batches for STEPS steps; prints the pool's held /
in-use bytes after every step.   python tools/pool_growth.py [EVENTS] [STEPS] [NB]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import generate_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 36
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batches = [generate_batch(0, i, B, (900, 900, 900), 50000, "tracks", 16) for i in range(NB)]
import torch
from sparseeventid_amd import _lib, scn
from sparseeventid_amd.train import Trainer
tr = Trainer(image_size=[1, 900, 900, 900])
scn.set_backward_streams(2)
lib = _lib.lib()
held, used = ctypes.c_int64(), ctypes.c_int64()
staged = [tr.stage(tr.to_scn_input(a), l) for a, l in batches]
ready = torch.cuda.Event(); ready.record()
nxt = tr.prefetch(staged[0][0], after=ready)
hist = []
for step in range(STEPS):
    tr.train_step(nxt, staged[step % NB][1])
    nxt = tr.prefetch(staged[(step + 1) % NB][0], after=ready)
    torch.cuda.synchronize()
    lib.sei_pool_stats(ctypes.byref(held), ctypes.byref(used))
    hist.append(held.value)
    print("step %3d  pool held %.3f GB in-use %.3f GB  torch reserved %.2f GB" % (step, held.value / 1e9, used.value / 1e9, torch.cuda.memory_reserved() / 1e9), flush=True)
last = max(i for i in range(1, STEPS) if hist[i] > hist[i - 1]) if any(hist[i] > hist[i - 1] for i in range(1, STEPS)) else 0
print("last growth at step %d of %d; held %.3f GB" % (last, STEPS, hist[-1] / 1e9))
