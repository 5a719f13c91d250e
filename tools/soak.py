#!/usr/bin/env python3
"""Soak test: many training steps on CHANGING batches (level sizes vary step to step); watches the loss, the
library's device pool and torch's allocator for growth."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import generate_batch
B, STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 24
PREFETCH = len(sys.argv) > 3 and sys.argv[3] in ("prefetch", "overlap", "overlap-prefetch")   # tools/soak.py 16 24 prefetch: rulebooks of step i+1 on a side stream
OVERLAP = len(sys.argv) > 3 and sys.argv[3].startswith("overlap")   # overlap | overlap-prefetch (no mode 2) | overlap-dw (no prefetch)   # ... built BESIDE the running step, dW joined once per backward pass;
                                                            # then the same steps serially: final parameters must be bit-identical
batches = [generate_batch(0, i, B, (900, 900, 900), 50000 - 3000 * (i % 4), "tracks", 16) for i in range(6)]
import torch
from sparseeventid_amd import _lib
from sparseeventid_amd.train import Trainer
from sparseeventid_amd import scn
torch.manual_seed(1234)
tr = Trainer(image_size=[1, 900, 900, 900])
if OVERLAP and sys.argv[3] in ("overlap", "overlap-dw"):   # overlap-none: the control (two serial runs)
    scn.set_backward_streams(2)
ready = None
lib = _lib.lib()
held, used = ctypes.c_int64(), ctypes.c_int64()
t0 = time.time()
staged_in = [tr.stage(tr.to_scn_input(arr), labels) for arr, labels in batches]
if OVERLAP:
    ready = torch.cuda.Event(); ready.record()
nxt = tr.prefetch(staged_in[0][0], after=ready) if PREFETCH else None
for step in range(STEPS):
    image, lab = staged_in[step % len(batches)]
    loss, _ = tr.train_step(nxt if PREFETCH else image, lab)
    if PREFETCH:
        nxt = tr.prefetch(staged_in[(step + 1) % len(batches)][0], after=ready)
    if step % 4 == 3 or step == STEPS - 1:
        torch.cuda.synchronize()
        lib.sei_pool_stats(ctypes.byref(held), ctypes.byref(used))
        print("step %3d loss %.4f  pool held %.2f GB in-use %.2f GB  torch alloc %.2f GB reserved %.2f GB  %.1f s"
              % (step, float(loss), held.value / 1e9, used.value / 1e9, torch.cuda.memory_allocated() / 1e9,
                 torch.cuda.memory_reserved() / 1e9, time.time() - t0), flush=True)
assert torch.isfinite(loss)
if OVERLAP:
    torch.cuda.synchronize()
    scn.set_backward_streams(0)
    torch.manual_seed(1234)
    ref = Trainer(image_size=[1, 900, 900, 900])
    for step in range(STEPS):
        image, lab = staged_in[step % len(batches)]
        ref.train_step(image, lab)
    torch.cuda.synchronize()
    same = torch.equal(ref.arena.flat, tr.arena.flat)
    print("overlapped run vs serial run after %d steps on changing batches: parameters %s" % (STEPS, "bit-identical" if same else "DIFFER"))
    assert same
print("soak ok")
