#!/usr/bin/env python3
"""Host time of one step: how long the python thread needs to QUEUE a train step (train_step returns with the GPU still
busy) and a prefetch, against the GPU's step time -- the margin the host has before the step becomes launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import generate_batch
arr, labels = generate_batch(0, 0, 64, (900, 900, 900), 50000, "tracks", min(16, os.cpu_count() or 8))
import torch
from sparseeventid_amd import scn
from sparseeventid_amd.train import Trainer
tr = Trainer(image_size=[1, 900, 900, 900], device="cuda", seed=0)
image, lab = tr.stage(tr.to_scn_input(arr), labels)
ready = torch.cuda.Event(); ready.record()
scn.set_backward_streams(2)
staged = tr.prefetch(image, after=ready)
for _ in range(3):
    tr.train_step(staged, lab); staged = tr.prefetch(image, after=ready)
torch.cuda.synchronize()
q, p = [], []
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter(); tr.train_step(staged, lab); b = time.perf_counter()
    staged = tr.prefetch(image, after=ready); c = time.perf_counter()
    q.append((b - a) * 1e3); p.append((c - b) * 1e3)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 8 * 1e3
print("GPU step %.1f ms; host: queueing a train step %.1f ms (min %.1f), prefetch incl. its read-backs %.1f ms (min %.1f)"
      % (wall, sum(q) / 8, min(q), sum(p) / 8, min(p)))
