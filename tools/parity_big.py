#!/usr/bin/env python3
"""One-off parity run at a larger batch than the test suite affords: the default DUNE-3D network on B full-size synthetic
events (50 000 voxels each, 900^3), engine (MI355X) vs the C++/OpenMP restatement of the SCN-CPU algorithm
(oracle/scn_omp.cpp through oracle/scn_cpu.py's "omp" backend -- gather -> sgemm -> scatter-add, a different code path from
the torch-CPU oracle the tests use).  Test infrastructure, not product code.   python tools/parity_big.py 8"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import scn_cpu
from sparseeventid_amd import build_networks
from sparseeventid_amd.config import Config
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d, synthetic_dune3d_batch
from sparseeventid_amd.loss import focal_loss

HEADS = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
IMAGE = [1, 900, 900, 900]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max() / b.detach().cpu().double().abs().max().clamp_min(1e-300))
l2 = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm().clamp_min(1e-300))

arr, labels = synthetic_dune3d_batch(0, 5, B)
coords, feats, bs = larcvsparse_to_scnsparse_3d(arr)
labels = {k: torch.from_numpy(v) for k, v in labels.items()}
ocfg = scn_cpu.NetConfig()
sd = scn_cpu.init_params(ocfg, seed=13, heads=HEADS)
enc, head = build_networks(Config(), IMAGE, HEADS)
enc.load_state_dict({k: v for k, v in sd.items() if not k.startswith("classification_head")})
head.load_state_dict({k: v for k, v in sd.items() if k.startswith("classification_head")})
enc.cuda().train(); head.cuda().eval()
t0 = time.time()
dense = enc((torch.from_numpy(coords), torch.from_numpy(feats), B))
logits = head(dense)
loss = focal_loss(logits, {k: v.cuda() for k, v in labels.items()})
loss.backward(); torch.cuda.synchronize()
t_gpu = time.time() - t0

torch.set_num_threads(16)
scn_cpu.use_backend("omp", threads=16)
osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone()) for k, v in sd.items()}
t0 = time.time()
odense = scn_cpu.encoder_forward(osd, ocfg, (coords, torch.from_numpy(feats), B), IMAGE)
ologits = scn_cpu.head_forward(osd, odense, HEADS)
oloss = scn_cpu.focal_loss(ologits, labels)
oloss.backward()
t_cpu = time.time() - t0
print("default DUNE-3D network, %d events x 50000 voxels, 900^3; engine (first call, incl. builds) %.1f s, OpenMP oracle %.1f s" % (B, t_gpu, t_cpu))
print("dense encoder output  max-norm rel err %.2e   (active entries %d vs %d)" % (rel(dense, odense), int((dense != 0).sum()), int((odense != 0).sum())))
for k in HEADS:
    print("logits %-12s      max-norm rel err %.2e" % (k, rel(logits[k], ologits[k])))
print("focal loss            %.7f vs %.7f" % (float(loss.detach()), float(oloss.detach())))
named = dict(enc.named_parameters())
named.update({k: v for k, v in head.named_parameters()})
worst = []
gmax = max(float(osd[k].grad.abs().max()) for k in named)
for k, p in named.items():
    og = osd[k].grad
    if float(og.abs().max()) < 1e-5 * gmax:
        continue   # conv bias in front of BatchNorm: analytically zero on both sides
    worst.append((l2(p.grad, og), k))
worst.sort(reverse=True)
print("parameter gradients: %d tensors, relative L2 error: median %.2e, worst %.2e (%s)" % (
    len(worst), worst[len(worst) // 2][0], worst[0][0], worst[0][1]))
ok = rel(dense, odense) < 1e-4 and all(rel(logits[k], ologits[k]) < 1e-4 for k in HEADS) and worst[0][0] < 5e-3
print("PARITY %s (1e-4 forward, 5e-3 relative L2 on gradients as in tests/test_gpu_fullnet.py)" % ("OK" if ok else "FAILED"))
sys.exit(0 if ok else 1)
