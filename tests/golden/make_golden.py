#!/usr/bin/env python3
"""Generates the committed golden fixtures (run in the build container, where /root/reference exists).

1. transform_3d.npz / transform_2d.npz -- inputs and outputs of the REFERENCE's own
   larcvsparse_to_scnsparse_{3d,2d} (imported by file path from
   /root/reference/src/utils/core/larcvio/data_transforms.py, the torch-free twin of
   src/io/data_transforms.py:21-49,198-253): pins sparseeventid_amd.data transforms.
2. lr_schedule.npz -- values of the REFERENCE's own WarmupFlatDecay schedule (imported by file path from
   /root/reference/src/utils/learning_rate_scheduler.py:92-126) for a few parameter sets: pins train.WarmupFlatDecay.
3. encoder_tiny.npz -- seeded tiny network input and the ORACLE's outputs (dense encoder output,
   logits, loss).  The reference cannot produce these (its arithmetic is in the absent third-party
   `sparseconvnet`), so this fixture pins the engine + oracle against drift, not against the reference.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def reference_transforms():
    path = "/root/reference/src/utils/core/larcvio/data_transforms.py"
    spec = importlib.util.spec_from_file_location("ref_data_transforms", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_lr_schedules():
    path = "/root/reference/src/utils/learning_rate_scheduler.py"
    spec = importlib.util.spec_from_file_location("ref_lr_scheduler", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    # (peak, floor, epoch_length, decay_epochs, total_epochs); the first is the reference's "standard" default
    sets = [(3e-3, 1e-3, 100, 5, 25), (3e-3, 1e-3, 10, 5, 25), (1e-2, 1e-4, 7, 3, 6), (1e-3, 1e-5, 1, 1, None)]
    for i, (peak, floor, el, de, te) in enumerate(sets):
        s = mod.WarmupFlatDecay(peak_learning_rate=peak, decay_floor=floor, epoch_length=el, decay_epochs=de, total_epochs=te)
        n = el * ((te or 25) + 2)
        out["params_%d" % i] = np.array([peak, floor, el, de, -1 if te is None else te], np.float64)
        out["values_%d" % i] = np.array([float(s[j]) for j in range(n)], np.float64)
    np.savez_compressed(os.path.join(HERE, "lr_schedule.npz"), **out)


def main():
    reference_lr_schedules()
    ref = reference_transforms()
    rng = np.random.default_rng(20261003)
    # larcv 3-D batch [B,1,M,4] = (x,y,z,value), -999 padded, ragged fill levels incl. an empty event
    B, M = 4, 40
    a3 = np.full((B, 1, M, 4), -999.0, np.float32)
    for b, n in enumerate([40, 17, 0, 1]):
        a3[b, 0, :n, :3] = rng.integers(0, 900, (n, 3))
        a3[b, 0, :n, 3] = rng.standard_normal(n)
    c, f, bs = ref.larcvsparse_to_scnsparse_3d(a3)
    np.savez_compressed(os.path.join(HERE, "transform_3d.npz"), input=a3, coords=c, features=f, batch_size=bs)
    # larcv 2-D batch [B,3,M,3] = (x,y,value)
    a2 = np.full((3, 3, 30, 3), -999.0, np.float32)
    for b in range(3):
        for p in range(3):
            n = int(rng.integers(0, 30))
            a2[b, p, :n, 0] = rng.integers(0, 1024, n)
            a2[b, p, :n, 1] = rng.integers(0, 1536, n)
            a2[b, p, :n, 2] = rng.standard_normal(n)
    c2, f2, bs2 = ref.larcvsparse_to_scnsparse_2d(a2)
    np.savez_compressed(os.path.join(HERE, "transform_2d.npz"), input=a2, coords=c2, features=f2, batch_size=bs2)

    from oracle import scn_cpu
    from tests.util import blob_cloud
    heads = {"labelneutID": 3, "labelprotID": 3, "labelnpiID": 2, "labelcpiID": 2}
    kw = dict(depth=2, n_initial_filters=16, n_output_filters=16, blocks_per_layer=1)
    cfg = scn_cpu.NetConfig(**kw)
    sd = scn_cpu.init_params(cfg, seed=11, heads=heads)
    image_size = [1, 32, 32, 32]
    coords, feats = blob_cloud(5, 2, image_size[1:], 300, 1)
    labels = {k: rng.integers(0, n, 2) for k, n in heads.items()}
    dense = scn_cpu.encoder_forward(sd, cfg, (coords, torch.from_numpy(feats), 2), image_size)
    logits = scn_cpu.head_forward(sd, dense, heads)
    loss = scn_cpu.focal_loss(logits, {k: torch.from_numpy(v) for k, v in labels.items()})
    out = {"coords": coords, "features": feats, "dense": dense.numpy(), "loss": float(loss)}
    out.update({"logits_" + k: v.numpy() for k, v in logits.items()})
    out.update({"label_" + k: v for k, v in labels.items()})
    out.update({"param_" + k: v.numpy() for k, v in scn_cpu.init_params(cfg, seed=11, heads=heads).items()})
    np.savez_compressed(os.path.join(HERE, "encoder_tiny.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
