"""Shared helpers for tests: small seeded sparse inputs."""
import numpy as np
import torch


def random_cloud(seed, batch, spatial, n_per_event, channels=1, dup_frac=0.0, dtype=np.float64):
    """Random unique voxels per event (+ optional duplicates), batch-major like the
    reference transform (src/io/data_transforms.py:21-49): coords [N,4]=(x0,x1,x2,b)."""
    rng = np.random.default_rng(seed)
    cs, fs = [], []
    for b in range(batch):
        n = n_per_event if np.isscalar(n_per_event) else n_per_event[b]
        if n == 0:
            continue
        vol = spatial[0] * spatial[1] * spatial[2]
        flat = rng.choice(vol, size=min(n, vol), replace=False)
        x0, rem = np.divmod(flat, spatial[1] * spatial[2])
        x1, x2 = np.divmod(rem, spatial[2])
        c = np.stack([x0, x1, x2, np.full_like(x0, b)], -1)
        nd = int(dup_frac * len(c))
        if nd:
            c = np.concatenate([c, c[rng.integers(0, len(c), nd)]])
        cs.append(c)
        fs.append(rng.standard_normal((len(c), channels)).astype(np.float32))
    coords = np.concatenate(cs).astype(dtype)
    feats = np.concatenate(fs)
    return coords, feats


def blob_cloud(seed, batch, spatial, n_per_event, channels=1):
    """Clustered voxels (random walks) so that neighbourhoods are populated."""
    rng = np.random.default_rng(seed)
    cs, fs = [], []
    sp = np.asarray(spatial)
    for b in range(batch):
        pts = set()
        p = rng.integers(0, sp)
        while len(pts) < n_per_event:
            p = np.clip(p + rng.integers(-1, 2, 3), 0, sp - 1)
            pts.add(tuple(int(v) for v in p))
            if rng.random() < 0.02:
                p = rng.integers(0, sp)
        c = np.array(sorted(pts), dtype=np.int64)
        c = np.concatenate([c, np.full((len(c), 1), b)], 1)
        cs.append(c)
        fs.append(rng.standard_normal((len(c), channels)).astype(np.float32))
    return np.concatenate(cs).astype(np.float64), np.concatenate(fs)


def sort_rows(coords, feats):
    """Canonical row order (b, x0, x1, x2) for comparing row sets."""
    c = np.asarray(coords).astype(np.int64)
    key = ((c[:, 3] * 4096 + c[:, 0]) * 4096 + c[:, 1]) * 4096 + c[:, 2]
    o = np.argsort(key, kind="stable")
    f = feats.detach().cpu().numpy() if isinstance(feats, torch.Tensor) else np.asarray(feats)
    return c[o], f[o]


def rel_err(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elem_close(a, b, rtol=1e-4, atol=1e-6):
    """Element-wise  |a - b| <= rtol |b| + atol  (the strict reading of "1e-4 relative": every element, not only against
    the largest one as rel_err does).  Returns (ok, worst relative deviation) for the assertion message."""
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    d = (a - b).abs()
    return bool((d <= rtol * b.abs() + atol).all()), float((d / b.abs().clamp_min(atol)).max())
