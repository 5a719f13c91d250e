"""Input side of the hot path.

* ``larcvsparse_to_scnsparse_3d / _2d`` -- host transforms with the semantics of
  /root/reference/src/io/data_transforms.py:21-49,198-253: strip the ``-999`` padding of a larcv
  batch ``[B, planes, MaxVoxels, D+1]`` and emit ``(coords float64 [N,4], features fp32 [N,1], B)``
  (3-D: batch-major ``(x,y,z,batch)``; 2-D: plane-major ``(plane,y,x,batch)``).
* ``synthetic_dune3d_batch`` -- the seeded track+shower generator of SURVEY.md 8(d) used by bench.py
  (there is no larcv3 / dataset in this environment): exactly ``n_voxels`` active voxels per event,
  larcv ``[B,1,n_voxels,4]`` layout, features normalised ``(x-1)/0.5`` (larcv_fetcher.py:100-108).
"""
import numpy as np

PAD = -999


def larcvsparse_to_scnsparse_3d(input_array):
    arr = np.asarray(input_array)
    batch_size = arr.shape[0]
    values = arr[..., -1]
    idx = np.nonzero(values != PAD)          # (batch, plane, voxel), C-order == reference order
    features = values[idx][:, None]
    cols = [arr[..., d][idx] for d in range(arr.shape[-1] - 1)]
    cols.append(idx[0])
    coords = np.stack(cols, axis=-1)         # float32 columns + int64 batch -> float64, as upstream
    return coords, features, batch_size


def larcvsparse_to_scnsparse_2d(input_array):
    arr = np.asarray(input_array)
    batch_size, n_planes = arr.shape[0], arr.shape[1]
    coords, feats = [], []
    for p in range(n_planes):
        plane = arr[:, p]
        b, v = np.nonzero(plane[..., 2] != PAD)
        x, y = plane[b, v, 0], plane[b, v, 1]
        coords.append(np.stack([np.full(x.shape, p), y, x, b], axis=-1))
        feats.append(plane[b, v, 2][:, None])
    return [np.concatenate(coords), np.concatenate(feats), batch_size]


# ---------------------------------------------------------------------------------- synthetic
def _unit(rng, n):
    v = rng.standard_normal((n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def synthetic_event(seed, grid=(900, 900, 900), n_voxels=50000):
    """One event: 6-13 straight tracks + 2-4 showers from a common vertex."""
    rng = np.random.default_rng(seed)
    g = np.asarray(grid, dtype=np.float64)
    vertex = g * rng.uniform(0.3, 0.7, 3)
    pts = []
    for d in _unit(rng, int(rng.integers(6, 14))):
        length = rng.uniform(150, 600)
        s = np.arange(0.0, length, 0.25)
        pts.append(vertex + s[:, None] * d + rng.normal(0, 0.7, (len(s), 3)))
    axes = _unit(rng, int(rng.integers(2, 5)))

    def shower_points(n):
        out = []
        per = n // len(axes) + 1
        for a in axes:
            s = rng.gamma(3.0, 40.0, per)
            # two unit vectors orthogonal to the axis
            u = np.cross(a, [1.0, 0.0, 0.0])
            if np.linalg.norm(u) < 1e-3:
                u = np.cross(a, [0.0, 1.0, 0.0])
            u /= np.linalg.norm(u)
            w = np.cross(a, u)
            sig = 2.0 + 0.08 * s
            out.append(vertex + s[:, None] * a + (rng.normal(0, 1, per) * sig)[:, None] * u
                       + (rng.normal(0, 1, per) * sig)[:, None] * w)
        return np.concatenate(out)

    def voxelise(p):
        v = np.floor(p).astype(np.int64)
        v = np.clip(v, 0, (g - 1).astype(np.int64))
        key = (v[:, 0] * grid[1] + v[:, 1]) * grid[2] + v[:, 2]
        return np.unique(key)

    keys = voxelise(np.concatenate(pts))
    while len(keys) < n_voxels:
        keys = np.union1d(keys, voxelise(shower_points(max(n_voxels - len(keys), 2000) * 2)))
    if len(keys) > n_voxels:
        keys = np.sort(rng.choice(keys, n_voxels, replace=False))
    x, rem = np.divmod(keys, grid[1] * grid[2])
    y, z = np.divmod(rem, grid[2])
    energy = rng.lognormal(0.0, 0.5, n_voxels)
    val = (energy - 1.0) / 0.5
    return np.stack([x, y, z, val], axis=-1).astype(np.float32)


def uniform_event(seed, grid=(900, 900, 900), n_voxels=50000):
    """Uniform-random cloud (R/N ~ 1): worst case for hashing, trivial for the GEMM."""
    rng = np.random.default_rng(seed)
    vol = grid[0] * grid[1] * grid[2]
    keys = np.unique(rng.integers(0, vol, int(n_voxels * 1.05)))
    while len(keys) < n_voxels:
        keys = np.union1d(keys, rng.integers(0, vol, n_voxels))
    keys = np.sort(rng.choice(keys, n_voxels, replace=False))
    x, rem = np.divmod(keys, grid[1] * grid[2])
    y, z = np.divmod(rem, grid[2])
    val = (rng.lognormal(0.0, 0.5, n_voxels) - 1.0) / 0.5
    return np.stack([x, y, z, val], axis=-1).astype(np.float32)


def synthetic_dune3d_batch(rank, batch_index, batch_size=64, grid=(900, 900, 900), n_voxels=50000, kind="tracks",
                           n_classes=(3, 3, 2, 2)):
    """larcv-layout batch ``[B,1,n_voxels,4]`` fp32 + labels. seed = 1000003*rank + 64*batch + event."""
    gen = synthetic_event if kind == "tracks" else uniform_event
    arr = np.empty((batch_size, 1, n_voxels, 4), np.float32)
    for e in range(batch_size):
        arr[e, 0] = gen(1000003 * rank + 64 * batch_index + e, grid, n_voxels)
    rng = np.random.default_rng(7 + 1000003 * rank + batch_index)
    names = ("labelneutID", "labelprotID", "labelnpiID", "labelcpiID")
    labels = {k: rng.integers(0, n, batch_size).astype(np.int64) for k, n in zip(names, n_classes)}
    return arr, labels


def synthetic_sbnd_event(seed, planes=3, hw=(2048, 1280), n_pix=15000):
    """One SBND-like event: per wire plane a 2-D projection of tracks + blobs, exactly ``n_pix`` distinct
    pixels; larcv 2-D layout [planes, n_pix, 3] = (x, y, value) (larcv_scripts/preprocess_fullres_2D.cfg)."""
    rng = np.random.default_rng(seed)
    H, W = hw
    out = np.empty((planes, n_pix, 3), np.float32)
    vertex = np.array([H, W]) * rng.uniform(0.3, 0.7, 2)
    for p in range(planes):
        pts = []
        for _ in range(int(rng.integers(6, 14))):
            ang = rng.uniform(0, 2 * np.pi)
            s = np.arange(0.0, rng.uniform(150, 900), 0.25)
            pts.append(vertex + s[:, None] * np.array([np.sin(ang), np.cos(ang)]) + rng.normal(0, 0.8, (len(s), 2)))
        keys = np.empty(0, np.int64)
        pts = np.concatenate(pts)
        while True:
            v = np.clip(np.floor(pts).astype(np.int64), 0, [H - 1, W - 1])
            keys = np.union1d(keys, v[:, 0] * W + v[:, 1])
            if len(keys) >= n_pix:
                break
            c = vertex + rng.normal(0, 120, 2)
            pts = c + rng.normal(0, rng.uniform(5, 40), (2 * (n_pix - len(keys)) + 500, 2))
        keys = np.sort(rng.choice(keys, n_pix, replace=False))
        out[p, :, 1] = keys // W            # y (row, < 2048)
        out[p, :, 0] = keys % W             # x (col, < 1280)
        out[p, :, 2] = (rng.lognormal(0.0, 0.5, n_pix) - 1.0) / 0.5
    return out
