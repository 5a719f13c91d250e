#!/usr/bin/env python3
"""What-if statistics (CPU only): how full the eight-wave rounds of the workgroup-tile kernels are when a round may span
1 (conv_tile_mfma_kernel), 2 (conv_round_mfma_kernel), 3 or any number of consecutive offsets of a 128-row tile.
  python tools/round_stats.py --events 4 --levels 2 3 4"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def pack(c):
    return ((c[:, 3].astype(np.int64) << 48) | (c[:, 0].astype(np.int64) << 32) | (c[:, 1].astype(np.int64) << 16) | c[:, 2].astype(np.int64))


def rounds_of(subs, span, nw):
    """subs: sub-chunks per nonempty offset of one tile, in offset order -> number of rounds"""
    r = 0
    i = 0
    left = subs[0] if len(subs) else 0
    while i < len(subs):
        room, used = nw, 0
        while i < len(subs) and room > 0 and used < span:
            t = min(room, left)
            room -= t; left -= t; used += 1
            if left == 0:
                i += 1
                left = subs[i] if i < len(subs) else 0
            else:
                break
        r += 1
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=4)
    ap.add_argument("--levels", type=int, nargs="*", default=[2, 3, 4])
    ap.add_argument("--tile", type=int, default=128)
    a = ap.parse_args()
    from bench import generate_batch
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", 8)
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    c = c.astype(np.int64)
    _, first = np.unique(pack(c), return_index=True)
    c = c[np.sort(first)]
    nw = a.tile // 16
    for L in range(max(a.levels) + 1):
        if L > 0:
            c = c.copy(); c[:, :3] //= 2
            _, first = np.unique(pack(c), return_index=True)
            c = c[np.sort(first)]
        if L not in a.levels:
            continue
        n = len(c)
        keys = pack(c); srt = np.argsort(keys); ks = keys[srt]
        ntile = (n + a.tile - 1) // a.tile
        cnt = np.zeros((ntile, 27), dtype=np.int64)
        k = 0
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    q = c.copy(); q[:, 0] += dx; q[:, 1] += dy; q[:, 2] += dz
                    kq = pack(q)
                    pos = np.searchsorted(ks, kq); pos[pos >= n] = n - 1
                    hit = (ks[pos] == kq) & (q[:, :3] >= 0).all(1)
                    cnt[:, k] = np.bincount(np.nonzero(hit)[0] // a.tile, minlength=ntile)
                    k += 1
        subs = (cnt + 15) // 16
        tot = int(subs.sum())
        print("level %d: %d rows, %d tiles of %d rows, %.1f rules / row, %.1f sub-chunks / tile, %.2f / (tile, nonempty offset), padding %.2f" % (
            L, n, ntile, a.tile, cnt.sum() / n, tot / ntile, tot / max((subs > 0).sum(), 1), cnt.sum() / (16.0 * tot)))
        for span in (1, 2, 3, 4, 27):
            r = sum(rounds_of([int(v) for v in row if v > 0], span, nw) for row in subs)
            print("   rounds spanning <= %2d offsets: %6.1f rounds / tile, fill %.2f" % (span, r / ntile, tot / (float(nw) * r)))


if __name__ == "__main__":
    main()
