#!/usr/bin/env python3
"""What-if statistics (CPU only) for the row-band kernel: per 64-row band and filter offset the number of 16-slot
sub-chunks, and how well NCW bands that share one weight stream (a ring of D slices, NS slices per offset) keep busy:
   python tools/band_stats.py --events 4 --levels 1 2 3 4"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def pack(c):
    return ((c[:, 3].astype(np.int64) << 48) | (c[:, 0].astype(np.int64) << 32) | (c[:, 1].astype(np.int64) << 16) | c[:, 2].astype(np.int64))


def simulate(ms, ns, D):
    """ms [NCW][K] sub-chunks per (band, offset) of the offsets in the union.  Time unit = one sub-chunk-slice of MFMAs.
    A band cannot start slice i before the slowest band has finished slice i-D (ring of D slots, loader never late)."""
    ncw, K = ms.shape
    F = K * ns
    cost = np.repeat(ms, ns, axis=1).astype(np.float64)   # [NCW][F]
    done = np.zeros((ncw, F + 1))
    for i in range(F):
        gate = done[:, i - D + 1].max() if i >= D else 0.0
        done[:, i + 1] = np.maximum(done[:, i], gate) + cost[:, i]
    return done[:, F].max()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=4)
    ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3, 4])
    a = ap.parse_args()
    from bench import generate_batch
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", 8)
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    c = c.astype(np.int64)
    _, first = np.unique(pack(c), return_index=True)
    c = c[np.sort(first)]
    planes = [32, 64, 96, 128, 160, 192]
    for L in range(max(a.levels) + 1):
        if L > 0:
            c = c.copy(); c[:, :3] //= 2
            _, first = np.unique(pack(c), return_index=True)
            c = c[np.sort(first)]
        if L not in a.levels:
            continue
        n = len(c)
        keys = pack(c)
        srt = np.argsort(keys); ks = keys[srt]
        nt64 = (n + 63) // 64
        cnt = np.zeros((nt64, 27), np.int64)
        k = 0
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    q = c.copy(); q[:, 0] += dx; q[:, 1] += dy; q[:, 2] += dz
                    kq = pack(q)
                    pos = np.searchsorted(ks, kq); pos[pos >= n] = n - 1
                    hit = (ks[pos] == kq) & (q[:, :3] >= 0).all(1)
                    np.add.at(cnt[:, k], np.nonzero(hit)[0] // 64, 1)
                    k += 1
        sub = (cnt + 15) // 16
        R = cnt.sum()
        ns = planes[L] // 16
        print("level %d (C=%d): rows %d, rules/row %.2f, 64-row bands: sub-chunks/offset %.2f, padding %.3f"
              % (L, planes[L], n, R / n, sub.mean(), R / (16.0 * sub.sum())))
        for ncw in (2, 4, 8):
            g = nt64 // ncw
            s = sub[:g * ncw].reshape(g, ncw, 27)
            tot = s.sum()
            inf = (s.sum(2).max(1) * ncw).sum()
            line = "   NCW=%d: balance (deep ring) %.3f" % (ncw, tot / inf)
            idx = np.random.default_rng(0).choice(g, min(g, 300), replace=False)
            for D in (2, 4, 8, 16, 32):
                t = w = 0.0
                for gi in idx:
                    m = s[gi]
                    m = m[:, m.sum(0) > 0]
                    t += simulate(m, ns, D) * ncw
                    w += m.sum() * ns
                line += "  D=%d %.3f" % (D, w / t)
            print(line)


if __name__ == "__main__":
    main()
