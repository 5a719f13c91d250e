#!/usr/bin/env python3
"""What-if statistics (CPU only): efficiency of DST-ALIGNED 16-row MFMA blocks -- rule slot i of a block IS dst row
16q+i, so accumulators could stay in registers -- for several row orders, next to the compacted sub-chunks the
kernels use today.   python tools/group_stats.py --events 4 --levels 1 2 3"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def pack(c):
    return ((c[:, 3].astype(np.int64) << 48) | (c[:, 0].astype(np.int64) << 32) | (c[:, 1].astype(np.int64) << 16) | c[:, 2].astype(np.int64))


def spread(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--events", type=int, default=4)
    ap.add_argument("--levels", type=int, nargs="*", default=[1, 2, 3])
    a = ap.parse_args()
    from bench import generate_batch
    from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
    arr, _ = generate_batch(0, 0, a.events, (900, 900, 900), 50000, "tracks", 8)
    c, f, b = larcvsparse_to_scnsparse_3d(arr)
    c = c.astype(np.int64)
    # first-appearance unique (InputLayer)
    _, first = np.unique(pack(c), return_index=True)
    c = c[np.sort(first)]
    for L in range(max(a.levels) + 1):
        if L > 0:
            c = c.copy(); c[:, :3] //= 2
            _, first = np.unique(pack(c), return_index=True)
            c = c[np.sort(first)]
        if L not in a.levels:
            continue
        n = len(c)
        orders = {"input": np.arange(n)}
        orders["morton"] = np.lexsort(((spread(c[:, 0]) << 2) | (spread(c[:, 1]) << 1) | spread(c[:, 2]), c[:, 3]))
        orders["lex"] = np.lexsort((c[:, 2], c[:, 1], c[:, 0], c[:, 3]))
        orders["block4"] = np.lexsort((c[:, 2] & 3, c[:, 1] & 3, c[:, 0] & 3, c[:, 2] >> 2, c[:, 1] >> 2, c[:, 0] >> 2, c[:, 3]))
        for name, o in orders.items():
            cc = c[o]
            keys = pack(cc)
            srt = np.argsort(keys); ks = keys[srt]
            R = 0; G16 = 0; SUB64 = 0; SUB128 = 0; G8 = 0; G32 = 0
            crit = {}
            for dx in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dz in (-1, 0, 1):
                        q = cc.copy(); q[:, 0] += dx; q[:, 1] += dy; q[:, 2] += dz
                        ok = (q[:, :3] >= 0).all(1)
                        kq = pack(q)
                        pos = np.searchsorted(ks, kq); pos[pos >= n] = n - 1
                        hit = ok & (ks[pos] == kq)
                        rows = np.nonzero(hit)[0]
                        R += len(rows)
                        G16 += len(np.unique(rows // 16))
                        G8 += len(np.unique(rows // 8))
                        G32 += len(np.unique(rows // 32))
                        SUB64 += ((np.bincount(rows // 64) + 15) // 16).sum()
                        SUB128 += ((np.bincount(rows // 128) + 15) // 16).sum()
                        if name == "input":
                            for TR in (64, 128, 256):
                                ns = (np.bincount(rows // TR) + 15) // 16
                                ns = ns[ns > 0]
                                for NT in (4, 6):
                                    d = crit.setdefault((TR, NT), [0, 0, 0, 0])
                                    d[0] += (((ns + 3) // 4) * NT).sum()      # whole sub-chunks dealt to 4 waves
                                    d[1] += ((ns * NT + 3) // 4).sum()        # (sub-chunk, column tile) units dealt contiguously
                                    d[2] += (ns * NT).sum() / 4.0             # perfect
                                    d[3] += len(ns)
            print("L%d N=%d R/N=%.2f order %-7s dst-aligned blocks: eff16 %.3f (eff8 %.3f eff32 %.3f) | compacted sub-chunks: 64-row %.3f 128-row %.3f"
                  % (L, n, R / n, name, R / (16.0 * G16), R / (8.0 * G8), R / (32.0 * G32), R / (16.0 * SUB64), R / (16.0 * SUB128)), flush=True)
            for (TR, NT), d in sorted(crit.items()):
                print("    TR=%d NT=%d: critical path per phase (column-tile units): whole sub-chunks %.2f | unit dealing %.2f | perfect %.2f  (phases/tile-offset %d)"
                      % (TR, NT, d[0] / d[3], d[1] / d[3], d[2] / d[3], d[3]))


if __name__ == "__main__":
    main()
