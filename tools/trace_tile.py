#!/usr/bin/env python3
"""Timeline of ONE compute unit while the workgroup-tile conv kernel runs (diagnostic build:
make -C sparseeventid_amd/csrc -B EXTRA="-DSEI_EXPERIMENTS -DSEI_TRACE").
   python tools/trace_tile.py LEVEL C [EVENTS] [--dump FILE]
Every workgroup that ran on the traced CU wrote, per wave and phase, s_memtime at barrier exit (tag 0), loads issued (1),
MFMA block done (2 | ms << 2) and, per offset, write-back done (3)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import generate_batch

lvl, c = int(sys.argv[1]), int(sys.argv[2])
events = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 64
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
os.environ.setdefault("SEI_CONV_KERNEL", "tile")
arr, _ = generate_batch(0, 0, events, (900, 900, 900), 50000, "tracks", 16)
import sparseeventid_amd.scn as scn
from sparseeventid_amd import _lib
from sparseeventid_amd.data import larcvsparse_to_scnsparse_3d
cc, f, b = larcvsparse_to_scnsparse_3d(arr)
t = scn.InputLayer(3, [900] * 3)((torch.from_numpy(cc), torch.from_numpy(f).cuda(), b))
t.features = torch.randn(t.features.shape[0], 16, device="cuda")
for _ in range(lvl):
    t = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(t)
t.features = torch.randn(t.features.shape[0], c, device="cuda")
conv = scn.SubmanifoldConvolution(3, c, c, 3, True).cuda()
for _ in range(3):
    conv(t)
torch.cuda.synchronize()
l = ctypes.CDLL(_lib.LIB_PATH)
l.sei_debug_trace_bytes.restype = ctypes.c_longlong
l.sei_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]
nbytes = l.sei_debug_trace_bytes()
assert l.sei_debug_trace(None, 0, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); conv(t); e1.record(); torch.cuda.synchronize()
raw = np.zeros(nbytes, np.uint8)
assert l.sei_debug_trace(raw.ctypes.data, nbytes, 0) == 0
if dump:
    raw.tofile(dump)
S, W, E = 96, 8, 512
hdr = raw[:8].view(np.int32)
cu_key, n_slots = int(hdr[0]), min(int(hdr[1]), S)
o = 8
hw = raw[o:o + S * W * 4].view(np.int32).reshape(S, W); o += S * W * 4
nev = raw[o:o + S * W * 4].view(np.int32).reshape(S, W); o += S * W * 4
vid = raw[o:o + S * 4].view(np.int32); o += S * 4
o = (o + 7) // 8 * 8
t0 = raw[o:o + S * 8].view(np.uint64); o += S * 8
t1 = raw[o:o + S * 8].view(np.uint64); o += S * 8
ev = raw[o:o + S * W * E * 8].view(np.uint64).reshape(S, W, E)
print("kernel %.3f ms; traced CU key 0x%x (xcc %d, se/sh/cu bits 0x%02x); %d workgroups ran on it" % (
    e0.elapsed_time(e1), cu_key, cu_key >> 16, (cu_key >> 8) & 0xFF, n_slots))
base = int(t0[:n_slots].min())
end = int(t1[:n_slots].max())
span = end - base
print("CU timeline spans %d ticks (s_memtime; 100 MHz if constant-rate, else shader clock)" % span)
nw = int((nev[0] > 0).sum())
# per workgroup
rows = []
simd_busy = {sd: [] for sd in range(4)}     # (start, end, slot, wave) intervals of MFMA blocks with ms > 0
simd_all = {sd: [] for sd in range(4)}
stat = dict(issue=[], mfma=[], mfma_idle=[], wb=[], wait=[], phase=[])
last_reason = {"mfma": 0, "wb": 0, "issue": 0, "idle": 0}
for s in range(n_slots):
    per_wave = []
    for w in range(nw):
        n = int(nev[s, w])
        tt = (ev[s, w, :n] >> np.uint64(8)).astype(np.int64) - base
        tag = (ev[s, w, :n] & np.uint64(0xFF)).astype(np.int64)
        per_wave.append((tt, tag))
        sd = (int(hw[s, w]) >> 4) & 3
        i = 0
        prev_end = None
        while i < n:
            if tag[i] & 3 == 0 and i + 2 < n and tag[i + 1] == 1 and (tag[i + 2] & 3) == 2:
                b0, b1, b2 = tt[i], tt[i + 1], tt[i + 2]
                ms = tag[i + 2] >> 2
                if prev_end is not None:
                    stat["wait"].append(b0 - prev_end)
                stat["issue"].append(b1 - b0)
                (stat["mfma"] if ms else stat["mfma_idle"]).append(b2 - b1)
                if ms:
                    simd_busy[sd].append((b1, b2, s, w, ms))
                simd_all[sd].append((b0, b2, s, w))
                prev_end = b2
                i += 3
                if i < n and tag[i] == 3:
                    stat["wb"].append(tt[i] - b2)
                    prev_end = tt[i]
                    i += 1
            else:
                i += 1
    rows.append((s, int(vid[s]), int(t0[s]) - base, int(t1[s]) - base, [(int(hw[s, w]) >> 4) & 3 for w in range(nw)]))
print("workgroups on the CU (slot, work item, start, end, duration, SIMD of waves 0..):")
for r in rows[:12]:
    print("   %3d vid %6d  %9d .. %9d  (%7d)  simd %s" % (r[0], r[1], r[2], r[3], r[3] - r[2], r[4]))
dur = np.array([r[3] - r[2] for r in rows])
print("workgroup duration: mean %.0f  min %d  max %d; resident workgroups on average %.2f" % (dur.mean(), dur.min(), dur.max(), dur.sum() / span))
for k in ("issue", "mfma", "mfma_idle", "wb", "wait"):
    a = np.array(stat[k]) if stat[k] else np.zeros(1)
    print("  per phase, %-10s n=%7d mean %7.1f  p50 %6.0f  p90 %6.0f  p99 %6.0f" % (k, len(a), a.mean(), np.percentile(a, 50), np.percentile(a, 90), np.percentile(a, 99)))
# per SIMD: how many waves are inside an MFMA block (ms > 0) at a time
print("per SIMD: share of the CU's span with n waves inside an MFMA block (ms > 0); the block durations include their share of pipe contention")
for sd in range(4):
    evs = []
    for (a, b_, s, w, ms) in simd_busy[sd]:
        evs.append((a, 1)); evs.append((b_, -1))
    evs.sort()
    hist = np.zeros(9)
    cur, lastt = 0, 0
    for tt, d in evs:
        hist[min(cur, 8)] += tt - lastt
        lastt = tt
        cur += d
    hist[0] += span - lastt
    need = sum(ms_ * 1 for (_, _, _, _, ms_) in simd_busy[sd])
    print("   SIMD %d: " % sd + "  ".join("%d:%.3f" % (i, hist[i] / span) for i in range(5)) + "   MFMA blocks %d" % len(simd_busy[sd]))
# which wave arrives last at each barrier, and what it did in that phase
print("last arriver at each barrier (per workgroup phase): what the phase's critical wave spent")
crit = dict(n=0, issue=0.0, mfma=0.0, wb=0.0, early_wait=0.0, ms=[])
for s in range(n_slots):
    seqs = []
    for w in range(nw):
        n = int(nev[s, w])
        tt = (ev[s, w, :n] >> np.uint64(8)).astype(np.int64) - base
        tag = (ev[s, w, :n] & np.uint64(0xFF)).astype(np.int64)
        ph = []
        i = 0
        while i + 2 < n:
            if tag[i] == 0 and tag[i + 1] == 1 and (tag[i + 2] & 3) == 2:
                endt = tt[i + 2]; wb = 0
                j = i + 3
                if j < n and tag[j] == 3:
                    wb = tt[j] - tt[i + 2]; endt = tt[j]; j += 1
                ph.append((tt[i], tt[i + 1] - tt[i], tt[i + 2] - tt[i + 1], wb, endt, tag[i + 2] >> 2))
                i = j
            else:
                i += 1
        seqs.append(ph)
    nph = min(len(p) for p in seqs)
    for p in range(nph - 1):
        ends = [seqs[w][p][4] for w in range(nw)]
        wl = int(np.argmax(ends))
        b0, iss, mf, wb, endt, ms = seqs[wl][p]
        nxt = min(seqs[w][p + 1][0] for w in range(nw))
        crit["n"] += 1; crit["issue"] += iss; crit["mfma"] += mf; crit["wb"] += wb
        crit["early_wait"] += nxt - endt
        crit["ms"].append(ms)
        stat["phase"].append(seqs[0][p + 1][0] - seqs[0][p][0])
n = max(crit["n"], 1)
ph = np.array(stat["phase"])
print("   phases %d: mean length %.0f ticks (p50 %.0f p90 %.0f); critical wave: issue %.0f  MFMA block %.0f  write-back %.0f  arrive->release %.0f; it held sub-chunks: %s" % (
    crit["n"], ph.mean(), np.percentile(ph, 50), np.percentile(ph, 90), crit["issue"] / n, crit["mfma"] / n, crit["wb"] / n, crit["early_wait"] / n,
    np.bincount(np.array(crit["ms"]), minlength=3).tolist()))
