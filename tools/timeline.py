#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv of bench.py and prints, for the LAST full step: per queue busy time, time with 1 / 2 / 3
queues busy, idle time, and the order in which the queues fall silent at the end of the backward pass."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
ev.sort()
# steps are delimited by the fused Adam kernel (multi_tensor_apply ... adam)
adam = [e for e in ev if "adam" in e[3].lower()]
ends = sorted(set(e[1] for e in adam))
# group adam launches closer than 5 ms
marks = []
for t in ends:
    if not marks or t - marks[-1] > 5e6:
        marks.append(t)
    else:
        marks[-1] = t
if len(marks) < 2:
    print("fewer than two optimizer steps in the trace"); sys.exit(0)
t0, t1 = marks[-2], marks[-1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("last step: %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
queues = sorted(set(e[2] for e in step))
busy = defaultdict(float)
for s, e, q, n in step:
    busy[q] += (e - s) / 1e6
for q in queues:
    ks = [x for x in step if x[2] == q]
    top = defaultdict(float)
    for s, e, _, n in ks:
        top[n.split("(")[0].replace("void ", "")[:48]] += (e - s) / 1e6
    print("queue %s: %d kernels, sum of durations %.1f ms, first start +%.1f ms, last end +%.1f ms" % (
        q, len(ks), busy[q], (ks[0][0] - t0) / 1e6, (max(k[1] for k in ks) - t0) / 1e6))
    for n, v in sorted(top.items(), key=lambda kv: -kv[1])[:4]:
        print("      %-50s %7.1f ms" % (n, v))
for pat in ("probe_kernel", "tile_rules_kernel", "bn_bwd_apply", "colreduce", "conv_dw"):
    per = defaultdict(int)
    for s_, e_, q, n in step:
        if pat in n:
            per[q] += 1
    print("   %-20s launches per queue: %s" % (pat, dict(per)))
# concurrency profile (sweep line)
pts = []
for s, e, q, n in step:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
lvl, last, hist = 0, t0, defaultdict(float)
for t, d in pts:
    hist[min(lvl, 3)] += (t - last) / 1e6
    last = t
    lvl += d
hist[0] += (t1 - last) / 1e6
print("time with 0 / 1 / 2 / >=3 kernels in flight: %.1f / %.1f / %.1f / %.1f ms" % (hist[0], hist[1], hist[2], hist[3]))
# forward / backward boundary: first dW kernel
dws = [e for e in step if "conv_dw" in e[3]]
if dws:
    print("first dW starts at +%.1f ms (forward pass + loss), last dW ends at +%.1f ms" % ((dws[0][0] - t0) / 1e6, (max(d[1] for d in dws) - t0) / 1e6))
    mainq = max(queues, key=lambda q: sum(1 for x in step if x[2] == q))
    dx = [e for e in step if e[2] == mainq and ("conv_tile" in e[3] or "conv_wave" in e[3]) and e[0] > dws[0][0]]
    if dx:
        print("last dX kernel on the main queue ends at +%.1f ms" % ((max(d[1] for d in dx) - t0) / 1e6))
