#!/usr/bin/env python3
"""Where do a kernel's register spills sit?  Reads hipcc's device assembly (hipcc ... --cuda-device-only -S -o x.s) and prints,
per kernel whose name contains PATTERN: lines, MFMA count, scratch instructions, and how many of those lie between the first
and the last MFMA (i.e. inside the main loop rather than in the prologue / epilogue).
  python tools/isa_scratch.py /tmp/conv_fwd.s mround"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN3sei\S+):', s, re.M)]
for (a, name), nxt in zip(starts, starts[1:] + [(len(s), None)]):
    if pat not in name:
        continue
    lines = s[a:nxt[0]].split('\n')
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    if not mf:
        continue
    inside = [i for i in sc if mf[0] < i < mf[-1]]
    nm = re.sub(r'^_ZN3sei\d+', '', name); nm = re.sub(r'EEEvPK.*', '', nm)
    print("%-48s lines %6d mfma %5d scratch %4d inside-loop %4d" % (nm, len(lines), len(mf), len(sc), len(inside)))
