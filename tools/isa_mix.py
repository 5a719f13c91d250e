#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc's device assembly (hipcc ... --cuda-device-only -S -o x.s): for every
kernel whose name contains PATTERN, the instructions between the first and the last MFMA by class (MFMA / other VALU / SALU /
LDS / vector memory / waits + barriers), and VALU issue cycles per MFMA (4 per plain VALU instruction, 8 of an MFMA's 32).
  python tools/isa_mix.py /tmp/conv_fwd.s mround"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN3sei\S+):', s, re.M)]
for (a, name), nxt in zip(starts, starts[1:] + [(len(s), None)]):
    if pat not in name:
        continue
    lines = [l.strip() for l in s[a:nxt[0]].split('\n')]
    ins = [l for l in lines if l and not l.startswith(('.', ';', '_Z')) and not l.endswith(':')]
    mf = [i for i, l in enumerate(ins) if l.startswith('v_mfma')]
    if not mf:
        continue
    body = ins[mf[0]:mf[-1] + 1]
    cls = {"mfma": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "wait": 0, "other": 0}
    for l in body:
        op = l.split()[0]
        if op.startswith('v_mfma'): cls["mfma"] += 1
        elif op.startswith('v_'): cls["valu"] += 1
        elif op.startswith(('s_waitcnt', 's_barrier', 's_nop')): cls["wait"] += 1
        elif op.startswith('s_'): cls["salu"] += 1
        elif op.startswith('ds_'): cls["lds"] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): cls["vmem"] += 1
        else: cls["other"] += 1
    nm = re.sub(r'^_ZN3sei\d+', '', name); nm = re.sub(r'EEEvPK.*', '', nm)
    print("%-46s %s  valu/mfma %.1f  (valu issue cycles per MFMA ~ %.0f of 24 free)" % (
        nm, " ".join("%s %d" % kv for kv in cls.items()), cls["valu"] / cls["mfma"], 4.0 * cls["valu"] / cls["mfma"]))
