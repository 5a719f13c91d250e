#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per (kernel, counter):  python tools/pmc_sum.py DIR [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(float)
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if filt and filt not in k:
                    continue
                k = k.split("(")[0][:90]
                c = row.get("Counter_Name", "")
                acc[(k, c)] += float(row.get("Counter_Value", 0) or 0)
                cnt[(k, c)] += 1
    last = None
    for (k, c) in sorted(acc):
        if k != last:
            print(k)
            last = k
        print("   %-34s %18.0f   (%d dispatches)" % (c, acc[(k, c)], cnt[(k, c)]))


if __name__ == "__main__":
    main()
