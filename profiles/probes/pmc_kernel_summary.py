#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run: per kernel name, the mean of each counter over its dispatches.
usage: pmc_kernel_summary.py <dir with *_counter_collection.csv> [kernel substring ...]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
want = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_\w+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        if want and not any(w in k for w in want):
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (s, n) in sorted(acc[k].items()):
        print("   %-28s mean %14.1f  over %d dispatches" % (c, s / n, n))
