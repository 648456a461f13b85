#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = defaultdict(float)
    meta = {}
    for row in csv.DictReader(open(f)):
        key = (row["Dispatch_Id"], row["Kernel_Name"], row["Counter_Name"])
        per_dispatch[key] += float(row["Counter_Value"])
    for (d, k, c), v in per_dispatch.items():
        acc[k][c].append(v)
for k, cs in acc.items():
    short = k.split("(")[0][-60:]
    print("==", short)
    for c, vs in sorted(cs.items()):
        print("   %-28s n=%-3d mean=%.6g" % (c, len(vs), sum(vs) / len(vs)))
