#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-24s n=%-5d mean=%.6g" % (c, len(v), sum(v) / len(v)))
