#!/usr/bin/env python3
"""Average kernel durations of a rocprofv3 --kernel-trace --stats run: tools/kstats.py <dir> [name-substring ...]"""
import csv
import glob
import sys

d = sys.argv[1]
subs = sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if not subs or any(s in n for s in subs):
            print(f"{float(r['AverageNs']) / 1e6:9.4f} ms x {r['Calls']:>5s}  {n[:110]}")
