#!/usr/bin/env python3
"""Mean raw counter values per launch of the kernels whose name contains argv[1] (directories argv[2:])."""
import csv
import glob
import sys

pat = sys.argv[1]
for d in sys.argv[2:]:
    agg = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%s %s: %d launches, mean %.4g (min %.4g, max %.4g)" % (d, k, len(v), sum(v) / len(v), min(v), max(v)))
