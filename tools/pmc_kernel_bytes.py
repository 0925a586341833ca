#!/usr/bin/env python3
"""Mean FETCH_SIZE / WRITE_SIZE (KiB as reported by rocprofv3 -> bytes) per launch of the kernels whose name
contains argv[1], from the counter_collection.csv files under the directories argv[2:].
FETCH_SIZE reads 1/2 on gfx950 (see tools/parse_pmc.py): the x2 correction is applied here."""
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
    for k, v in agg.items():
        corr = 1.997 if k == "FETCH_SIZE" else 1.0
        print("%s %s: %d launches, mean %.1f MB per launch (min %.1f, max %.1f)" % (
            d, k, len(v), sum(v) / len(v) * 1024 * corr / 1e6, min(v) * 1024 * corr / 1e6, max(v) * 1024 * corr / 1e6))
