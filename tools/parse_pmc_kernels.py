#!/usr/bin/env python3
"""Per-wave averages of SQ counters for kernels whose name contains a substring:
   parse_pmc_kernels.py <substring> <n_waves> <dir> [<dir> ...]
(SQ cycle counters are in quad-cycles on gfx950: scaled by 4 here.)"""
import collections
import csv
import glob
import sys

sub, waves = sys.argv[1], float(sys.argv[2])
for d in sys.argv[3:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    row = {k: sum(v[-4:]) / len(v[-4:]) for k, v in agg.items()}
    print(sub, d.split("/")[-1], " ".join("%s=%.0f" % (k.replace("SQ_", ""), (v * (4 if "CYCLES" in k or "WAIT" in k or "ACTIVE" in k else 1)) / waves)
                                          for k, v in sorted(row.items())))
