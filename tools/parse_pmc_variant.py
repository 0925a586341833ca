#!/usr/bin/env python3
import csv, glob, sys, collections
for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "step_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    waves = 16384.0
    row = {k: sum(v[-20:]) / len(v[-20:]) for k, v in agg.items()}
    print(d, " ".join("%s=%.0f" % (k.replace("SQ_", ""), (v * (4 if "CYCLES" in k or "WAIT" in k or "ACTIVE" in k else 1)) / waves)
                      for k, v in sorted(row.items())))
