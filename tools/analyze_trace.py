import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 kernels before the big block of 1000 timing launches: find index of the 26th step kernel from the start of stepping
names = [r["Kernel_Name"] for r in rows]
steps = [i for i, n in enumerate(names) if "step_kernel" in n]
# warmup 5 + timed 20 = first 25 step kernels after reset
first = steps[0]
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + 60]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f us  +%7.1f us  %s" % (s / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
