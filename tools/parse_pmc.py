#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of tools/pmc_probe.py into per-launch HBM
bytes of the step kernel, calibrated on refresh_kernel whose traffic is known
(reads the board planes -- 8 words per env when packed, else 10 -- + 8 B meta, writes 8 B meta + 1 B n_valid per env).

Per MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE under-reports wide streaming reads and other widths are uncalibrated,
so the known-traffic kernel of the same access pattern sets the correction factor.
"""
import csv
import glob
import json
import os
import sys


def load(dirname, counter):
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def mean_tail(vals, n):
    vals = vals[-n:]
    return sum(vals) / len(vals)


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    B = 1 << 20
    word = int(os.environ.get("PROBE_WORD_BYTES", "4"))
    out = {"envs": B, "columns": 10, "rows": int(os.environ.get("PROBE_ROWS", "20")), "pieces": "default",
           "unit_note": "FETCH_SIZE/WRITE_SIZE reported in KiB by rocprofv3"}
    f = load(fetch_dir, "FETCH_SIZE")
    w = load(write_dir, "WRITE_SIZE")
    rows = int(os.environ.get("PROBE_ROWS", "20"))
    planes = 8 if rows + 4 <= 6 * word else 10  # packed board storage (tetris_hip_n_planes) for 10 columns
    known_read = B * (planes * word + 8)
    known_write = B * (8 + 1)
    for tag, rows, known in (("fetch", f, known_read), ("write", w, known_write)):
        ref = [v for k, v in rows if "refresh_kernel" in k]
        step = [v for k, v in rows if "step_kernel" in k]
        raw_ref = mean_tail(ref, 3) * 1024
        raw_step = mean_tail(step, 30) * 1024
        out[tag] = {"refresh_raw_bytes": raw_ref, "refresh_known_bytes": known, "correction": known / raw_ref,
                    "step_raw_bytes": raw_step, "step_corrected_bytes": raw_step * known / raw_ref}
    out["step_kernel_hbm_bytes_per_launch"] = out["fetch"]["step_corrected_bytes"] + out["write"]["step_corrected_bytes"]
    out["step_kernel_hbm_bytes_per_env_step"] = out["step_kernel_hbm_bytes_per_launch"] / B
    # key the profile to the kernel sources it measured: bench.py quotes it only while they are unchanged
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["csrc_hash"] = bench.csrc_hash()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
