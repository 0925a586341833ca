#!/usr/bin/env python3
"""Repeat / shard consistency of the afterstate family for every board width and both word sizes on a
batch where several workgroups share a compute unit (tetris_amd.selftest.afterstate_family_consistency), N runs.
   stress_consistency.py [path-of-another-libtetris_hip.so]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import _lib, selftest  # noqa: E402

if len(sys.argv) > 1:
    _lib._install_test_backend(_lib._Binding(ctypes.CDLL(sys.argv[1])))
    print("library:", sys.argv[1])
print("source hash of the library:", _lib.load().source_hash())
runs = int(os.environ.get("STRESS_RUNS", "1"))
configs = [(C, R, pieces) for C in [int(x) for x in os.environ.get("STRESS_C", "5 6 7 8 9 10 11 12").split()]
           for R, pieces in ((20, "default"), (20, "standard7"), (40, "default"), (24, "default"), (50, "default"))]
fails = {c: [] for c in configs}
t0 = time.perf_counter()
for run in range(runs):
    for cfg in configs:
        C, R, pieces = cfg
        try:
            selftest.afterstate_family_consistency("cuda", C=C, R=R, pieces=pieces, B=int(os.environ.get("STRESS_B", str(3 << 16))))
        except AssertionError as exc:
            fails[cfg].append(str(exc).split(" (rep")[0])
    print("run %d done (%.0f s)" % (run, time.perf_counter() - t0), flush=True)
bad = 0
for cfg in configs:
    C, R, pieces = cfg
    f = fails[cfg]
    bad += bool(f)
    print("%2d x %2d %-9s %s envs: %s" % (C, R, pieces, os.environ.get("STRESS_B", "196,608"), "ok in %d runs" % runs if not f else
                                                "FAILED in %d of %d runs: %s" % (len(f), runs, sorted(set(f)))), flush=True)
print("configurations that failed:", bad)
sys.exit(1 if bad else 0)
