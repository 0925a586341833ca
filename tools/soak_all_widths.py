#!/usr/bin/env python3
"""One-off soak: every board width 5..12 on 32- and 64-bit boards, 262,144 envs in lock-step with the oracle
(every output of every step, boards every step, afterstate matrices every 8 steps); output kept under profiles/."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.lib()
for C in range(5, 13):
    for R, pieces in ((20, "default"), (40, "standard7"), (24, "standard7"), (50, "default")):
        t0 = time.perf_counter()
        eps = pc.lockstep("cuda", orc, C, R, 1 << 18, pieces, steps=24, seed=21, check_after_every=8)
        print("%2d x %2d %-9s 262,144 envs x 24 steps bit-exact (obs / reward / done / lines / n_valid / piece / action / boards every step, "
              "afterstate matrices every 8); %d episodes finished; %.0f s" % (C, R, pieces, eps, time.perf_counter() - t0), flush=True)
print("done")
