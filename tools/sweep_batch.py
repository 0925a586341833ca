#!/usr/bin/env python3
"""Step-kernel time vs batch size (how the kernel scales with the number of wave rounds)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris, _lib  # noqa: E402

# SWEEP_LIB=<path of a variant built by tools/ablate.py>: time that library's kernels instead
if os.environ.get("SWEEP_LIB"):  # the whole run (allocation included) goes through the variant library
    import ctypes
    _lib._install_test_backend(_lib._Binding(ctypes.CDLL(os.environ["SWEEP_LIB"])))

for B in [int(x) for x in (sys.argv[1:] or ["65536", "163840", "327680", "655360", "1048576", "1310720", "2097152",
                                            "4194304"])]:
    env = VecTetris(10, int(os.environ.get("SWEEP_ROWS", "20")), B, device="cuda", auto_reset=True, seed=0,
                    pieces=os.environ.get("SWEEP_PIECES", "default"))
    for t in range(120):
        env.step()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for t in range(40):
            env.step()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 40)
    print("B=%8d (%5.2f waves per SIMD)  %.1f us per step  %.2f G env-steps/s" % (B, B / 64 / 1024, best * 1e3,
                                                                                  B / best / 1e6))
    del env
