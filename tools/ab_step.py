#!/usr/bin/env python3
"""A/B of step-kernel variants in one process (interleaved): usage ab_step.py name=path.so ..."""
import ctypes, os, sys
import torch
ROOT = "/root/repo" if os.path.exists("/root/repo/tetris_amd") else os.getcwd()
sys.path.insert(0, ROOT)
from tetris_amd import _lib, VecTetris
libs = {}
for spec in sys.argv[1:]:
    n, pth = spec.split("=")
    libs[n] = _lib._Binding(ctypes.CDLL(pth))
rows = int(os.environ.get("ABL_ROWS", "20"))
for B in (1 << 20, 1 << 22, 1 << 16):
    base = VecTetris(10, rows, B, device="cuda", auto_reset=True, seed=0)
    for t in range(200):
        base.step()
    snap = base.state_dict()
    res = {n: [] for n in libs}
    for rep in range(5):
        for n, lib in libs.items():
            base.load_state_dict(snap)
            base._lib = lib
            base._step_call = None
            for t in range(5):
                base.step()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for t in range(50):
                base.step()
            e.record()
            torch.cuda.synchronize()
            res[n].append(s.elapsed_time(e) / 50 * 1e3)
    for n in libs:
        print("B=%8d %-12s %.2f us  (%s)" % (B, n, min(res[n]), " ".join("%.1f" % x for x in res[n])), flush=True)
    del base
