#!/usr/bin/env python3
"""Time the step kernel for several TETRIS_STEP_BLOCKS_PER_CU values (persistent grid size)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris  # noqa: E402

rows = int(os.environ.get("ABL_ROWS", "20"))
B = int(os.environ.get("ABL_B", str(1 << 20)))
env = VecTetris(10, rows, B, device="cuda", auto_reset=True, seed=0)
for t in range(200):
    env.step()
snap = env.state_dict()
for k in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4", "5", "6", "8", "16"])]:
    os.environ["TETRIS_STEP_BLOCKS_PER_CU"] = str(k)
    best = 1e9
    for rep in range(3):
        env.load_state_dict(snap)
        evs = []
        for t in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            env.step()
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        best = min(best, sum(ts[2:-2]) / len(ts[2:-2]))
    print("blocks/CU=%2d  step kernel %.1f us" % (k, 1e3 * best))
