#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the afterstate kernels: steady-state boards, then a few launches
of tetris_hip_afterstates and tetris_hip_policy_greedy (PROBE_ROWS = 20 / 40)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_amd import VecTetris  # noqa: E402

rows = int(os.environ.get("PROBE_ROWS", "20"))
env = VecTetris(10, rows, 1 << 20, device="cuda", auto_reset=True, seed=0)
for t in range(150):
    env.step()
torch.cuda.synchronize()
for _ in range(6):
    env.get_after_states()
    env.greedy_actions()
torch.cuda.synchronize()
print("probe done")
