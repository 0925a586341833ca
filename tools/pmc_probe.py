#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few launches with KNOWN byte counts
(reset / refresh, for calibrating FETCH_SIZE / WRITE_SIZE on this access pattern:
one dword or qword per lane, plane-major) followed by warm steady-state steps.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_probe.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/pmc_probe.py
  python tools/parse_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_amd import VecTetris  # noqa: E402

rows = int(os.environ.get("PROBE_ROWS", "20"))
B = 1 << 20
env = VecTetris(10, rows, B, device="cuda", auto_reset=True, seed=0)
for t in range(150):  # reach the steady-state height distribution
    env.step()
torch.cuda.synchronize()
for _ in range(3):
    env.refresh()     # reads 10 planes + meta, writes meta + n_valid
torch.cuda.synchronize()
for t in range(30):
    env.step()
torch.cuda.synchronize()
env.check()
print("probe done", env.stats())
