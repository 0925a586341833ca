#!/usr/bin/env python3
"""Run warm steady-state steps with an ablation variant of the library (see tools/ablate.py);
meant to sit under `rocprofv3 --pmc ...` to get per-variant instruction / stall counters."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import _lib, build, VecTetris  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rows = int(os.environ.get("ABL_ROWS", "20"))
out = os.path.join(ROOT, "build_variants", "libtetris_abl_%d.so" % m)  # built by tools/ablate.py (ABL_COMPILE_ONLY=1)
if not os.path.exists(out):
    sys.exit("build the variant first: ABL_COMPILE_ONLY=1 python tools/ablate.py %d" % m)
env = VecTetris(10, rows, 1 << 20, device="cuda", auto_reset=True, seed=0)
for t in range(150):
    env.step()
torch.cuda.synchronize()
env._lib = _lib._Binding(ctypes.CDLL(out))
for t in range(20):
    env.step()
torch.cuda.synchronize()
print("variant", m, "done")
