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
out = "/tmp/libtetris_abl_%d.so" % m
if not os.path.exists(out):
    src = os.path.join(ROOT, "tetris_amd", "csrc", "tetris_kernels.hip")
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-DTET_ABLATE=%d" % (m % 1000), "-DTET_STEP_WAVES=%d" % (m // 1000), src, "-o", out])
env = VecTetris(10, rows, 1 << 20, device="cuda", auto_reset=True, seed=0)
for t in range(150):
    env.step()
torch.cuda.synchronize()
env._lib = _lib._Binding(ctypes.CDLL(out))
for t in range(20):
    env.step()
torch.cuda.synchronize()
print("variant", m, "done")
