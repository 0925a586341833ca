#!/usr/bin/env python3
"""Timing-only ablation of the step kernel: builds libtetris_hip variants with
-DTET_ABLATE=<mask> (parts of the step removed -> wrong results, same memory
traffic) and times the step kernel of each with HIP events, interleaved A/B in
one process.  bit0 features, bit1 next-piece mask, bit3 clear, bit4 hole tables, bit5 wells, bit6 afterstates stores, bit7 table staging."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import _lib, build  # noqa: E402
from tetris_amd import VecTetris  # noqa: E402

masks = [int(x) for x in (sys.argv[1:] or ["0", "1", "2", "4", "8", "3", "15"])]
extra = os.environ.get("ABL_FLAGS", "").split()
per_mask_flags = {}
for spec in os.environ.get("ABL_VARIANTS", "").split(";"):
    # e.g. ABL_VARIANTS="100:-DTET_AFTER_WAVES=3;101:-DTET_SCRATCH_OR=0" (build variants keyed by pseudo-mask)
    if ":" in spec:
        k, v = spec.split(":", 1)
        per_mask_flags[int(k)] = v.split()
# a mask >= 1000 means: TET_STEP_WAVES = mask // 1000, TET_ABLATE = mask % 1000
rows = int(os.environ.get("ABL_ROWS", "20"))
pieces = os.environ.get("ABL_PIECES", "default")
src = os.path.join(ROOT, "tetris_amd", "csrc", "tetris_kernels.hip")
# variants are built into build_variants/ (git-ignored, travels with gpurun) so they can be
# compiled here, in parallel, instead of on GPU-box minutes: ABL_COMPILE_ONLY=1 stops after the build
import glob  # noqa: E402
newest_src = max(os.path.getmtime(f) for f in glob.glob(os.path.join(os.path.dirname(src), "*.h*")) +
                 glob.glob(os.path.join(os.path.dirname(src), "*.inc")))
vdir = os.path.join(ROOT, "build_variants")
os.makedirs(vdir, exist_ok=True)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
todo = []
for m in masks:
    out = os.path.join(vdir, "libtetris_abl_%d.so" % m)
    if os.path.exists(out) and os.environ.get("ABL_REBUILD") != "1" and os.path.getmtime(out) > newest_src:
        continue
    flags = (["-DTET_ABLATE=%d" % (m % 1000), "-DTET_STEP_WAVES=%d" % (m // 1000)] if m not in per_mask_flags else []) + \
        extra + per_mask_flags.get(m, [])
    todo.append((out, flags))
# (through the product's assembly pipeline: tetris_amd.build.build_variant, incl. the 64-bit-shift hazard patch)
with ThreadPoolExecutor(max_workers=max(1, min(8, len(todo) or 1))) as pool:
    list(pool.map(lambda t: build.build_variant(t[0], t[1]), todo))
if os.environ.get("ABL_COMPILE_ONLY") == "1":
    sys.exit(0)
libs = {m: _lib._Binding(ctypes.CDLL(os.path.join(vdir, "libtetris_abl_%d.so" % m))) for m in masks}

B = 1 << 20
base = VecTetris(10, rows, B, device="cuda", pieces=pieces, auto_reset=True, seed=0)
for t in range(200):
    base.step(base.random_actions())
snap = base.state_dict()
res = {m: [] for m in masks}
for rep in range(5):
    for m in masks:
        base.load_state_dict(snap)
        base._lib = libs[m]
        base._step_call = None  # the bound step call belongs to the library that prepared it
        acts = None
        evs = []
        for t in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            base.step(acts)
            e.record()
            evs.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in evs)
        res[m].append(sum(ts[2:-2]) / len(ts[2:-2]))
for m in masks:
    print("ablate=%2d  step kernel %.1f us  (runs: %s)" % (m, 1e3 * min(res[m]), " ".join("%.1f" % (1e3 * x) for x in res[m])))
