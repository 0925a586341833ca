#!/usr/bin/env python3
"""Times tetris_hip_afterstates (game.py:67-80 batched) on steady-state boards and prices it
against HBM: algorithmic bytes per env = C*W (board) + 8 (meta) + a_max*32 (feature rows) + 1."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris, _lib, build  # noqa: E402
import ctypes, subprocess  # noqa: E402

rows = int(os.environ.get("ABL_ROWS", "20"))
pieces = os.environ.get("ABL_PIECES", "default")
B = int(os.environ.get("ABL_B", str(1 << 20)))
layout = os.environ.get("ABL_LAYOUT", "env_major")
env = VecTetris(10, rows, B, device="cuda", pieces=pieces, auto_reset=True, seed=0, afterstate_layout=layout)
for t in range(150):
    env.step()
abl = int(os.environ.get("ABL_MASK", "0"))
if abl:  # variant built by tools/ablate.py (ABL_COMPILE_ONLY=1 python tools/ablate.py <mask>)
    env._lib = _lib._Binding(ctypes.CDLL(os.path.join(ROOT, "build_variants", "libtetris_abl_%d.so" % abl)))
if os.environ.get("TETRIS_VARIANT_LIB"):  # any variant library (full C-ABI)
    env._lib = _lib._Binding(ctypes.CDLL(os.environ["TETRIS_VARIANT_LIB"]))
res = {}
for inc in (False, True):
    for _ in range(3):
        env.get_after_states(include_terminal=inc)
    evs = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        env.get_after_states(include_terminal=inc)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    ms = sum(ts[1:-1]) / len(ts[1:-1])
    nbytes = 10 * env.desc.word_bytes + 8 + env.a_max * 32 * (2 if inc else 1) + (2 if inc else 1)
    res["include_terminal=%s" % inc] = dict(ms=ms, env_per_s=B / ms * 1e3, bytes_per_env=nbytes,
                                            GBps=nbytes * B / ms / 1e6, frac_of_8TBps=nbytes * B / ms / 1e6 / 8000)
print(json.dumps(dict(layout=layout, board="10x%d" % rows, pieces=pieces, envs=B, a_max=env.a_max, **res)))
