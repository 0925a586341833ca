#!/usr/bin/env python3
"""Is the co-residency fault about WHERE a workgroup sits (LDS base, wave slot) or about two workgroups of
the same kernel interacting?  65,536 envs (one workgroup per compute unit: never fails alone) are evaluated
while a do-nothing kernel on another stream holds part of every compute unit's LDS and wave slots.
   coresidency_probe.py <library that shows the fault> [columns rows]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris, _lib  # noqa: E402

_lib._install_test_backend(_lib._Binding(ctypes.CDLL(sys.argv[1])))
C, R = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 40)
occ = ctypes.CDLL(os.path.join(ROOT, "build_variants", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
B = 65536
env = VecTetris(C, R, B, device="cuda", auto_reset=True, seed=3)
for t in range(24):
    env.step()
torch.cuda.synchronize()
ba, bv, fit = env.greedy_actions(include_fitness=True)
torch.cuda.synchronize()
ref = fit.clone()
for rep in range(3):
    ba, bv, fit = env.greedy_actions(include_fitness=True)
    torch.cuda.synchronize()
    assert torch.equal(fit.view(torch.int32), ref.view(torch.int32)), "not even stable alone"
print("alone, one workgroup per compute unit: stable over 4 launches")
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for lds_bytes in (1024, 8192, 20480, 28672, 40960, 65536, 98304):
    bad = []
    for rep in range(3):
        occ.occupy(ctypes.c_void_p(side.cuda_stream), 256, lds_bytes, 2_000_000, ctypes.c_void_p(sink.data_ptr()))
        torch.cuda._sleep(200_000)  # let the occupying workgroups start first
        ba, bv, fit = env.greedy_actions(include_fitness=True)
        torch.cuda.synchronize()
        bad.append(int((fit.view(torch.int32) != ref.view(torch.int32)).any(dim=1).sum()))
    print("occupier holds %6d B of LDS per compute unit: bad envs in 3 launches %s" % (lds_bytes, bad), flush=True)
print("done")
