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
QUICK = os.environ.get("PROBE_QUICK") == "1"
side = torch.cuda.Stream()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for lds_bytes in (() if QUICK else (1024, 8192, 20480, 28672, 40960, 65536, 98304)):
    bad = []
    for rep in range(3):
        occ.occupy(ctypes.c_void_p(side.cuda_stream), 256, lds_bytes, 2_000_000, ctypes.c_void_p(sink.data_ptr()))
        torch.cuda._sleep(200_000)  # let the occupying workgroups start first
        ba, bv, fit = env.greedy_actions(include_fitness=True)
        torch.cuda.synchronize()
        bad.append(int((fit.view(torch.int32) != ref.view(torch.int32)).any(dim=1).sum()))
    print("occupier holds %6d B of LDS per compute unit: bad envs in 3 launches %s" % (lds_bytes, bad), flush=True)

# Two DISPATCHES of the failing kernel at one workgroup per compute unit each, on two streams at once:
# same code, same tables, different kernel arguments / scratch allocation per dispatch.
env2 = VecTetris(C, R, B, device="cuda", auto_reset=True, seed=4)
for t in range(24):
    env2.step()
torch.cuda.synchronize()
ref2 = env2.greedy_actions(include_fitness=True)[2].clone()
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad1 = bad2 = 0
for rep in range(30):
    with torch.cuda.stream(s1):
        f1 = env.greedy_actions(include_fitness=True)[2]
    with torch.cuda.stream(s2):
        f2 = env2.greedy_actions(include_fitness=True)[2]
    torch.cuda.synchronize()
    bad1 += int((f1.view(torch.int32) != ref.view(torch.int32)).any(dim=1).sum())
    bad2 += int((f2.view(torch.int32) != ref2.view(torch.int32)).any(dim=1).sum())
print("two dispatches of 65,536 envs on two streams, 30 rounds: bad envs %d + %d" % (bad1, bad2))
# the same two batches as ONE dispatch of 131,072 envs (two workgroups per compute unit)
env3 = VecTetris(C, R, 2 * B, device="cuda", auto_reset=True, seed=3)
cells = torch.cat([env.boards(), env2.boards()])
env3.set_boards(cells, piece=torch.cat([env.piece, env2.piece]).to(torch.int64))
bad3 = 0
for rep in range(5):
    f3 = env3.greedy_actions(include_fitness=True)[2]
    torch.cuda.synchronize()
    bad3 += int((f3.view(torch.int32) != torch.cat([ref, ref2]).view(torch.int32)).any(dim=1).sum())
print("the same boards as one dispatch of 131,072 envs, 5 rounds: bad envs %d" % bad3)

if QUICK:
    sys.exit(0)
# Which neighbour does it take?  The failing kernel (env2, stream 2) next to OTHER kernels on stream 1.
envb = VecTetris(C, 20, 1 << 18, device="cuda", auto_reset=True, seed=9)   # 32-bit boards: other instantiations of the same templates
for name, neighbour in (("get_after_states of the same geometry (other kernel, same tables)", lambda: env.get_after_states(include_terminal=True)),
                        ("rollouts of the same geometry", lambda: env.rollouts(length=2, n=1, policy="greedy")),
                        ("get_best_policy of x20 boards (same template, other instantiation)", lambda: envb.greedy_actions(include_fitness=True)),
                        ("eight steps of x20 boards", lambda: [envb.step() for _ in range(8)])):
    bad = 0
    for rep in range(20):
        with torch.cuda.stream(s1):
            neighbour()
        with torch.cuda.stream(s2):
            f2 = env2.greedy_actions(include_fitness=True)[2]
        torch.cuda.synchronize()
        bad += int((f2.view(torch.int32) != ref2.view(torch.int32)).any(dim=1).sum())
    print("next to %s, 20 rounds: bad envs %d" % (name, bad), flush=True)
