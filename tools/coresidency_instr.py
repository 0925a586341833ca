#!/usr/bin/env python3
"""Reads the wave-uniform state that an assembly-instrumented build of the failing kernel stored
(tools/asm_variant.py / DESIGN 3.2): per wave and loop iteration s65 (wu0), s66 (wu1), s36, s37, s20, s21, s24, s28."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris, _lib  # noqa: E402

_lib._install_test_backend(_lib._Binding(ctypes.CDLL(sys.argv[1])))
C, R, B = 12, 40, 65536
env = VecTetris(C, R, B, device="cuda", auto_reset=True, seed=3)
env2 = VecTetris(C, R, B, device="cuda", auto_reset=True, seed=4)
for t in range(24):
    env.step()
    env2.step()
torch.cuda.synchronize()
K = 16
big = torch.zeros(B * (K + 1), dtype=torch.float32, device="cuda")  # best_value [B] followed by K debug dwords per env
env2._best_action = torch.empty(B, dtype=torch.int32, device="cuda")
env2._best_value = big[:B]
env2._fitness_all = None
big1 = torch.zeros(B * (K + 1), dtype=torch.float32, device="cuda")  # the neighbour runs the same instrumented kernel
env._best_action = torch.empty(B, dtype=torch.int32, device="cuda")
env._best_value = big1[:B]
env._fitness_all = None
NAMES = os.environ.get("INSTR_NAMES", "d0 d1 full_lo full_hi valid_lo valid_hi wd0 wd1").split()


def words():
    return big[B:].view(torch.int32).cpu().numpy().reshape(B, K).copy()


ba, bv, fit = env2.greedy_actions(include_fitness=True)
torch.cuda.synchronize()
ref2 = fit.clone()
dbg_ref = words()
piece = env2.piece.cpu().numpy()
print("quiet run, env 0 and 1:", dict(zip(NAMES, dbg_ref[0][:len(NAMES)].tolist())), "|", dbg_ref[0][len(NAMES):].tolist(), "| piece", piece[0])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
shown = 0
tot_bad = tot_diff = tot_both = 0
for rep in range(30):
    with torch.cuda.stream(s1):
        env.greedy_actions(include_fitness=True)
    with torch.cuda.stream(s2):
        ba, bv, f2 = env2.greedy_actions(include_fitness=True)
    torch.cuda.synchronize()
    bad = (f2.view(torch.int32) != ref2.view(torch.int32)).any(dim=1).cpu().numpy()
    dbg = words()
    diff = (dbg != dbg_ref).any(axis=1)
    tot_bad += bad.sum(); tot_diff += diff.sum(); tot_both += (bad & diff).sum()
    for e in np.nonzero(diff)[0][:4]:
        if shown < 16:
            shown += 1
            cols = np.nonzero(dbg[e] != dbg_ref[e])[0].tolist()
            print("   env %6d (wave %4d lane %2d piece %d) wrong fitness=%d; words that differ from the quiet run: %s"
                  % (e, e // 64, e % 64, piece[e], bad[e], [(k, hex(int(dbg[e][k]) & 0xffffffff), hex(int(dbg_ref[e][k]) & 0xffffffff)) for k in cols]))
    if rep == 29:
        dcols = (dbg != dbg_ref)[diff]
        print("last round: which words differ, over %d envs:" % diff.sum(), dict(zip(NAMES, dcols.sum(axis=0)[:len(NAMES)].tolist())))
print("30 rounds: envs with wrong fitness %d, envs whose dumped words differ %d, both %d" % (tot_bad, tot_diff, tot_both))
print("done")
