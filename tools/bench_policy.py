#!/usr/bin/env python3
"""Times the policy-side kernels on steady-state boards: tetris_hip_policy_greedy (get_best_policy /
fitness, game.py:102-120), tetris_hip_rollouts (perform_rollouts, game.py:129-160) and the fused greedy
step_many, and prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tetris_amd import VecTetris, _lib
if os.environ.get('TETRIS_VARIANT_LIB'):  # time a variant library instead
    import ctypes
    _lib._install_test_backend(_lib._Binding(ctypes.CDLL(os.environ['TETRIS_VARIANT_LIB'])))  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


rows = int(os.environ.get("ABL_ROWS", "20"))
B = 1 << 20
env = VecTetris(10, rows, B, device="cuda", auto_reset=True, seed=0)
for _ in range(150):
    env.step()
res = {"board": "10x%d" % rows, "envs": B}
ms = timeit(lambda: env.greedy_actions())
res["greedy_actions"] = {"ms": ms, "env_per_s": B / ms * 1e3}
ms = timeit(lambda: env.greedy_actions(include_fitness=True))
res["greedy_actions_with_fitness_matrix"] = {"ms": ms, "env_per_s": B / ms * 1e3}
K = 10
traj = [None]


def fused():
    traj[0] = env.step_many(K, policy="greedy", out=traj[0])


ms = timeit(fused, reps=3, warm=1)
res["step_many_greedy"] = {"steps_per_launch": K, "ms_per_step": ms / K, "env_steps_per_s": B * K / ms * 1e3}
del traj
small = VecTetris(10, rows, 65536, device="cuda", auto_reset=True, seed=1)
for _ in range(150):
    small.step()
pairs = int(small.n_valid.sum().item())
for pol in ("random", "greedy"):
    ms = timeit(lambda: small.rollouts(length=5, n=5, policy=pol), reps=3, warm=1)
    res["rollouts_%s" % pol] = {"envs": 65536, "env_action_pairs": pairs, "length": 5, "n": 5, "ms": ms,
                                "rollout_steps_per_s": pairs * 25 / ms * 1e3}
print(json.dumps(res))
