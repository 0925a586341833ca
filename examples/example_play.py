#!/usr/bin/env python3
"""BASELINE config 1: one 10x20 board, random-action rollout through the facade.

The role of the reference's example_play.py (which crashes upstream because it iterates the
(features, None) tuple of get_after_states): unpack the tuple, pick a random valid
action, step, reset on done."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetris_amd import Tetris  # noqa: E402

import time  # noqa: E402

np.random.seed(0)
random.seed(0)
env = Tetris(10, 20)
total, episodes = 0, 0
t0 = time.perf_counter()
for t in range(200):
    features, _ = env.get_after_states()
    action = random.randrange(len(features))
    obs, reward, done, lines = env.step(action)
    total += lines
    if done:
        episodes += 1
        env.reset()
dt = time.perf_counter() - t0
env.render()
print("200 steps, %d episodes finished, %d lines cleared" % (episodes, total))
# BASELINE config 1 through the reference-shaped facade (a batch of ONE env: every call is a kernel
# launch plus device -> host copies); the reference itself: ~181 steps/s on one CPU core
print("%.0f steps/s through tetris_amd.Tetris (get_after_states + step + reset on done)" % (200 / dt))
