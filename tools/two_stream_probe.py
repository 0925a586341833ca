#!/usr/bin/env python3
"""Probe: does stepping two half-batches on two HIP streams (the tail of one launch overlapping the ramp
of the other) beat one launch over the whole batch?  Prints env-steps/s for both arrangements."""
import sys
import time

import torch

sys.path.insert(0, ".")
from tetris_amd import VecTetris  # noqa: E402

B = 1 << 20
steps = 600


def run_single():
    env = VecTetris(10, 20, B, device="cuda", auto_reset=True, seed=0)
    for _ in range(150):
        env.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


def run_split(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    envs = []
    for k in range(n):
        with torch.cuda.stream(streams[k]):
            envs.append(VecTetris(10, 20, B // n, device="cuda", auto_reset=True, seed=0, env_offset=k * (B // n)))
    for _ in range(150):
        for k in range(n):
            with torch.cuda.stream(streams[k]):
                envs[k].step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for k in range(n):
            with torch.cuda.stream(streams[k]):
                envs[k].step()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


print("one launch per step      : %.2f G env-steps/s" % (run_single() / 1e9))
for n in (2, 4):
    print("%d shards on %d streams    : %.2f G env-steps/s" % (n, n, run_split(n) / 1e9))
