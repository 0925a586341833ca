#!/usr/bin/env python3
"""One-off long parity soak at BASELINE's full batch size (not part of the test suite: minutes of
oracle time): 1,048,576 envs in lock-step with the CPU oracle, every output of every step bit-exact,
boards every 16 steps.  Prints one line per configuration; output kept under profiles/."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TETRIS_SOAK_PROGRESS"] = "1"  # a line every 16 steps (long silent runs look hung to the GPU runner)
import parity_cases as pc  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.lib()
for name, kw in (("config 3: 10x20, default pieces, 256 steps", dict(steps=256, R=20, pieces="default")),
                 ("config 5: 10x40, default pieces, 128 steps", dict(steps=128, R=40, pieces="default")),
                 ("10x20, standard-7 pieces, 128 steps", dict(steps=128, R=20, pieces="standard7"))):
    t0 = time.perf_counter()
    eps = pc.cfg3_full_size_bit_exact("cuda", orc, B=1 << 20, board_every=16, **kw)
    print("%s: 1,048,576 envs bit-exact on obs / reward / done / lines / n_valid / piece / action every step and the "
          "boards every 16 steps; %d episodes finished; %.0f s" % (name, eps, time.perf_counter() - t0), flush=True)

for name, kw in (("10x20, default pieces", dict(R=20, pieces="default", steps=160, every=32)),
                 ("10x40, default pieces", dict(R=40, pieces="default", steps=192, every=64)),
                 ("10x20, standard-7 pieces", dict(R=20, pieces="standard7", steps=96, every=32)),
                 ("12x20, default pieces", dict(R=20, C=12, pieces="default", steps=128, every=64))):
    t0 = time.perf_counter()
    n = pc.afterstate_family_full_size("cuda", orc, B=1 << 20, **kw)
    print("afterstate family, %s: get_after_states (valid + include-terminal matrices) and get_best_policy of all "
          "1,048,576 envs bit-exact at %d points of steady-state play; %.0f s" % (name, n, time.perf_counter() - t0),
          flush=True)
