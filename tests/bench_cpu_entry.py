#!/usr/bin/env python3
"""TEST ONLY: bench.py's host code and rank launcher on a machine without a GPU.

Binds tetris_amd to the g++ harness build of the per-lane source (tests/harness) and runs bench.main()
on CPU tensors; started with `--gpus N` it is the parent and every rank it starts is this script again.
The product benchmark (bench.py) has no such switch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import bench  # noqa: E402

if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ or "--gpus" not in sys.argv or sys.argv[sys.argv.index("--gpus") + 1] == "1":
        import harness_backend
        from tetris_amd import _lib
        _lib._install_test_backend(harness_backend.binding())
    if os.environ.get("TETRIS_TEST_FAIL_RANK") == os.environ.get("RANK", "-"):
        sys.exit(3)  # launcher test: one rank dies before the rendezvous
    sys.exit(bench.main(device="cpu", entry=os.path.abspath(__file__), prebuild=False))
