#!/usr/bin/env python3
"""Per-workgroup timeline of ONE step-kernel launch (diagnostic build -DTET_STAMPS=1).

Builds build_variants/libtetris_stamps.so (here or on the box), steps a 1 Mi-env batch to steady
state, reads the stamps of the last launch and prints: span of the launch, per-phase durations
(start -> loads landed -> compute done -> stores issued+drained), workgroups per CU, and how many
workgroups are in their compute phase over time (the VALU's supply of work).

  TIMELINE_COMPILE_ONLY=1 python tools/timeline.py     # build the variant where hipcc is
  python tools/timeline.py [envs] [rows]               # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "build_variants")
SO = os.path.join(VDIR, "libtetris_stamps.so")
SRC = os.path.join(ROOT, "tetris_amd", "csrc", "tetris_kernels.hip")
extra = os.environ.get("TIMELINE_FLAGS", "").split()
if not os.path.exists(SO) or os.environ.get("TIMELINE_COMPILE_ONLY") == "1":
    os.makedirs(VDIR, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-shared", "-fPIC", "-DTET_STAMPS=1"] + extra + [SRC, "-o", SO])
if os.environ.get("TIMELINE_COMPILE_ONLY") == "1":
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from tetris_amd import VecTetris, _lib  # noqa: E402

cdll = ctypes.CDLL(SO)
_lib._install_test_backend(_lib._Binding(cdll))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
env = VecTetris(10, rows, B, device="cuda", auto_reset=True, seed=0)
for _ in range(150):
    env.step()
torch.cuda.synchronize()
blk = 512 if rows + 4 <= 31 else 256
n_wg = (B + blk - 1) // blk
for rep in range(3):
    for _ in range(5):
        env.step()
    torch.cuda.synchronize()
    buf = np.zeros((n_wg, 20), np.uint64)
    rc = cdll.tetris_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), n_wg)
    assert rc == 0, rc
    # the persistent kernel launches fewer workgroups than tiles: keep the slots this launch wrote
    newest = buf[:, 0].max()
    buf = buf[(buf[:, 0] > 0) & (newest - buf[:, 0] < 100000) & (buf[:, 3] >= buf[:, 0])]
    n_wg = len(buf)
    per_tile = buf[:, 6:18].astype(np.int64)
    t = buf[:, :4].astype(np.int64)
    t0 = t[:, 0].min()
    t = (t - t0) * 0.01  # us (100 MHz)
    hw, xcc = buf[:, 4].astype(np.int64), buf[:, 5].astype(np.int64) & 15
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)  # xcc, se, sh, cu
    ncu = len(np.unique(cu))
    per_cu = np.bincount(np.unique(cu, return_inverse=True)[1])
    print("---- rep %d: %d workgroups (block %d) on %d CUs (min/median/max per CU %d/%d/%d)" % (
        rep, n_wg, blk, ncu, per_cu.min(), int(np.median(per_cu)), per_cu.max()))
    print("launch span %.2f us (first start -> last end); last start at %.2f us" % (t[:, 3].max(), t[:, 0].max()))
    for name, d in (("start -> loads landed + tables in LDS", t[:, 1] - t[:, 0]),
                    ("compute", t[:, 2] - t[:, 1]),
                    ("stores issued + drained", t[:, 3] - t[:, 2]), ("whole workgroup", t[:, 3] - t[:, 0])):
        print("  %-36s mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us" % (
            name, d.mean(), *np.percentile(d, [10, 50, 90]), d.max()))
    # workgroups in each phase over time (whole chip), 1 us bins
    T = min(int(np.ceil(t[:, 3].max())) + 1, 400)
    grid = np.arange(T) + 0.5
    rows_ = []
    for a, b in ((0, 1), (1, 2), (2, 3)):
        rows_.append([int(((t[:, a] <= x) & (t[:, b] > x)).sum()) for x in grid])
    print("  t(us)   loading computing storing   (workgroups chip-wide; %d CUs x 3 resident = %d slots)" % (ncu, 3 * ncu))
    for k in range(T):
        print("  %5.1f  %7d %9d %7d" % (grid[k], rows_[0][k], rows_[1][k], rows_[2][k]))
    # per-tile stamps of wave 0 of every workgroup: compute done / wait passed / stores issued
    pt = np.where(per_tile > 0, (per_tile - t0) * 0.01, np.nan)
    for k in range(4):
        c, w_, st_ = pt[:, 3 * k], pt[:, 3 * k + 1], pt[:, 3 * k + 2]
        ok = ~np.isnan(c) & (c >= 0) & (c < 1000)
        if ok.sum() == 0:
            continue
        print("  tile %d (%4d wgs): compute done at p10 %.2f p50 %.2f p90 %.2f | wait before stores p50 %.2f max %.2f | "
              "store issue p50 %.2f max %.2f us" % (k, ok.sum(), *np.percentile(c[ok], [10, 50, 90]),
                                                    np.median((w_ - c)[ok]), (w_ - c)[ok].max(),
                                                    np.median((st_ - w_)[ok]), (st_ - w_)[ok].max()))
    # per-CU finish time spread
    fin = np.array([t[cu == c, 3].max() for c in np.unique(cu)])
    print("  per-CU finish: min %.2f  p50 %.2f  max %.2f us" % (fin.min(), np.median(fin), fin.max()))
