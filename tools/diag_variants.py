#!/usr/bin/env python3
"""Diagnostic: the 12-column greedy kernel built several ways, same boards, bad env counts."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from tetris_amd import VecTetris, _lib  # noqa: E402

orc.lib()
C, B = int(os.environ.get("DIAG_C", "12")), int(os.environ.get("DIAG_B", str(1 << 17)))
R = int(os.environ.get("DIAG_R", "20"))
w32 = np.array(VecTetris.BCTS_WEIGHTS, np.float32)
ref = orc.OracleVecEnv(C, R, B, auto_reset=True, seed=3, nthreads=0)
for t in range(int(os.environ.get("DIAG_T", "8"))):
    ref.step()
rf, rnv, rfa, rna = ref.afterstates(include_terminal=True)
want = rfa[..., 0] * w32[0]
for q in range(1, 8):
    want = (want + rfa[..., q] * w32[q]).astype(np.float32)
A = rfa.shape[1]
k = np.arange(A)[None, :]
for name in sys.argv[1:]:
    if name == "default":
        _lib._install_test_backend(None)
    else:
        _lib._install_test_backend(_lib._Binding(ctypes.CDLL(os.path.join(ROOT, "build_variants", name))))
    env = VecTetris(C, R, B, device="cuda", auto_reset=True, seed=3)
    for t in range(int(os.environ.get("DIAG_T", "8"))):
        env.step()
    torch.cuda.synchronize()
    counts = []
    for rep in range(4):
        ba, bv, fit = env.greedy_actions(include_fitness=True)
        torch.cuda.synchronize()
        got = fit.cpu().numpy()[:, :A]
        bad = (got != want) & (k < rna[:, None])
        counts.append(int(bad.any(axis=1).sum()))
        be = bad.any(axis=1)
        if be.any():
            wantv = np.where(k < rnv[:, None], rf[..., 0] * w32[0], 0)
            wv = rf[..., 0] * w32[0]
            for q in range(1, 8):
                wv = (wv + rf[..., q] * w32[q]).astype(np.float32)
            wv = np.where(k < rnv[:, None], wv, -np.inf)
            best = np.where(rnv > 0, wv.argmax(axis=1), -1)
            ba_n = ba.cpu().numpy()
            e = np.nonzero(be)[0]
            waves = np.unique(e // 64)
            first = np.array([np.nonzero(bad[i])[0][0] for i in e])
            shift_ok = sum(bool((got[i, first[j]:rna[i] - 1] == want[i, first[j] + 1:rna[i]]).all()) for j, i in enumerate(e))
            p1 = np.array([(ref.piece[64 * w:64 * w + 64] == 1).sum() for w in waves])
            b1 = np.array([be[64 * w:64 * w + 64].sum() for w in waves])
            print("   rep", rep, "bad waves", len(waves), "first bad row hist", np.bincount(first).tolist(),
                  "| pure one-row shift in", shift_ok, "of", len(e), "| wrong best_action among bad envs:",
                  int((ba_n[e] != best[e]).sum()), "(all envs:", int((ba_n != best).sum()), ")")
            print("   piece-1 lanes per bad wave", p1.tolist()[:20], "bad lanes per bad wave", b1.tolist()[:20])
            print("   bad waves", waves.tolist()[:40])
            for i in e[:6]:
                bi = np.nonzero(bad[i])[0]
                print("     env", i, "piece", ref.piece[i], "na", rna[i], "bad rows", bi.tolist()[:12])
                for r in bi[:4]:
                    where = np.nonzero(want[i, :rna[i]] == got[i, r])[0].tolist()
                    print("        row", r, "got", got[i, r], "want", want[i, r], "| got equals want of rows", where)
            # hypothesis H1: placement 12 is dropped entirely (never evaluated, never counted)
            fb = int(np.bincount(first).argmax())
            print("     most common first bad row:", fb, "; pieces of bad envs", np.bincount(ref.piece[e]).tolist())
            h1 = np.delete(wv, fb, axis=1)
            h1_best = np.where(rnv > 0, h1.argmax(axis=1), -1)
            print("     best_action == H1 prediction (most common first bad row dropped) in", int((ba_n[e] == h1_best[e]).sum()), "of", len(e),
                  "| == true best in", int((ba_n[e] == best[e]).sum()))
            if "rows" in name:
                dw = bv.cpu().numpy().view(np.uint32)
                wl0, wl1 = dw & 255, (dw >> 8) & 255
                isb = np.zeros(B // 64, bool)
                isb[waves] = True
                w0 = wl0[::64]; w1 = wl1[::64]
                print("     (wu0 | wu1 << 4) of loop 0: bad waves", np.unique(w0[isb], return_counts=True), "| other waves", np.unique(w0[~isb], return_counts=True))
                print("     (wu0 | wu1 << 4) of loop 1: bad waves", np.unique(w1[isb], return_counts=True), "| other waves", np.unique(w1[~isb], return_counts=True))
                print("     lanes of a wave agree on the words:", bool((dw.reshape(-1, 64)[:, :1] & 0xFFFF == dw.reshape(-1, 64) & 0xFFFF).all()))
            if "hwid" in name:
                hw = bv.cpu().numpy().view(np.uint32)[::64]
                fld = lambda lo, n: (hw >> lo) & ((1 << n) - 1)
                isb = np.zeros(len(hw), bool)
                isb[waves] = True
                for nm, lo, n in (("wave_id", 0, 4), ("simd", 4, 2), ("pipe", 6, 2), ("cu", 8, 4), ("sh", 12, 1), ("se", 13, 3),
                                  ("tg_id", 16, 4), ("vm", 20, 4), ("queue", 24, 3)):
                    print("     %-8s bad waves:" % nm, np.bincount(fld(lo, n)[isb], minlength=1 << n).tolist(),
                          "| all waves:", np.bincount(fld(lo, n), minlength=1 << n).tolist())
    f, nv, fa, na = env.get_after_states(include_terminal=True)
    m_bad = int((fa.cpu().numpy()[:, :A] != rfa).any(axis=(1, 2)).sum())
    print(name, "greedy bad envs over 4 calls:", counts, "| matrix bad envs:", m_bad, flush=True)
print("done")
