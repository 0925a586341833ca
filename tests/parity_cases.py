"""Shared parity cases.  Each takes the torch device to run tetris_amd on:
"cuda" for the parity tests proper (tests/test_gpu_parity.py, HIP kernels through
the C-ABI) or "cpu" with the harness backend (tests/test_host_logic.py, the same
lane logic compiled by g++ -- covers the host code and catches logic errors
before a GPU run)."""
import os

import numpy as np
import pytest
import torch

STANDARD7 = ["Straight", "RCorner", "LCorner", "Square", "SnakeR", "SnakeL", "T"]


def lockstep(device, orc, C, R, B, pieces, steps, seed, auto_reset=True, env_offset=0, check_after_every=0):
    from tetris_amd import VecTetris
    env = VecTetris(C, R, B, device=device, pieces=pieces, auto_reset=auto_reset, seed=seed, env_offset=env_offset)
    ref = orc.OracleVecEnv(C, R, B, pieces=pieces, auto_reset=auto_reset, seed=seed, env_offset=env_offset,
                           nthreads=0)
    np.testing.assert_array_equal(env.piece.cpu().numpy(), ref.piece)
    np.testing.assert_array_equal(env.n_valid.cpu().numpy(), ref.n_valid)
    episodes = 0
    for t in range(steps):
        if check_after_every and t % check_after_every == 0:
            f, nv, fa, na = env.get_after_states(include_terminal=True)
            rf, rnv, rfa, rna = ref.afterstates(include_terminal=True)
            np.testing.assert_array_equal(nv.cpu().numpy(), rnv)
            np.testing.assert_array_equal(na.cpu().numpy(), rna)
            np.testing.assert_array_equal(f.cpu().numpy(), rf)
            np.testing.assert_array_equal(fa.cpu().numpy(), rfa)
        if t % 2 == 0:  # explicit actions (from the policy kernel) ...
            a = env.random_actions().clone()
            obs, rew, done, lines = env.step(a)
            o_obs, o_rew, o_done, o_lines, n_bad = ref.step(a.cpu().numpy())
        else:           # ... and the policy fused into the step kernel
            obs, rew, done, lines = env.step()
            o_obs, o_rew, o_done, o_lines, n_bad = ref.step()
            np.testing.assert_array_equal(env.action.cpu().numpy(), ref.action)
        assert n_bad == 0
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs, err_msg="obs t=%d" % t)  # bit-exact float32
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew)
        np.testing.assert_array_equal(done.cpu().numpy(), o_done.astype(bool))
        np.testing.assert_array_equal(lines.cpu().numpy(), o_lines)
        np.testing.assert_array_equal(env.n_valid.cpu().numpy(), ref.n_valid)
        np.testing.assert_array_equal(env.piece.cpu().numpy(), ref.piece)
        np.testing.assert_array_equal(env.boards().cpu().numpy(), ref.cells)
        episodes += int(o_done.sum())
    st = env.stats()
    assert st["invalid"] == 0 and st["episodes"] == episodes and st["steps"] == steps * B
    return episodes


def lockstep_small(device, orc, C, R, pieces, B=1024 + 37, steps=150):
    episodes = lockstep(device, orc, C, R, B, pieces, steps=steps, seed=11, check_after_every=25)
    assert episodes > 0


def cfg2_bit_exact(device, orc, B=65536, steps=512):
    """BASELINE config 2: 65,536 envs, 10x20, random actions, every output every step."""
    episodes = lockstep(device, orc, 10, 20, B, "default", steps=steps, seed=0)
    assert episodes > B // 64


def no_auto_reset_and_invalid_actions(device, orc, B=4096):
    from tetris_amd import VecTetris
    env = VecTetris(10, 20, B, device=device, seed=5)
    ref = orc.OracleVecEnv(10, 20, B, seed=5)
    for t in range(80):
        a = env.random_actions().clone()
        # finished envs have n_valid == 0: any action is out of range there
        obs, rew, done, lines = env.step(a)
        ref.step(a.cpu().numpy())
        live = ref.invalid == 0
        np.testing.assert_array_equal(env.boards().cpu().numpy(), ref.cells)
        np.testing.assert_array_equal(obs.cpu().numpy()[live], ref.obs[live])
        np.testing.assert_array_equal(rew.cpu().numpy()[live], ref.reward[live])
        np.testing.assert_array_equal(env.n_valid.cpu().numpy(), ref.n_valid)
    assert env.stats()["invalid"] > 0
    with pytest.raises(IndexError):
        env.check()
    # host-driven reset of the finished envs only; the bag survives (game.py:50)
    dead = env.n_valid == 0
    assert dead.any()
    env.reset(mask=dead)
    cells = env.boards()
    assert not cells[dead].any() and (env.n_valid[dead] > 0).all()


def n_golden_seeds(g):
    n = 0
    while ("s%d_action" % n) in g:
        n += 1
    return n


def golden_trajectories_replay(device, orc, golden_dir):
    """Recorded game.Tetris runs (live reference, NumPy MT19937 piece stream) replayed
    through the HIP kernels: all seeds of one config form one batch."""
    from tetris_amd import VecTetris
    for tag, R in (("default", 20), ("default", 40), ("standard7", 20), ("standard7", 40)):
        g = np.load(os.path.join(golden_dir, "g2_traj_%s_10x%d.npz" % (tag, R)))
        seeds = range(n_golden_seeds(g))  # all 32 recorded games of the config as ONE batch
        assert len(seeds) == 32
        T = len(g["s0_action"])
        pieces = "default" if tag == "default" else STANDARD7
        n_pieces = len(pieces) if tag != "default" else 2
        streams = []
        for s in seeds:
            rng = orc.NumpyLegacyRNG(s)
            bag = orc.BagSampler(rng, n_pieces)
            n_draws = 1 + T + int(g["s%d_done" % s].sum())
            streams.append([bag.next() for _ in range(n_draws)] + [0] * (2 * T))
        L = min(len(x) for x in streams)
        stream = np.array([x[:L] for x in streams], np.uint8).T.copy()  # [L, B]
        env = VecTetris(10, R, len(seeds), device=device, pieces=pieces, auto_reset=True, piece_stream=stream)
        for t in range(T):
            act = np.array([g["s%d_action" % s][t] for s in seeds], np.int32)
            np.testing.assert_array_equal(env.piece.cpu().numpy(), [g["s%d_piece" % s][t] for s in seeds])
            np.testing.assert_array_equal(env.n_valid.cpu().numpy(), [g["s%d_n_valid" % s][t] for s in seeds])
            if t < 40:
                f, nv, fa, na = env.get_after_states(include_terminal=True)
                for i, s in enumerate(seeds):
                    if ("s%d_after_valid" % s) in g:
                        np.testing.assert_array_equal(f[i].cpu().numpy(), g["s%d_after_valid" % s][t][:env.a_max])
                        np.testing.assert_array_equal(fa[i].cpu().numpy(), g["s%d_after_all" % s][t][:env.a_max])
            obs, rew, done, lines = env.step(torch.from_numpy(act))
            obs, rew, done, lines = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), lines.cpu().numpy()
            cols = env.columns().cpu().numpy().astype(np.uint64).T  # [B, C]
            for i, s in enumerate(seeds):
                np.testing.assert_array_equal(obs[i], g["s%d_obs" % s][t])
                assert int(rew[i]) == g["s%d_reward" % s][t] and bool(done[i]) == bool(g["s%d_done" % s][t])
                assert int(lines[i]) == g["s%d_lines" % s][t]
                if not g["s%d_done" % s][t]:
                    np.testing.assert_array_equal(cols[i] & np.uint64((1 << (R + 4)) - 1), g["s%d_cols" % s][t])
        assert env.stats()["invalid"] == 0


def golden_wide_trajectories(device, orc, golden_dir):
    """g10: seeded reference games on 12x20 (default set) and 11x24 (seven pieces) boards, all seeds of a config
    as one batch, pieces from the NumPy-exact device stream: every output of every step and the afterstate
    matrices of seed 0."""
    from tetris_amd import VecTetris
    g = np.load(os.path.join(golden_dir, "g10_traj_wide.npz"))
    for tag, pieces, C, R in (("default_12x20", "default", 12, 20), ("standard7_11x24", STANDARD7, 11, 24)):
        S = 8
        T = len(g[tag + "_s0_action"])
        env = VecTetris(C, R, S, device=device, pieces=pieces, auto_reset=True, numpy_seeds=list(range(S)),
                        stream_len=2 * T + 8)
        for t in range(T):
            np.testing.assert_array_equal(env.piece.cpu().numpy(), [g["%s_s%d_piece" % (tag, s)][t] for s in range(S)])
            np.testing.assert_array_equal(env.n_valid.cpu().numpy(), [g["%s_s%d_n_valid" % (tag, s)][t] for s in range(S)])
            if t < 60:
                f, nv, fa, na = env.get_after_states(include_terminal=True)
                np.testing.assert_array_equal(f[0].cpu().numpy(), g[tag + "_s0_after_valid"][t][:env.a_max])
                np.testing.assert_array_equal(fa[0].cpu().numpy(), g[tag + "_s0_after_all"][t][:env.a_max])
            act = np.array([g["%s_s%d_action" % (tag, s)][t] for s in range(S)], np.int32)
            obs, rew, done, lines = env.step(torch.from_numpy(act))
            obs, rew, done, lines = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), lines.cpu().numpy()
            cols = env.columns().cpu().numpy().astype(np.uint64).T
            for s in range(S):
                p = "%s_s%d_" % (tag, s)
                np.testing.assert_array_equal(obs[s], g[p + "obs"][t])
                assert int(rew[s]) == g[p + "reward"][t] and bool(done[s]) == bool(g[p + "done"][t])
                assert int(lines[s]) == g[p + "lines"][t]
                if not g[p + "done"][t]:
                    np.testing.assert_array_equal(cols[s], g[p + "cols"][t])
        env.check()


def golden_placements_afterstates(device, orc, golden_dir):
    """g1 fixtures: every placement of all 9 pieces on hand-made boards, via set_boards + afterstates."""
    from tetris_amd import VecTetris
    from tetris_amd.tetromino import CATALOGUE
    for name in ("g1_placements_10x20.npz", "g1_placements_10x40.npz", "g1_placements_6x10.npz", "g1_placements_12x20.npz"):
        g = np.load(os.path.join(golden_dir, name))
        R, C = int(g["R"]), int(g["C"])
        boards = g["boards"]
        nb = len(boards)
        cells = orc.cols_to_cells(boards, R + 4)
        env = VecTetris(C, R, nb, device=device, pieces=list(CATALOGUE))
        for pi in range(len(CATALOGUE)):
            env.set_boards(cells, piece=np.full(nb, pi))
            f, nv, fa, na = env.get_after_states(include_terminal=True)
            fa, na, f, nv = fa.cpu().numpy(), na.cpu().numpy(), f.cpu().numpy(), nv.cpu().numpy()
            for b in range(nb):
                sel = (g["board_ix"] == b) & (g["piece"] == pi)
                want = g["feats"][sel]
                term = g["terminal"][sel].astype(bool)
                assert na[b] == len(want) and nv[b] == int((~term).sum())
                np.testing.assert_array_equal(fa[b, :na[b]], want)
                np.testing.assert_array_equal(f[b, :nv[b]], want[~term])
                assert env.n_valid[b] == nv[b]


def sharding_equals_single_batch(device, B=2048):
    """Two shards with env_offset draw the same pieces as one batch (multi-GPU layout, section 8e)."""
    from tetris_amd import VecTetris
    whole = VecTetris(10, 20, B, device=device, auto_reset=True, seed=9)
    lo = VecTetris(10, 20, B // 2, device=device, auto_reset=True, seed=9, env_offset=0)
    hi = VecTetris(10, 20, B // 2, device=device, auto_reset=True, seed=9, env_offset=B // 2)
    for t in range(100):
        a = whole.random_actions().clone()
        whole.step(a)
        lo.step(a[:B // 2].contiguous())
        hi.step(a[B // 2:].contiguous())
    assert torch.equal(whole.cols, torch.cat([lo.cols, hi.cols], dim=0))
    assert torch.equal(whole.obs, torch.cat([lo.obs, hi.obs]))
    assert torch.equal(whole.meta, torch.cat([lo.meta, hi.meta]))


def full_size_properties(device, B=1 << 20, steps=60, R=20):
    """BASELINE config 3 size (1,048,576 envs): size-independent invariants."""
    from tetris_amd import VecTetris
    env = VecTetris(10, R, B, device=device, auto_reset=True, seed=1)
    cells_before = None
    total_lines = 0
    for t in range(steps):
        before = env.boards().sum(dim=(1, 2), dtype=torch.int32) if t % 20 == 0 else None
        piece_cells = 3  # both default pieces have 3 cells
        obs, rew, done, lines = env.step()
        if before is not None:
            after = env.boards().sum(dim=(1, 2), dtype=torch.int32)
            keep = ~done
            # cell conservation: +piece cells, -10 per cleared line
            assert torch.equal(after[keep], (before + piece_cells - 10 * lines.to(torch.int32))[keep])
            assert (after[done] == 0).all()
        assert torch.equal(rew, lines.to(torch.int32) - 1 - 100 * done.to(torch.int32))
        h = env.heights()
        assert int(h.max()) <= R  # playable states never reach the overflow rows
        assert ((env.n_valid > 0) | ~done).all()
    st = env.stats()
    assert st["invalid"] == 0 and st["steps"] == steps * B and st["episodes"] > 0
    # no full row survives in any board
    cols = env.columns()
    full = cols[0]
    for c in range(1, 10):
        full = full & cols[c]
    assert not full.any()


def mask_rescue_stress(device, orc, n_boards=1500, R=20, C=10, seed=0):
    """Near-top boards whose rows R-3..R-1 miss only a few cells: the placements that poke
    above row R-1 are valid only if their line clear pulls the stack back (state.py:33 before
    :36).  Valid masks / n_valid / feature rows vs the oracle for all nine pieces."""
    from tetris_amd import VecTetris
    from tetris_amd.tetromino import CATALOGUE
    rng = np.random.default_rng(seed)
    rows = R + 4
    cells = np.zeros((n_boards, rows, C), np.int8)
    for b in range(n_boards):
        hts = rng.integers(R - 5, R + 1, size=C)
        for c in range(C):
            hc = int(hts[c])
            colv = (rng.random(hc) > 0.15).astype(np.int8)
            if hc:
                colv[hc - 1] = 1
            cells[b, :hc, c] = colv
        # make 1-3 of the top rows nearly full: missing run of width 1..4
        for r in rng.choice(np.arange(R - 4, R), size=rng.integers(1, 4), replace=False):
            w = int(rng.integers(1, 5))
            c0 = int(rng.integers(0, C - w + 1))
            cells[b, r, :] = 1
            cells[b, r, c0:c0 + w] = 0
            # columns in the gap must not have cells above the gap row (heights stay consistent)
            cells[b, r:, c0:c0 + w] = 0
        for r in range(rows):  # no full rows in a reachable board
            if cells[b, r].sum() == C:
                cells[b, r, rng.integers(0, C)] = 0
        # cells above a removed cell may now float: that is fine for the reference semantics
        # as long as heights are recomputed from the board (State(lowest_free_rows=None))
        cells[b, R:, :] = 0
    desc = orc.make_desc(C, R, list(CATALOGUE))
    env = VecTetris(C, R, n_boards, device=device, pieces=list(CATALOGUE))
    n_rescued = 0
    for pi, name in enumerate(CATALOGUE):
        env.set_boards(cells, piece=np.full(n_boards, pi))
        f, nv, fa, na = env.get_after_states(include_terminal=True)
        f, nv, fa, na = f.cpu().numpy(), nv.cpu().numpy(), fa.cpu().numpy(), na.cpu().numpy()
        env_nv = env.n_valid.cpu().numpy()
        for b in range(n_boards):
            out = orc.placements(desc, cells[b], name)
            term = out["terminal"].astype(bool)
            assert na[b] == len(term)
            assert nv[b] == int((~term).sum()) == env_nv[b], (name, b, nv[b], int((~term).sum()), env_nv[b])
            np.testing.assert_array_equal(fa[b, :na[b]], out["feats"])
            np.testing.assert_array_equal(f[b, :nv[b]], out["feats"][~term])
            # rescued = poked above R-1 before the clear yet non-terminal after it
            n_rescued += int(((out["anchor_row"] + 0 >= 0) & (~term) & (out["n_cleared"] > 0) &
                              (out["heights"].max(axis=1) + out["n_cleared"] > R)).sum())
    assert n_rescued > 50, n_rescued
    return n_rescued


def edge_geometries(device, orc):
    """Smallest / largest row counts of each word size, odd batch sizes, every column count built."""
    eps = 0
    for C, R, pieces, B in [(10, 4, "default", 130), (10, 27, "standard7", 65), (10, 28, "standard7", 63),
                            (10, 59, "default", 31), (6, 4, "standard7", 1), (8, 27, "default", 257),
                            (6, 59, "standard7", 64),
                            # every column count the library is built for (5..12), odd ones included
                            (5, 12, "standard7", 33), (5, 20, "default", 40), (7, 24, "standard7", 50),
                            (9, 20, "default", 70), (9, 40, "standard7", 30), (5, 30, "standard7", 20),
                            # 11 and 12 columns: 12-bit level fields / 64-bit missing-cell words in the valid mask,
                            # nine board planes, a_max = 44; both word sizes, packed and unpacked boards
                            (11, 20, "default", 70), (12, 20, "standard7", 90), (12, 20, "default", 64),
                            (11, 40, "standard7", 40), (12, 33, "default", 50), (12, 24, "standard7", 45),
                            (12, 10, ["Straight", "ThreeLine", "Square"], 60),
                            # either side of every kernel-variant switch: 10-row / 12-row table chunks (R = 20 | 21),
                            # chunk borders inside the board (R = 10, 12, 13), compile-time chunk counts of the
                            # u64 kernels (stored rows 36 | 37 and 48 | 49)
                            (6, 20, "default", 77), (10, 21, "standard7", 70), (10, 10, "standard7", 45),
                            (8, 12, "default", 33), (10, 13, "standard7", 40), (10, 32, "default", 50),
                            (10, 33, "standard7", 50), (10, 44, "default", 40), (10, 45, "standard7", 40)]:
        eps += lockstep(device, orc, C, R, B, pieces, steps=90, seed=3, check_after_every=15)
    assert eps > 0


def step_without_obs(device, orc, B=300):
    """compute_obs=False: identical dynamics, obs untouched; the skipped observation equals row
    `action` of the get_after_states matrix (game.py:91 vs :70-72)."""
    from tetris_amd import VecTetris
    a = VecTetris(10, 20, B, device=device, auto_reset=True, seed=4)
    b = VecTetris(10, 20, B, device=device, auto_reset=True, seed=4, compute_obs=False)
    for t in range(60):
        feats, nv = b.get_after_states()
        obs, rew, done, lines = a.step()
        _, rew2, done2, lines2 = b.step()
        assert torch.equal(a.action, b.action) and torch.equal(rew, rew2) and torch.equal(done, done2)
        assert torch.equal(a.cols, b.cols) and torch.equal(a.meta, b.meta) and not b.obs.any()
        picked = feats[torch.arange(B, device=feats.device), b.action.long()]
        assert torch.equal(picked, obs)


def greedy_policy(device, orc, golden_dir, B=400):
    """tetris_hip_policy_greedy vs (a) float32 left-to-right fitness of the oracle's features and
    (b) the get_best_policy vectors recorded from the reference (g6)."""
    from tetris_amd import VecTetris
    from tetris_amd.tetromino import CATALOGUE
    w32 = np.array(VecTetris.BCTS_WEIGHTS, np.float32)

    def fit(feats):  # game.py:109-118 under NumPy >= 2: float32 products and sums, left to right
        acc = feats[..., 0] * w32[0]
        for q in range(1, 8):
            acc = (acc + feats[..., q] * w32[q]).astype(np.float32)
        return acc

    env = VecTetris(10, 20, B, device=device, pieces="standard7", auto_reset=True, seed=8)
    ref = orc.OracleVecEnv(10, 20, B, pieces="standard7", auto_reset=True, seed=8)
    for t in range(40):
        ba, bv, fa = env.greedy_actions(include_fitness=True)
        rf, rnv, rfa, rna = ref.afterstates(include_terminal=True)
        want_all = fit(rfa)
        got = fa.cpu().numpy()
        for i in range(B):
            np.testing.assert_array_equal(got[i, :rna[i]], want_all[i, :rna[i]])
            if rnv[i]:
                v = fit(rf[i, :rnv[i]])
                assert int(ba[i]) == int(np.argmax(v)) and float(bv[i]) == float(v.max())
            else:
                assert int(ba[i]) == -1
        # play the greedy action (random where none exists: those envs are done anyway)
        act = torch.where(ba >= 0, ba, torch.zeros_like(ba))
        env.step(act)
        ref.step(act.cpu().numpy())
    # recorded reference policies
    g = np.load(os.path.join(golden_dir, "g6_policy.npz"))
    for tag, pieces in (("default_20", "default"), ("standard7_20", STANDARD7)):
        boards, plist = g[tag + "_boards"], g[tag + "_pieces"]
        n = len(boards)
        e2 = VecTetris(10, 20, n, device=device, pieces=pieces)
        e2.set_boards(orc.cols_to_cells(boards, 24), piece=plist.astype(np.int64))
        _, _, fa = e2.greedy_actions(include_fitness=True)
        fa = fa.cpu().numpy()
        for t in range(n):
            pol_ref = g[tag + "_policy"][t]
            na = int((pol_ref > 0).nonzero()[0].max()) + 1 if pol_ref.any() else 0
            k = len(np.trim_zeros(g[tag + "_fitness"][t], "b"))
            k = max(k, na)
            f = fa[t, :k]
            pol = (f == f.max()).astype(float)
            pol /= pol.sum()
            np.testing.assert_array_equal(pol, pol_ref[:k])
            np.testing.assert_array_equal(f, g[tag + "_fitness"][t][:k].astype(np.float32))


def rollouts(device, orc, B=96, geometries=((10, "default", 20, 5), (10, "standard7", 12, 6))):
    """tetris_hip_rollouts (game.py:129-160 fan-out) vs the oracle, both policies, mid-game boards."""
    from tetris_amd import VecTetris
    for C, pieces, R, seed in geometries:
        env = VecTetris(C, R, B, device=device, pieces=pieces, auto_reset=False, seed=seed)
        ref = orc.OracleVecEnv(C, R, B, pieces=pieces, auto_reset=False, seed=seed, nthreads=0)
        for t in range(int(R * 1.6)):  # play towards the top so that rollouts do die
            env.step()
            ref.step()
        np.testing.assert_array_equal(env.boards().cpu().numpy(), ref.cells)
        before = (env.cols.clone(), env.meta.clone())
        for policy, length, n in (("random", 5, 5), ("random", 8, 3), ("greedy", 4, 2)):
            got = env.rollouts(length=length, n=n, policy=policy).cpu().numpy()
            want = ref.rollouts(length=length, n=n, policy=policy)
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(want, nan=7.0))
            vals = np.unique(got[~np.isnan(got)])
            assert len(vals) > 2, vals  # deaths (-1 in the mean) and survivals both occur
        assert torch.equal(env.cols, before[0]) and torch.equal(env.meta, before[1])  # envs untouched


def step_many_equals_steps(device, orc, B=700):
    """tetris_hip_step_many(K) == K launches of tetris_hip_step (random policy), and the greedy
    variant against step(greedy_actions()); also checked against the oracle."""
    from tetris_amd import VecTetris
    for rows, pieces in ((20, "default"), (40, "standard7")):
        a = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=13)
        b = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=13)
        ref = orc.OracleVecEnv(10, rows, B, pieces=pieces, auto_reset=True, seed=13, nthreads=0)
        out = None
        for rep in range(4):
            K = 7
            out = a.step_many(K, out=out)
            for k in range(K):
                obs, rew, done, lines = b.step()
                o_obs, o_rew, o_done, o_lines, n_bad = ref.step()
                assert torch.equal(out["obs"][k], obs) and torch.equal(out["reward"][k], rew)
                assert torch.equal(out["done"][k], done) and torch.equal(out["lines"][k], lines)
                assert torch.equal(out["action"][k], b.action) and torch.equal(out["n_valid"][k], b.n_valid)
                assert torch.equal(out["piece"][k], b.piece)
                np.testing.assert_array_equal(obs.cpu().numpy(), o_obs)
                np.testing.assert_array_equal(rew.cpu().numpy(), o_rew)
            assert torch.equal(a.cols, b.cols) and torch.equal(a.meta, b.meta)
            assert a.stats() == b.stats() and a.step_idx == b.step_idx
            # the per-step attributes describe the last fused step
            assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
            assert torch.equal(a.lines, b.lines) and torch.equal(a.action, b.action)
            assert torch.equal(a.n_valid, b.n_valid) and torch.equal(a.piece, b.piece)
            np.testing.assert_array_equal(a.boards().cpu().numpy(), ref.cells)
    # greedy policy fused vs unfused
    a = VecTetris(10, 20, 200, device=device, pieces="standard7", auto_reset=True, seed=2)
    b = VecTetris(10, 20, 200, device=device, pieces="standard7", auto_reset=True, seed=2)
    out = a.step_many(12, policy="greedy")
    for k in range(12):
        ba, _ = b.greedy_actions()
        obs, rew, done, lines = b.step(ba.clone())
        assert torch.equal(out["action"][k], ba) and torch.equal(out["obs"][k], obs) and torch.equal(out["done"][k], done)
    assert torch.equal(a.cols, b.cols) and torch.equal(a.meta, b.meta)


def terminal_boards_are_refused(device):
    """A board with a cell in the overflow rows is a terminal State (state.py:33-36): set_boards refuses
    it (the kernels assume current boards have none) and the facade's is_game_over answers True."""
    import pytest
    from tetris_amd import Tetris, VecTetris
    from tetris_amd.state import State
    C, R = 10, 20
    env = VecTetris(C, R, 2, device=device, auto_reset=False, seed=0)
    cells = np.zeros((2, R + 4, C), np.int8)
    cells[0, :R, 3] = 1            # column 3 filled to the top legal row: still a legal board
    env.set_boards(torch.from_numpy(cells), piece=np.array([0, 0]))
    assert int(env.n_valid[0]) > 0
    cells[1, R, 5] = 1             # a cell in row R
    with pytest.raises(ValueError):
        env.set_boards(torch.from_numpy(cells), piece=np.array([0, 0]))
    game = Tetris(C, R, device=device)
    rep = np.zeros((R + 4, C), np.int_)
    rep[:R + 1, 2] = 1
    assert game.is_game_over(State(rep)) is True


DIRS = [-1, -1, -1, -1, -1, -1, 1, -1]  # the reference's usual feature_directions


def golden_edges_through_kernels(device, orc, golden_dir):
    """g4 edge boards (3-line rescue of an over-height piece, 4-line Straight clear, one valid
    placement, all dead: state.py:33 before :36) through set_boards -> get_after_states(include_terminal)
    AND one step per valid action, against the fixture recorded from the reference."""
    from tetris_amd import VecTetris
    from tetris_amd.tetromino import CATALOGUE
    g = np.load(os.path.join(golden_dir, "g4_edges.npz"))
    R, C = 20, 10
    for name in ("e1_rescue", "e2_tetris", "e3_onevalid", "e4_dead"):
        board, pi = g[name + "_board"], int(g[name + "_piece"])
        term = g[name + "_terminal"].astype(bool)
        feats, kid_cols, ncl = g[name + "_feats"], g[name + "_cols"], g[name + "_n_cleared"]
        nv = int((~term).sum())
        B = max(nv, 1)
        cells = orc.cols_to_cells(np.repeat(board[None], B, axis=0), R + 4)
        env = VecTetris(C, R, B, device=device, pieces=list(CATALOGUE), auto_reset=False, seed=1)
        env.set_boards(cells, piece=np.full(B, pi))
        f, n1, fa, na = env.get_after_states(include_terminal=True)
        assert int(na[0]) == len(term) and int(n1[0]) == nv == int(env.n_valid[0])
        np.testing.assert_array_equal(fa[0, :len(term)].cpu().numpy(), feats)
        np.testing.assert_array_equal(f[0, :nv].cpu().numpy(), feats[~term])
        assert not f[0, nv:].any()
        if name == "e4_dead":
            assert nv == 0
            env.step(torch.zeros(B, dtype=torch.int32))  # game.py:83: IndexError
            with pytest.raises(IndexError):
                env.check()
            continue
        # env k plays valid action k
        obs, rew, done, lines = env.step(torch.arange(B, dtype=torch.int32))
        env.check()
        np.testing.assert_array_equal(obs.cpu().numpy(), feats[~term])
        np.testing.assert_array_equal(lines.cpu().numpy(), ncl[~term])
        np.testing.assert_array_equal(env.columns().cpu().numpy().astype(np.uint64).T, kid_cols[~term])
        np.testing.assert_array_equal(rew.cpu().numpy(), ncl[~term].astype(np.int32) - 1 - 100 * done.cpu().numpy())
    assert int(g["e1_rescue_n_cleared"][0]) == 3 and int(g["e2_tetris_n_cleared"][0]) == 4


def golden_placements_through_step(device, orc, golden_dir):
    """Every NON-terminal g1 placement (all nine pieces, 10x20 / 10x40 / 6x10) played with tetris_hip_step:
    new board, lowest_free_rows (tetris_hip_decode heights vs the reference's own array), lines and
    observation against the fixture."""
    from tetris_amd import VecTetris
    from tetris_amd.tetromino import CATALOGUE
    for fname in ("g1_placements_10x20.npz", "g1_placements_10x40.npz", "g1_placements_6x10.npz", "g1_placements_12x20.npz"):
        g = np.load(os.path.join(golden_dir, fname))
        R, C = int(g["R"]), int(g["C"])
        term = g["terminal"].astype(bool)
        bix, pc = g["board_ix"], g["piece"]
        # action index of a placement = its rank among the non-terminal placements of its (board, piece)
        key = bix.astype(np.int64) * 16 + pc
        act = np.zeros(len(term), np.int32)
        for k in np.unique(key):
            sel = np.nonzero(key == k)[0]
            act[sel] = np.cumsum(~term[sel]) - 1
        keep = np.nonzero(~term)[0]
        B = len(keep)
        cells = orc.cols_to_cells(g["boards"][bix[keep]], R + 4)
        env = VecTetris(C, R, B, device=device, pieces=list(CATALOGUE), auto_reset=False, seed=2)
        env.set_boards(cells, piece=pc[keep].astype(np.int64))
        # heights of the parent boards (state.py:162-172)
        want_h = (cells != 0).any(axis=1) * (R + 4 - np.argmax(cells[:, ::-1, :] != 0, axis=1))
        np.testing.assert_array_equal(env.heights().cpu().numpy(), want_h)
        obs, rew, done, lines = env.step(torch.from_numpy(act[keep]))
        env.check()
        np.testing.assert_array_equal(env.columns().cpu().numpy().astype(np.uint64).T, g["cols"][keep])
        np.testing.assert_array_equal(env.heights().cpu().numpy(), g["heights"][keep])
        np.testing.assert_array_equal(lines.cpu().numpy(), g["n_cleared"][keep])
        np.testing.assert_array_equal(obs.cpu().numpy(), g["feats"][keep])


def feature_directions_in_kernels(device, orc, golden_dir, B=500):
    """The kernels' direct_by multiply (state.py:49-50 through game.py:72,91): step observations and both
    afterstate matrices of VecTetris(feature_directions=d) == those of the plain env times d, and the g5
    values recorded from the reference."""
    from tetris_amd import VecTetris
    d = torch.tensor(DIRS, dtype=torch.float32, device=device)
    for rows, pieces in ((20, "default"), (40, "standard7")):
        a = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=6)
        b = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=6, feature_directions=DIRS)
        for t in range(50):
            fa, na, faa, naa = a.get_after_states(include_terminal=True)
            fb, nb, fba, nba = b.get_after_states(include_terminal=True)
            assert torch.equal(fa * d, fb) and torch.equal(faa * d, fba) and torch.equal(na, nb)
            oa, ra, da, la = a.step()
            ob, rb, db, lb = b.step()
            assert torch.equal(oa * d, ob) and torch.equal(ra, rb) and torch.equal(a.cols, b.cols)
        out = b.step_many(5)
        for k in range(5):
            oa, _, _, _ = a.step()
            assert torch.equal(out["obs"][k], oa * d)
    g = np.load(os.path.join(golden_dir, "g5_dtypes.npz"))
    np.random.seed(0)
    rng = orc.NumpyLegacyRNG(0)
    bag = orc.BagSampler(rng, 2)
    stream = np.array([[bag.next()], [bag.next()], [0], [0]], np.uint8)
    e = VecTetris(10, 20, 1, device=device, feature_directions=DIRS, piece_stream=stream)
    f, nv = e.get_after_states()
    np.testing.assert_array_equal(f[0, :int(nv[0])].cpu().numpy().astype(np.float64), g["after_directed"])
    obs, _, _, _ = e.step(torch.zeros(1, dtype=torch.int32))
    np.testing.assert_array_equal(obs[0].cpu().numpy().astype(np.float64), g["obs_directed"])


def action_major_layout(device, B=333):
    """afterstate_layout="action_major" ([a_max, B, 8] storage) returns the same [B, a_max, 8] values."""
    from tetris_amd import VecTetris
    for rows, pieces in ((20, "default"), (40, "standard7")):
        a = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=3)
        b = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=3,
                      afterstate_layout="action_major")
        for t in range(40):
            fa, na, faa, naa = a.get_after_states(include_terminal=True)
            fb, nb, fba, nba = b.get_after_states(include_terminal=True)
            assert fb.shape == fa.shape and not fb.is_contiguous()
            assert torch.equal(fa, fb) and torch.equal(faa, fba) and torch.equal(na, nb) and torch.equal(naa, nba)
            a.step()
            b.step()


def state_dict_roundtrip(device, B=600):
    """state_dict mid-episode -> load_state_dict in another env continues bit-identically (device bag and
    replay stream)."""
    from tetris_amd import VecTetris
    for kw in (dict(), dict(piece_stream=np.random.default_rng(0).integers(0, 2, size=(400, B)).astype(np.uint8))):
        a = VecTetris(10, 20, B, device=device, auto_reset=True, seed=17, **kw)
        for t in range(33):
            a.step()
        sd = a.state_dict()
        ref = []
        for t in range(25):
            o, r, d, l = a.step()
            ref.append((o.clone(), r.clone(), d.clone(), l.clone(), a.action.clone(), a.piece.clone()))
        b = VecTetris(10, 20, B, device=device, auto_reset=True, seed=99, **kw)
        for t in range(5):
            b.step()
        b.load_state_dict(sd)
        for t in range(25):
            o, r, d, l = b.step()
            for x, y in zip(ref[t], (o, r, d, l, b.action, b.piece)):
                assert torch.equal(x, y)
        assert torch.equal(a.cols, b.cols) and torch.equal(a.meta, b.meta) and a.stats() == b.stats()
        with pytest.raises(ValueError):
            VecTetris(10, 40, B, device=device).load_state_dict(sd)


def replay_stream_exhaustion(device, B=70):
    """An env that runs out of recorded pieces is reported (counted as invalid, left untouched), never
    continued on a repeated last row."""
    from tetris_amd import VecTetris
    L = 12
    stream = np.random.default_rng(1).integers(0, 2, size=(L, B)).astype(np.uint8)
    env = VecTetris(10, 20, B, device=device, auto_reset=True, piece_stream=stream)
    for t in range(L - 2):  # reset took row 0; each step needs cursor + 2 <= L under auto-reset
        env.step()
        env.check()
    before = (env.cols.clone(), env.meta.clone())
    env.step()
    assert env.stats()["invalid"] == B
    assert torch.equal(env.cols, before[0]) and torch.equal(env.meta, before[1])
    with pytest.raises(IndexError):
        env.check()
    # without auto-reset one row per step is enough
    env = VecTetris(10, 20, B, device=device, auto_reset=False, piece_stream=stream)
    for t in range(L - 1):
        env.step(torch.zeros(B, dtype=torch.int32))
    assert env.stats()["invalid"] == 0
    env.step(torch.zeros(B, dtype=torch.int32))
    assert env.stats()["invalid"] == B
    # a reset past the end of the stream is reported the same way (it consumes one row): untouched, counted
    before = (env.cols.clone(), env.meta.clone(), env._cursor.clone())
    env.reset()
    assert env.stats()["invalid"] == 2 * B
    assert torch.equal(env.cols, before[0]) and torch.equal(env.meta, before[1]) and torch.equal(env._cursor, before[2])
    half = torch.arange(B, device=env.device) % 2 == 0
    env._cursor[half] = 3  # these envs still have rows left
    env.reset()
    assert env.stats()["invalid"] == 2 * B + int((~half).sum())
    assert not env.boards()[half].any() and torch.equal(env._cursor[half], torch.full_like(env._cursor[half], 4))
    assert torch.equal(env.cols.clone().view(-1), env.cols.view(-1)) and torch.equal(env.meta[~half], before[1][~half])


def gather_payload_from_the_step(device, B=64 * 37 + 11):
    """tetris_hip_step_call_run_gather: the done bitmask and the counter snapshot the step kernel writes from
    its epilogue equal what the separate kernels produce (pack_done_bits(done), the live counter slots) at that
    step, and later steps do not touch them."""
    from tetris_amd import VecTetris
    from tetris_amd.distributed import pack_done_bits, unpack_done_bits
    for auto in (True, False):
        env = VecTetris(10, 20, B, device=device, auto_reset=auto, seed=3)
        pl = [env.gather_payload(), env.gather_payload()]
        seen_done = 0
        for t in range(90):
            p = pl[t & 1]
            env.step(gather=p) if t % 3 else env.step(env.random_actions().clone(), gather=p)
            want_bits = pack_done_bits(env.done)
            assert torch.equal(p["done_bits"][:want_bits.numel()], want_bits), t
            assert torch.equal(unpack_done_bits(p["done_bits"], B), env.done)
            assert torch.equal(p["counters"], env.status.view(-1, 4))
            seen_done += int(env.done.sum())
            snap = (p["done_bits"].clone(), p["counters"].clone())
            env.step()  # a plain step in between leaves the payload alone
            assert torch.equal(p["done_bits"], snap[0]) and torch.equal(p["counters"], snap[1])
            assert not torch.equal(p["counters"], env.status.view(-1, 4))
        assert seen_done > 0
        if auto:
            env.check()


def device_bag_properties(device, B=1 << 20, steps=48):
    """tetromino.py:12-22 / game.py:50 for the counter-based device bag, checked on the bag bits of
    `meta` alone (no oracle mirror): every draw removes a piece that was in the bag, a bag is refilled
    only when empty (so the draws between refills form a permutation of the piece list), and the bag
    crosses in-kernel auto-reset and host reset(mask) untouched."""
    from tetris_amd import VecTetris
    for pieces, n in (("default", 2), ("standard7", 7),
                      (["Straight", "Square", "SnakeR", "ThreeLine", "ThreeL", "SnakeL", "T", "RCorner", "LCorner"], 9)):
        full = (1 << n) - 1
        env = VecTetris(10, 20 if n != 9 else 10, B if n == 2 else max(B // 16, 4096), device=device, pieces=pieces,
                        auto_reset=True, seed=123)

        def bag_of():
            return (env.meta >> 52) & 0xFFF

        def piece_bit():
            return torch.ones_like(env.meta) << env.piece.to(torch.int64)

        def popc(x):
            c = torch.zeros_like(x)
            for b in range(12):
                c += (x >> b) & 1
            return c

        bag = bag_of()
        # construction: one draw from a fresh bag
        assert torch.equal(bag, full ^ piece_bit())
        first_counts = torch.bincount(env.piece.to(torch.int64), minlength=n).double()
        assert (first_counts / first_counts.sum() - 1.0 / n).abs().max() < 0.01 + 2.0 / n / (env.batch_size ** 0.5) * 3
        n_done = 0
        for t in range(steps):
            before = bag
            _, _, done, _ = env.step()
            bag = bag_of()
            pb = piece_bit()
            assert ((bag & pb) == 0).all() and (bag <= full).all()          # the piece in play left the bag
            src = torch.where(before == 0, torch.full_like(before, full), before)
            one = ~done
            assert torch.equal(bag[one], (src ^ pb)[one]) and ((src & pb) != 0)[one].all()
            # a finished episode drew twice (game.py:87 then :60): the bag shrank by two pieces, refilling
            # when it ran empty in between, and was NOT re-initialised
            two = done
            if two.any():
                n_done += int(two.sum())
                s0 = src[two]
                s1 = bag[two] | pb[two]            # the bag the second draw saw
                refilled = s1 == full              # ... a fresh one: the first draw must have emptied its bag
                assert (popc(s0)[refilled] == 1).all()
                mid = s1[~refilled]                # otherwise: the first bag minus exactly one piece
                assert ((mid & ~s0[~refilled]) == 0).all() and (popc(s0[~refilled]) - popc(mid) == 1).all()
        assert n_done > 0
        # host reset of some envs: one draw, bag otherwise kept
        before = bag
        mask = torch.zeros(env.batch_size, dtype=torch.bool, device=env.device)
        mask[::3] = True
        env.reset(mask=mask)
        bag, pb = bag_of(), piece_bit()
        src = torch.where(before == 0, torch.full_like(before, full), before)
        assert torch.equal(bag[mask], (src ^ pb)[mask]) and torch.equal(bag[~mask], before[~mask])


def rollouts_pinned_to_reference(device, orc, golden_dir):
    """g7 part B: tetris_hip_rollouts against returns recorded from the reference's single_rollout
    (game.py:129-146) on one-piece sets -- the piece sequence is then deterministic -- with the greedy
    float32 policy: -1 on death at any step, else the rewards of steps 2..length."""
    from tetris_amd import VecTetris
    g = np.load(os.path.join(golden_dir, "g7_rollouts.npz"))
    w = [float(x) for x in g["weights"]]
    for tag, name in (("one_T_10", "T"), ("one_ThreeL_8", "ThreeL"), ("one_Straight_9", "Straight")):
        R, length = int(g[tag + "_rows"]), int(g[tag + "_length"])
        boards, want = g[tag + "_boards"], g[tag + "_returns"]
        env = VecTetris(10, R, len(boards), device=device, pieces=[name], auto_reset=False, seed=0)
        env.set_boards(orc.cols_to_cells(boards, R + 4), piece=np.zeros(len(boards), np.int64))
        for n in (1, 3):  # deterministic game: the mean over n rollouts is the single return
            got = env.rollouts(length=length, n=n, policy="greedy", weights=w).cpu().numpy()
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want[:, :env.a_max]))
            np.testing.assert_array_equal(np.nan_to_num(got, nan=9.0), np.nan_to_num(want[:, :env.a_max], nan=9.0))
        assert (want == -1).any() and (want < -1).any()


def rollouts_fed_pieces(device, orc, golden_dir):
    """g9: perform_rollouts on MULTI-piece sets (game.py:129-160) -- returns recorded from the reference
    together with the piece its sampler handed out at every rollout step (the global bag advances across
    rollouts) -- reproduced by tetris_hip_rollouts with those pieces fed in: mean of the n returns per first
    action, -1 on death, NaN beyond n_valid."""
    from tetris_amd import VecTetris
    g = np.load(os.path.join(golden_dir, "g9_rollouts_fed.npz"))
    w = [float(x) for x in g["weights"]]
    for tag, pieces in (("default_9", "default"), ("standard7_11", STANDARD7)):
        R, length, n = int(g[tag + "_rows"]), int(g[tag + "_length"]), int(g[tag + "_n"])
        boards, want = g[tag + "_boards"], g[tag + "_returns"]
        env = VecTetris(10, R, len(boards), device=device, pieces=pieces, auto_reset=False, seed=0)
        env.set_boards(orc.cols_to_cells(boards, R + 4), piece=g[tag + "_piece"].astype(np.int64))
        fed = g[tag + "_fed"][:, :env.a_max]
        got = env.rollouts(length=length, n=n, policy="greedy", weights=w, pieces=fed).cpu().numpy()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want[:, :env.a_max]))
        np.testing.assert_array_equal(np.nan_to_num(got, nan=9.0), np.nan_to_num(want[:, :env.a_max], nan=9.0))
        v = want[~np.isnan(want)]
        assert (v == -1).any() and (v < -1).any()  # deaths and full runs
        if tag == "standard7_11":
            assert (v != np.round(v)).any()  # a mean over rollouts that fared differently (the bag moved on)
        # the fed pieces matter: with the env's own bag fork the returns differ somewhere
        own = env.rollouts(length=length, n=n, policy="greedy", weights=w).cpu().numpy()
        assert not np.array_equal(np.nan_to_num(own, nan=9.0), np.nan_to_num(got, nan=9.0))
        with pytest.raises(ValueError):
            env.rollouts(length=length, n=n + 1, policy="greedy", weights=w, pieces=fed)


def cfg3_full_size_bit_exact(device, orc, B=1 << 20, steps=48, R=20, pieces="default", board_every=12):
    """BASELINE config 3 at its full size -- 1,048,576 envs, 10x20 -- in lock-step with the oracle: every
    output of every step bit-exact, boards compared every 12 steps (a 252 MB decode each)."""
    from tetris_amd import VecTetris
    env = VecTetris(10, R, B, device=device, pieces=pieces, auto_reset=True, seed=0)
    ref = orc.OracleVecEnv(10, R, B, pieces=pieces, auto_reset=True, seed=0, nthreads=0)
    episodes = 0
    for t in range(steps):
        obs, rew, done, lines = env.step()  # policy in the kernel, as bench.py runs it
        o_obs, o_rew, o_done, o_lines, n_bad = ref.step()
        assert n_bad == 0
        np.testing.assert_array_equal(env.action.cpu().numpy(), ref.action)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs, err_msg="obs t=%d" % t)
        np.testing.assert_array_equal(rew.cpu().numpy(), o_rew)
        np.testing.assert_array_equal(done.cpu().numpy(), o_done.astype(bool))
        np.testing.assert_array_equal(lines.cpu().numpy(), o_lines)
        np.testing.assert_array_equal(env.n_valid.cpu().numpy(), ref.n_valid)
        np.testing.assert_array_equal(env.piece.cpu().numpy(), ref.piece)
        if t % board_every == board_every - 1:
            np.testing.assert_array_equal(env.boards().cpu().numpy(), ref.cells)
            if os.environ.get("TETRIS_SOAK_PROGRESS"):
                print("  step %d ok" % (t + 1), flush=True)
        episodes += int(o_done.sum())
    st = env.stats()
    assert st["invalid"] == 0 and st["episodes"] == episodes and st["steps"] == steps * B
    assert episodes > 1000 or steps < 40
    return episodes


def afterstate_family_full_size(device, orc, B=1 << 18, R=20, C=10, pieces="default", steps=24, every=8):
    """get_after_states (valid and include-terminal matrices) and get_best_policy for EVERY env of a large
    batch against the oracle, every `every` steps of steady-state random play; whole-array comparisons."""
    from tetris_amd import VecTetris
    w32 = np.array(VecTetris.BCTS_WEIGHTS, np.float32)
    env = VecTetris(C, R, B, device=device, pieces=pieces, auto_reset=True, seed=3)
    ref = orc.OracleVecEnv(C, R, B, pieces=pieces, auto_reset=True, seed=3, nthreads=0)
    checks = 0
    for t in range(steps):
        if t % every == every - 1:
            f, nv, fa, na = env.get_after_states(include_terminal=True)
            rf, rnv, rfa, rna = ref.afterstates(include_terminal=True)
            np.testing.assert_array_equal(nv.cpu().numpy(), rnv)
            np.testing.assert_array_equal(na.cpu().numpy(), rna)
            np.testing.assert_array_equal(f.cpu().numpy(), rf)
            np.testing.assert_array_equal(fa.cpu().numpy(), rfa)
            del f, fa
            ba, bv, fit = env.greedy_actions(include_fitness=True)
            A = rfa.shape[1]
            want_all = rfa[..., 0] * w32[0]  # game.py:109-118: float32 products and sums, left to right
            want = rf[..., 0] * w32[0]
            for q in range(1, 8):
                want_all = (want_all + rfa[..., q] * w32[q]).astype(np.float32)
                want = (want + rf[..., q] * w32[q]).astype(np.float32)
            k = np.arange(A)[None, :]
            got = fit.cpu().numpy()[:, :A]
            np.testing.assert_array_equal(np.where(k < rna[:, None], got, 0), np.where(k < rna[:, None], want_all, 0))
            want = np.where(k < rnv[:, None], want, -np.inf)
            best = np.where(rnv > 0, want.argmax(axis=1), -1)
            np.testing.assert_array_equal(ba.cpu().numpy(), best)
            has = rnv > 0
            np.testing.assert_array_equal(bv.cpu().numpy()[has], want.max(axis=1)[has])
            checks += 1
            if os.environ.get("TETRIS_SOAK_PROGRESS"):
                print("  step %d: matrices + policy of %d envs ok" % (t + 1, B), flush=True)
        env.step()
        ref.step()
    assert checks >= 1
    return checks


def repeat_and_shard_consistency(device, **kw):
    """tetris_amd.selftest.afterstate_family_consistency (the shipped self-test): repeated launches agree,
    whole batch == shards, twin envs under the in-kernel greedy policy stay identical."""
    from tetris_amd import selftest
    return selftest.afterstate_family_consistency(device, **kw)


def numpy_exact_bag_stream(device, orc, golden_dir):
    """tetris_hip_numpy_bag_stream = the reference's TetrominoSampler on NumPy's legacy MT19937 stream
    (tetromino.py:12-22): (a) against the bags np.random.permutation produced in the reference's
    process for seeds 0..15 (g3), (b) with a batch of envs seeded 0..5 replaying the recorded g2
    games piece for piece, with no host-side stream."""
    from tetris_amd import VecTetris
    g = np.load(os.path.join(golden_dir, "g3_rng.npz"))
    for n in (2, 7, 9):
        bags = g["bags_n%d" % n]                       # [16 seeds, 64 bags, n]
        want = bags.reshape(16, -1).T.astype(np.uint8)  # [64 n, 16]: bag after bag, front to back
        got = VecTetris.numpy_piece_stream(np.arange(16), n, want.shape[0], device).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    # large seeds (32-bit) and a long stream (several MT19937 twists) against the oracle's restatement
    seeds = np.array([0, 1, 2**31 - 1, 2**32 - 1, 123456789], np.uint64)
    got = VecTetris.numpy_piece_stream(seeds, 7, 3000, device).cpu().numpy()
    for k, s in enumerate(seeds):
        bag = orc.BagSampler(orc.NumpyLegacyRNG(int(s)), 7)
        np.testing.assert_array_equal(got[:, k], [bag.next() for _ in range(3000)])
    # whole games: every seeded reference run of a config as one batch
    for tag, R in (("default", 20), ("standard7", 40)):
        gg = np.load(os.path.join(golden_dir, "g2_traj_%s_10x%d.npz" % (tag, R)))
        pieces = "default" if tag == "default" else STANDARD7
        T = len(gg["s0_action"])
        S = n_golden_seeds(gg)
        env = VecTetris(10, R, S, device=device, pieces=pieces, auto_reset=True, numpy_seeds=list(range(S)),
                        stream_len=2 * T + 8)
        for t in range(T):
            np.testing.assert_array_equal(env.piece.cpu().numpy(), [gg["s%d_piece" % s][t] for s in range(S)])
            act = np.array([gg["s%d_action" % s][t] for s in range(S)], np.int32)
            obs, rew, done, lines = env.step(torch.from_numpy(act))
            obs, rew, done = obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            for s in range(S):
                np.testing.assert_array_equal(obs[s], gg["s%d_obs" % s][t])
                assert int(rew[s]) == gg["s%d_reward" % s][t] and bool(done[s]) == bool(gg["s%d_done" % s][t])
        env.check()


def graph_steps_equal_steps(device, B=5000):
    """VecTetris.capture_steps(K): replays of the captured graph == the same number of step() calls
    (built-in policy and an action tensor produced inside the graph), also when plain steps come
    in between."""
    from tetris_amd import VecTetris
    for rows, pieces in ((20, "default"), (40, "standard7")):
        a = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=31)
        b = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=31)
        for _ in range(3):
            a.step()
            b.step()
        g = a.capture_steps(6)
        for rep in range(4):
            obs, rew, done, lines = g.replay()
            for k in range(6):
                b.step()
            assert torch.equal(a.cols, b.cols) and torch.equal(a.meta, b.meta) and torch.equal(obs, b.obs)
            assert torch.equal(a.action, b.action) and torch.equal(done, b.done) and a.step_idx == b.step_idx
            if rep == 1:  # plain steps in between
                a.step()
                b.step()
        assert a.stats() == b.stats()
        # actions computed inside the graph (here: the policy kernel into a static tensor)
        a2 = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=32)
        b2 = VecTetris(10, rows, B, device=device, pieces=pieces, auto_reset=True, seed=32)
        act = torch.zeros(B, dtype=torch.int32, device=a2.device)

        def action_fn(env, k):
            # greedy on feature 3 (landing height) among the valid placements, computed by torch ops
            f, nv = env.get_after_states()
            score = f[:, :, 3] + torch.where(torch.arange(env.a_max, device=f.device)[None, :] < nv[:, None].long(), 0.0, 1e9)
            act.copy_(score.argmin(dim=1).to(torch.int32))
            return act

        if a2.device.type == "cuda":
            action_fn(a2, 0)  # warm the allocator outside the capture
        g2 = a2.capture_steps(3, action_fn=action_fn)
        for rep in range(3):
            g2.replay()
            for k in range(3):
                f, nv = b2.get_after_states()
                score = f[:, :, 3] + torch.where(torch.arange(b2.a_max, device=f.device)[None, :] < nv[:, None].long(), 0.0, 1e9)
                b2.step(score.argmin(dim=1).to(torch.int32))
            assert torch.equal(a2.cols, b2.cols) and torch.equal(a2.meta, b2.meta) and torch.equal(a2.obs, b2.obs)
        a2.check()


def done_bit_packing(device):
    """tetris_hip_pack_done_bits (one ballot per wavefront) == the torch formulation, odd sizes included."""
    from tetris_amd import VecTetris, _lib
    from tetris_amd.distributed import pack_done_bits, unpack_done_bits
    VecTetris(10, 20, 64, device=device)  # makes sure the library is loaded
    for n in (1, 63, 64, 65, 1003, 1 << 20):
        d = torch.rand(n, device=device) < 0.3
        got = pack_done_bits(d)
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=d.device)
        pad = (-n) % 8
        dd = torch.cat([d.to(torch.uint8), d.new_zeros(pad, dtype=torch.uint8)]) if pad else d.to(torch.uint8)
        want = (dd.view(-1, 8) * w).sum(dim=1).to(torch.uint8)
        assert torch.equal(got, want), n
        assert torch.equal(unpack_done_bits(got, n), d)
