"""Pins the CPU oracle (oracle/tetris_oracle.c) to fixtures captured from the
live reference by tests/golden/make_golden.py.  CPU only."""
import os

import numpy as np
import pytest

CATALOGUE = ["Straight", "Square", "SnakeR", "ThreeLine", "ThreeL", "SnakeL", "T", "RCorner", "LCorner"]
STANDARD7 = ["Straight", "RCorner", "LCorner", "Square", "SnakeR", "SnakeL", "T"]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("name", ["g1_placements_10x20.npz", "g1_placements_10x40.npz", "g1_placements_6x10.npz",
                                  "g1_placements_12x20.npz"])
def test_g1_placements(orc, golden_dir, name):
    """tetromino.py get_after_states + state.py clear/terminal/features, all 9 pieces."""
    g = _load(golden_dir, name)
    R, C = int(g["R"]), int(g["C"])
    rows = R + 4
    desc = orc.make_desc(C, R)
    k = 0
    n_checked = 0
    n_cleared_total = 0
    for b, board in enumerate(g["boards"]):
        cells = orc.cols_to_cells(board, rows)
        for pi, pname in enumerate(CATALOGUE):
            out = orc.placements(desc, cells, pname)
            n = len(out["terminal"])
            sel = slice(k, k + n)
            assert (g["board_ix"][sel] == b).all() and (g["piece"][sel] == pi).all()
            np.testing.assert_array_equal(orc.cells_to_cols(out["cells"]), g["cols"][sel])
            np.testing.assert_array_equal(out["heights"], g["heights"][sel])
            np.testing.assert_array_equal(out["n_cleared"], g["n_cleared"][sel])
            np.testing.assert_array_equal(out["terminal"], g["terminal"][sel])
            np.testing.assert_array_equal(out["anchor_row"], g["anchor_row"][sel])
            np.testing.assert_array_equal(out["anchor_col"], g["anchor_col"][sel])
            np.testing.assert_array_equal(out["feats"], g["feats"][sel])  # bit-exact float32
            k += n
            n_checked += n
            n_cleared_total += int(out["n_cleared"].sum())
    assert k == len(g["piece"])
    assert n_checked > 1000 and n_cleared_total > 0


def test_placement_counts(orc):
    """SURVEY App. A totals at C = 10."""
    want = dict(Straight=17, Square=9, SnakeR=17, ThreeLine=18, ThreeL=36, SnakeL=17, T=34, RCorner=34, LCorner=34)
    for name, n in want.items():
        assert orc.n_placements(name, 10) == n


def test_g3_numpy_legacy_rng(orc, golden_dir):
    """np.random.seed(s); np.random.permutation(n) restated on raw MT19937."""
    g = _load(golden_dir, "g3_rng.npz")
    for n in (2, 7, 9):
        bags = g[f"bags_n{n}"]
        for s in range(bags.shape[0]):
            rng = orc.NumpyLegacyRNG(s)
            for b in range(bags.shape[1]):
                np.testing.assert_array_equal(rng.permutation(n), bags[s, b])


def test_numpy_rng_against_installed_numpy(orc):
    for s in (0, 1, 12345, 2**32 - 1):
        np.random.seed(s)
        rng = orc.NumpyLegacyRNG(s)
        for n in (2, 3, 7, 9, 16):
            np.testing.assert_array_equal(rng.permutation(n), np.random.permutation(n))


def _replay(orc, golden_dir, tag, R, seed):
    g = _load(golden_dir, f"g2_traj_{tag}_10x{R}.npz")
    p = f"s{seed}_"
    T = len(g[p + "action"])
    names = "default" if tag == "default" else STANDARD7
    n_pieces = 2 if tag == "default" else 7
    # reproduce the piece stream from the oracle's own MT19937 + bag
    rng = orc.NumpyLegacyRNG(seed)
    bag = orc.BagSampler(rng, n_pieces)
    stream = []
    # draws: one at construction/reset, one per step, one more per reset-after-done
    n_draws = 1 + T + int(g[p + "done"].sum())
    for _ in range(n_draws):
        stream.append(bag.next())
    stream = np.array(stream, np.uint8)[:, None]
    env = orc.OracleVecEnv(10, R, 1, pieces=names, auto_reset=True, piece_stream=stream)
    return g, p, T, env


@pytest.mark.parametrize("tag", ["default", "standard7"])
@pytest.mark.parametrize("R", [20, 40])
def test_g2_trajectories(orc, golden_dir, tag, R):
    """game.py step/reset/is_game_over + bag-across-reset vs recorded runs."""
    total_done = 0
    for seed in range(32):  # seeds 0..31
        g, p, T, env = _replay(orc, golden_dir, tag, R, seed)
        assert env.piece[0] == int(g[p + "first_piece"])
        for t in range(T):
            assert env.piece[0] == g[p + "piece"][t], (seed, t)
            assert env.n_valid[0] == g[p + "n_valid"][t], (seed, t)
            if (p + "after_valid") in g:
                fv, nv, fa, na = env.afterstates(include_terminal=True)
                assert nv[0] == g[p + "n_valid"][t] and na[0] == g[p + "n_all"][t]
                A = fv.shape[1]
                np.testing.assert_array_equal(fv[0], g[p + "after_valid"][t][:A])
                np.testing.assert_array_equal(fa[0], g[p + "after_all"][t][:A])
                assert not g[p + "after_valid"][t][A:].any()
            obs, reward, done, lines, n_bad = env.step(np.array([g[p + "action"][t]]))
            assert n_bad == 0
            np.testing.assert_array_equal(obs[0], g[p + "obs"][t])
            assert reward[0] == g[p + "reward"][t] and done[0] == g[p + "done"][t] and lines[0] == g[p + "lines"][t]
            if done[0]:
                total_done += 1
                # auto-reset already happened: board empty
                assert not env.cells.any()
            else:
                np.testing.assert_array_equal(orc.cells_to_cols(env.cells[0]), g[p + "cols"][t])
        assert env.piece[0] == int(g[p + "last_piece"])
    assert total_done > 10


def test_g4_edges(orc, golden_dir):
    g = _load(golden_dir, "g4_edges.npz")
    desc = orc.make_desc(10, 20)
    for name in ("e1_rescue", "e2_tetris", "e3_onevalid", "e4_dead"):
        cells = orc.cols_to_cells(g[name + "_board"], 24)
        out = orc.placements(desc, cells, CATALOGUE[int(g[name + "_piece"])])
        np.testing.assert_array_equal(orc.cells_to_cols(out["cells"]), g[name + "_cols"])
        np.testing.assert_array_equal(out["n_cleared"], g[name + "_n_cleared"])
        np.testing.assert_array_equal(out["terminal"], g[name + "_terminal"])
        np.testing.assert_array_equal(out["feats"], g[name + "_feats"])
    assert g["e1_rescue_n_cleared"][0] == 3 and g["e1_rescue_terminal"][0] == 0
    assert g["e2_tetris_n_cleared"][0] == 4
    assert int(g["e5_last_reward"]) == -101 + int(g["e5_last_lines"])
    assert int(g["e5_post_done_nvalid"]) == 0 and int(g["e5_post_done_step_raises"]) == 1
    for R in (20, 40):
        f = orc.board_features(orc.make_desc(10, R), np.zeros((R + 4, 10), np.int8))
        np.testing.assert_array_equal(f, g[f"e6_reset_feats_{R}"])


def test_invalid_action_flag(orc):
    env = orc.OracleVecEnv(10, 20, 4, seed=1)
    before = env.cells.copy()
    a = np.array([0, 99, -1, int(env.n_valid[3])])
    _, _, _, _, n_bad = env.step(a)
    assert n_bad == 3 and list(env.invalid) == [0, 1, 1, 1]
    np.testing.assert_array_equal(env.cells[1:], before[1:])


@pytest.mark.parametrize("tag,names,C,R", [("default_12x20", "default", 12, 20), ("standard7_11x24", STANDARD7, 11, 24)])
def test_g10_wide_trajectories(orc, golden_dir, tag, names, C, R):
    """Seeded reference games on boards of 11 and 12 columns (game.py:21-28 takes any width)."""
    g = _load(golden_dir, "g10_traj_wide.npz")
    n_pieces = 2 if names == "default" else len(names)
    done_total = 0
    for seed in range(8):
        p = "%s_s%d_" % (tag, seed)
        T = len(g[p + "action"])
        bag = orc.BagSampler(orc.NumpyLegacyRNG(seed), n_pieces)
        stream = np.array([bag.next() for _ in range(1 + T + int(g[p + "done"].sum()))], np.uint8)[:, None]
        env = orc.OracleVecEnv(C, R, 1, pieces=names, auto_reset=True, piece_stream=stream)
        for t in range(T):
            assert env.piece[0] == g[p + "piece"][t] and env.n_valid[0] == g[p + "n_valid"][t], (seed, t)
            if (p + "after_valid") in g:
                fv, nv, fa, na = env.afterstates(include_terminal=True)
                A = fv.shape[1]
                assert na[0] == g[p + "n_all"][t]
                np.testing.assert_array_equal(fv[0], g[p + "after_valid"][t][:A])
                np.testing.assert_array_equal(fa[0], g[p + "after_all"][t][:A])
            obs, reward, done, lines, n_bad = env.step(np.array([g[p + "action"][t]]))
            assert n_bad == 0
            np.testing.assert_array_equal(obs[0], g[p + "obs"][t])
            assert reward[0] == g[p + "reward"][t] and done[0] == g[p + "done"][t] and lines[0] == g[p + "lines"][t]
            done_total += int(done[0])
    assert done_total > 0
