"""Facade (tetris_amd.Tetris, batch of one) against the recorded reference runs: same
constructor, same protocol, same NumPy global RNG -> same game.  `device` as in parity_cases."""
import os

import numpy as np

STANDARD7 = ["Straight", "RCorner", "LCorner", "Square", "SnakeR", "SnakeL", "T"]


def _cols_of(rep):
    rep = np.asarray(rep).astype(np.uint64)
    w = (np.uint64(1) << np.arange(rep.shape[0], dtype=np.uint64))[:, None]
    return (rep * w).sum(axis=0).astype(np.uint64)


def golden_trajectory(device, golden_dir, tag, R, seed, steps=None):
    from tetris_amd import Tetris
    g = np.load(os.path.join(golden_dir, "g2_traj_%s_10x%d.npz" % (tag, R)))
    p = "s%d_" % seed
    T = len(g[p + "action"]) if steps is None else steps
    np.random.seed(seed)  # the facade's bag uses the same global legacy stream as the reference
    env = Tetris(10, R, pieces="default" if tag == "default" else STANDARD7, device=device)
    assert env.tetrominos.index(env.current_tetromino) == int(g[p + "first_piece"])
    for t in range(T):
        assert env.tetrominos.index(env.current_tetromino) == g[p + "piece"][t]
        fv, fa = env.get_after_states(include_terminal=True)
        assert fv.dtype == np.float64 and fv.shape == (g[p + "n_valid"][t], 8) and fa.shape == (g[p + "n_all"][t], 8)
        if (p + "after_valid") in g:
            np.testing.assert_array_equal(fv, g[p + "after_valid"][t][:len(fv)])
            np.testing.assert_array_equal(fa, g[p + "after_all"][t][:len(fa)])
        obs, reward, done, lines = env.step(int(g[p + "action"][t]))
        assert obs.dtype == np.float32 and isinstance(reward, int) and isinstance(done, bool) and isinstance(lines, int)
        np.testing.assert_array_equal(obs, g[p + "obs"][t])
        assert (reward, done, lines) == (g[p + "reward"][t], bool(g[p + "done"][t]), g[p + "lines"][t])
        np.testing.assert_array_equal(_cols_of(env.current_state.representation), g[p + "cols"][t])
        assert env.current_state.n_cleared_lines == lines
        if done:
            assert env.get_after_states()[0].shape == (0, 8)  # post-done protocol
            try:
                env.step(0)
                raise AssertionError("step after done must raise IndexError (game.py:83)")
            except IndexError:
                pass
            st, piece = env.reset()
            assert not st.representation.any()
    return env


def dtypes_and_directions(device, golden_dir):
    from tetris_amd import Tetris
    g = np.load(os.path.join(golden_dir, "g5_dtypes.npz"))
    np.random.seed(0)
    env = Tetris(10, 20, device=device)
    env.get_after_states()
    o1 = env.step(0)[0]
    assert str(o1.dtype) == str(g["obs_plain_dtype"])
    np.testing.assert_array_equal(o1, g["obs_plain"])
    np.random.seed(0)
    env = Tetris(10, 20, feature_directions=np.array([-1, -1, -1, -1, -1, -1, 1, -1]), device=device)
    fv, _ = env.get_after_states()
    o2 = env.step(0)[0]
    assert str(o2.dtype) == str(g["obs_directed_dtype"]) and str(fv.dtype) == str(g["after_dtype"])
    np.testing.assert_array_equal(o2, g["obs_directed"])
    np.testing.assert_array_equal(fv, g["after_directed"])


def reset_features(device, golden_dir):
    from tetris_amd import Tetris
    g = np.load(os.path.join(golden_dir, "g4_edges.npz"))
    for R in (20, 40):
        env = Tetris(10, R, device=device)
        st, _ = env.reset()
        np.testing.assert_array_equal(st.get_features(), g["e6_reset_feats_%d" % R])
        np.testing.assert_array_equal(env.get_state(), g["e6_reset_feats_%d" % R])


def best_policy(device, golden_dir):
    """get_best_policy / fitness (game.py:102-120) on recorded states."""
    from tetris_amd import Tetris
    from tetris_amd.state import State
    g = np.load(os.path.join(golden_dir, "g6_policy.npz"))
    for tag, pieces in (("default_20", "default"), ("standard7_20", STANDARD7)):
        env = Tetris(10, 20, pieces=pieces, device=device)
        boards, plist = g[tag + "_boards"], g[tag + "_pieces"]
        for t in range(0, len(boards), 3):
            rep = ((boards[t][None, :] >> np.arange(24, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(np.int_)
            env._restore(State(rep), env.tetrominos[int(plist[t])])
            pol = env.get_best_policy()
            want = g[tag + "_policy"][t][:len(pol)]
            np.testing.assert_array_equal(pol, want)
            assert not g[tag + "_policy"][t][len(pol):].any()


def rollouts_and_misc(device):
    from tetris_amd import Tetris
    np.random.seed(4)
    env = Tetris(10, 10, device=device)
    assert env.is_game_over(env.current_state) is False
    fv, _ = env.get_after_states()
    before = env.current_state.representation.copy()
    piece_before = env.current_tetromino

    def greedy(state, feats):
        return int(np.argmin(feats[:, 2] + feats[:, 3]))

    acts, rets = env.perform_rollouts(list(range(min(4, len(fv)))), greedy, length=4, n=2)
    assert len(acts) == len(rets) == min(4, len(fv))
    # rollouts restore the env (game.py:147-148)
    np.testing.assert_array_equal(env.current_state.representation, before)
    assert env.current_tetromino is piece_before
    fv2, _ = env.get_after_states()
    np.testing.assert_array_equal(fv, fv2)
    # afterstates entries expose boards on demand
    child = env.afterstates[0]
    assert child.representation.sum() == before.sum() + 3 - 10 * child.n_cleared_lines
    # stale afterstates are refused, numpy-style negative indices accepted
    env.step(-1)
    try:
        env.step(0)
        raise AssertionError("stale afterstates must be refused")
    except RuntimeError:
        pass
    # wrong feature type fails where the reference does (state.py:91-95)
    bad = Tetris(10, 10, feature_type="foo", device=device)
    try:
        bad.get_after_states()
        raise AssertionError
    except ValueError:
        pass
    assert "|" in repr(env.current_state) and "██" in repr(env.tetrominos[0])


def _policy_holes_height(state, feats):
    return int(np.argmin(feats[:, 2] + feats[:, 3]))


def _bag_row(env):
    return np.pad(np.asarray(env.tetromino_sampler.current_batch, np.int64) + 1, (0, 12))[:12]


def _rng_fingerprint():
    import zlib
    st = np.random.get_state()
    return np.array([st[2], zlib.crc32(st[1].tobytes())], np.int64)


def rollout_script(device, golden_dir):
    """g7 part A: the seeded script of single_rollout / perform_rollouts calls recorded from the reference
    (game.py:129-160): every return, the bag and NumPy's global stream after every block, and the
    restored state must match.  perform_rollouts with length > 1 is compared with the reference's loop
    run on a refreshed afterstate list (see make_golden.gen_rollouts for the stale-list quirk)."""
    from tetris_amd import Tetris
    g = np.load(os.path.join(golden_dir, "g7_rollouts.npz"))
    for tag, pieces, R in (("default_10", "default", 10), ("standard7_12", STANDARD7, 12)):
        np.random.seed(int(g[tag + "_seed"]))
        env = Tetris(10, R, pieces=pieces, device=device)
        k = 0
        deaths = 0
        for op in g[tag + "_ops"]:
            if op >= 0:
                env.get_after_states()
                env.step(int(op))
            elif op == -1:
                env.reset()
            else:
                np.testing.assert_array_equal(_cols_of(env.current_state.representation), g[tag + "_boards"][k])
                piece = env.tetrominos.index(env.current_tetromino)
                assert piece == g[tag + "_pieces"][k]
                rec = g[tag + "_single"][k]
                length, n_act = int(rec[0]), int(rec[1])
                fv, _ = env.get_after_states()
                assert min(fv.shape[0], 6) == n_act
                for act in range(n_act):
                    env.get_after_states()
                    r = env.single_rollout(act, _policy_holes_height, length)
                    assert r == rec[2 + act], (tag, k, act, r, rec[2 + act])
                    deaths += int(r == -1 and length > 2)
                    assert env.tetrominos.index(env.current_tetromino) == piece       # game.py:147-148
                    np.testing.assert_array_equal(_cols_of(env.current_state.representation), g[tag + "_boards"][k])
                np.testing.assert_array_equal(_bag_row(env), g[tag + "_bags"][2 * k])  # the bag advanced as upstream
                np.testing.assert_array_equal(_rng_fingerprint(), g[tag + "_rng"][2 * k])
                env.get_after_states()
                acts, rets = env.perform_rollouts(list(range(min(fv.shape[0], 4))), _policy_holes_height, length=1, n=3)
                np.testing.assert_array_equal(rets, g[tag + "_perform_len1"][k][:len(rets)])
                assert acts == list(range(min(fv.shape[0], 4)))
                acts, rets = env.perform_rollouts(list(range(min(fv.shape[0], 3))), _policy_holes_height,
                                                  length=length, n=2)
                np.testing.assert_array_equal(rets, g[tag + "_perform_fresh"][k][:len(rets)])
                np.testing.assert_array_equal(_bag_row(env), g[tag + "_bags"][2 * k + 1])
                np.testing.assert_array_equal(_rng_fingerprint(), g[tag + "_rng"][2 * k + 1])
                k += 1
        assert k == len(g[tag + "_boards"]) and deaths > 0


def render_strings(device, golden_dir):
    """g8: State.__repr__ / print_board_to_string (state.py:69-81: the R legal rows), utils.print_board_to_string
    (utils.py:179-191: all R + 4 rows) and the piece reprs, character for character."""
    from tetris_amd import Tetris
    from tetris_amd.state import State, print_board_to_string
    g = np.load(os.path.join(golden_dir, "g8_render.npz"))
    for i in range(int(g["n_states"])):
        cols = g["s%d_cols" % i]
        rep = ((cols[None, :] >> np.arange(12, dtype=np.uint64)[:, None]) & np.uint64(1)).astype(np.int_)
        st = State(rep)
        assert st.print_board_to_string() == str(g["s%d_state_str" % i])
        assert repr(st) == str(g["s%d_repr" % i])
        assert print_board_to_string(st) == str(g["s%d_utils_str" % i])
    env = Tetris(10, 8, pieces=["Straight", "Square", "SnakeR", "ThreeLine", "ThreeL", "SnakeL", "T", "RCorner",
                                "LCorner"], device=device)
    for t in env.tetrominos:
        if int(g["piece_%s_has_custom_repr" % t.name]):
            assert repr(t) == str(g["piece_%s_repr" % t.name]), t.name
        else:
            assert "██" in repr(t)  # upstream falls back to the default object repr (tetromino.py:162 typo)


def unsupported_widths_are_refused(device):
    """game.py:21-28 accepts any num_columns; this build has kernels for 5..12 columns
    and says so: the facade and the batched env raise TetrisHipError with the library's text, for widths on
    both sides of the range, before anything is allocated or launched."""
    import pytest
    from tetris_amd import Tetris, VecTetris
    from tetris_amd._lib import TetrisHipError
    for C in (4, 13, 16, 40):
        with pytest.raises(TetrisHipError, match="num_columns not compiled into libtetris_hip"):
            Tetris(C, 20, device=device)
        with pytest.raises(TetrisHipError, match="num_columns not compiled into libtetris_hip"):
            VecTetris(C, 20, 8, device=device)
    for C in (5, 12):  # the ends of the supported range work
        env = Tetris(C, 12, device=device)
        fv, _ = env.get_after_states()
        assert fv.shape[0] > 0 and fv.shape[1] == 8
