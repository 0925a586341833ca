"""Tetris: single-env facade with the surface of the reference's ``game.Tetris``
(game.py:8-160), backed by a batch-of-one :class:`VecTetris` on the GPU.

Same constructor arguments, same method names, same return shapes / dtypes,
same protocol (``get_after_states()`` then ``step(action)``), same use of
NumPy's GLOBAL legacy RNG for the piece bag, so that
``np.random.seed(s); env = Tetris(10, 20)`` plays the same game as the reference.
Differences are listed in DESIGN.md ("facade deviations").
"""
import numpy as np
import torch

from .state import State, print_board_to_string
from .tetromino import ORIENTATIONS, TetrominoSampler
from .vec_env import VecTetris



def _placements_in_action_order(mask, C):
    """(loop, column, orientation) of the set bits of a valid mask in the reference's
    enumeration order.  Mask layout: four 12-bit fields, field 2L + o at bit 12 (2L + o), bit c."""
    out = []
    for loop in (0, 1):
        for c in range(C):
            for o in (0, 1):
                if (mask >> (12 * (2 * loop + o) + c)) & 1:
                    out.append((loop, c, o))
    return out


class Tetris:
    """
    Features (game.py:9-19): 0 rows_with_holes, 1 column_transitions, 2 holes,
    3 landing height, 4 cumulative_wells, 5 row_transitions, 6 eroded pieces, 7 hole_depth
    """

    def __init__(self, num_columns, num_rows, feature_directions=None, feature_type="bcts", num_features=8,
                 tetromino_size=4, pieces="default", device="cuda"):
        if tetromino_size != 4 or num_features != 8:
            raise ValueError("the kernels implement the reference's live configuration: "
                             "tetromino_size=4 (four overflow rows) and num_features=8")
        self.feature_directions = feature_directions
        self.num_columns = num_columns
        self.num_rows = num_rows
        self.feature_type = feature_type
        self.num_features = num_features
        self.tetromino_size = tetromino_size
        self.loss_reward = -100     # game.py:34
        self.timestep_reward = -1   # game.py:35

        # replay buffer of two pieces: [0] the draw inside step()/reset(), [1] unused
        self._env = VecTetris(num_columns, num_rows, 1, device=device, pieces=pieces, auto_reset=False,
                              piece_stream=np.zeros((2, 1), np.uint8))
        self._scratch = None
        self._action_buf = torch.zeros(1, dtype=torch.int32, device=self._env.device)
        self.tetrominos = self._env.tetrominos
        self.tetromino_sampler = TetrominoSampler(self.tetrominos)  # game.py:50 (draws bag #1)
        self._after_token = None
        self._token = 0
        self.current_state, self.current_tetromino = self.reset()  # game.py:51

    # ------------------------------------------------------------------ plumbing
    def _feed_piece(self, piece):
        self._env._stream[0, 0] = piece.list_index
        self._env._cursor.zero_()

    def _decode_state(self, **kw):
        rep = self._env.boards()[0].cpu().numpy().astype(np.int_)
        return State(rep, lowest_free_rows=self._env.heights()[0].cpu().numpy().astype(np.int_), **kw)

    def _push_state(self, env, state, tetromino):
        env.set_boards(torch.from_numpy(np.ascontiguousarray(state.representation, dtype=np.int8))[None],
                       piece=np.array([tetromino.list_index]))

    def _check_feature_type(self):
        if self.feature_type != "bcts":
            raise ValueError("Only 'bcts' features implemented.")  # state.py:91-95

    # ------------------------------------------------------------------ game.py:53-63
    def reset(self):
        piece = self.tetromino_sampler.next_tetromino()  # game.py:60; the bag survives reset
        self._feed_piece(piece)
        self._env.reset()
        # reset State: changed_lines = [0], bonus 0 (state.py:7-9) -> landing height feature 1
        feats = self._board_features()
        self.current_state = self._decode_state(features=feats)
        self.current_tetromino = piece
        self._token += 1
        return self.current_state, self.current_tetromino

    def _board_features(self):
        """Features of the empty board seen as the reset State (state.py:7-9: changed_lines = [0],
        bonus 0): one column transition per column (state.py:194), landing height 0 + 0 + 1
        (state.py:102), row transitions R for the right wall (state.py:190) + R for the left
        wall next to the empty column 0 (state.py:253-254)."""
        f = np.zeros(8, np.float32)
        f[1], f[3], f[5] = self.num_columns, 1.0, 2 * self.num_rows
        return f

    def _get_scratch(self):
        if self._scratch is None:
            self._scratch = VecTetris(self.num_columns, self.num_rows, 1, device=self._env.device,
                                      pieces=self._env.piece_names, piece_stream=np.zeros((2, 1), np.uint8))
        return self._scratch

    # ------------------------------------------------------------------ game.py:67-80
    def get_after_states(self, include_terminal=False):
        self._check_feature_type()
        f, nv, fa, na = self._env.get_after_states(include_terminal=True)
        # everything the host needs in ONE device -> host copy: both feature matrices, the two counts and
        # the 48-bit valid mask (as two 24-bit halves: exact in float32)
        A = self._env.a_max
        m = self._env.meta[0] & ((1 << 48) - 1)
        box = torch.cat([f[0].reshape(-1), fa[0].reshape(-1),
                         torch.stack([nv[0].float(), na[0].float(), (m & 0xFFFFFF).float(), (m >> 24).float()])]).cpu().numpy()
        n_valid, n_all = int(box[2 * A * 8]), int(box[2 * A * 8 + 1])
        mask = int(box[2 * A * 8 + 2]) | (int(box[2 * A * 8 + 3]) << 24)
        feats = box[:A * 8].reshape(A, 8)[:n_valid].copy()
        feats_all = box[A * 8:2 * A * 8].reshape(A, 8)[:n_all].copy()
        slots = _placements_in_action_order(mask, self.num_columns)
        assert len(slots) == n_valid
        self._after_slots = slots
        self._after_feats = feats
        self._after_token = self._token
        self._after_parent = (self.current_state, self.current_tetromino)
        self.afterstates = np.array([_AfterState(self, k) for k in range(n_valid)], dtype=object)  # game.py:69
        action_features = np.zeros((n_valid, self.num_features))  # float64, game.py:70
        for ix in range(n_valid):
            action_features[ix] = self._directed(feats[ix])
        if include_terminal:
            all_afterstates = np.zeros((n_all, self.num_features))
            for ix in range(n_all):
                all_afterstates[ix] = self._directed(feats_all[ix])
            return action_features, all_afterstates
        return action_features, None

    def _directed(self, feats):
        return feats if self.feature_directions is None else feats * self.feature_directions  # state.py:49-50

    # ------------------------------------------------------------------ game.py:82-92
    def step(self, action):
        if not hasattr(self, "afterstates"):
            raise AttributeError("'Tetris' object has no attribute 'afterstates'")  # game.py:83 before :67 ran
        if self._after_token != self._token:
            raise RuntimeError("step() needs a fresh get_after_states() for the current state "
                               "(the reference would silently reuse the stale list)")
        n_valid = len(self.afterstates)
        k = int(action)
        if not -n_valid <= k < n_valid:  # numpy indexing of self.afterstates: game.py:83
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (k, n_valid))
        k %= n_valid
        loop, col, oi = self._after_slots[k]
        w, b, n = ORIENTATIONS[self.current_tetromino.name][loop][oi]
        bonus = (max(bj + nj for bj, nj in zip(b, n)) - 1) / 2.0
        nxt = self.tetromino_sampler.next_tetromino()  # game.py:87
        self._feed_piece(nxt)
        env = self._env
        env.step(self._action_buf.fill_(k))
        # outputs, the invalid counter and the decoded new state in ONE device -> host copy
        box = torch.cat([env.obs[0], torch.stack([env.reward[0].float(), env._done[0].float(), env.lines[0].float(),
                                                  (env.status.view(-1, 4)[:, 0].sum()).float()]),
                         env.boards()[0].reshape(-1).float(), env.heights()[0].float()]).cpu().numpy()
        obs = box[:8].astype(np.float32)
        reward, done, lines = int(box[8]), bool(box[9]), int(box[10])
        if box[11]:
            raise IndexError("out-of-range action reached the kernel")  # (cannot happen: checked above)
        R4, C = self.num_rows + 4, self.num_columns
        rep = box[12:12 + R4 * C].reshape(R4, C).astype(np.int_)
        heights = box[12 + R4 * C:12 + R4 * C + C].astype(np.int_)
        self.current_state = State(rep, lowest_free_rows=heights, features=obs, anchor_col=col,
                                   anchor_row=int(round(float(obs[3]) - bonus - 1.0)), n_cleared_lines=lines,
                                   landing_height_bonus=bonus)
        self.current_tetromino = nxt
        self._token += 1
        return self.get_state(), reward, done, lines

    # ------------------------------------------------------------------ game.py:94-100
    def is_game_over(self, state):
        if np.any(np.asarray(state.representation)[self.num_rows:]):
            return True  # a terminal State (cells in the overflow rows): every placement on it stays terminal
        sc = self._get_scratch()
        self._push_state(sc, state, self.current_tetromino)
        return int(sc.n_valid[0]) == 0

    # ------------------------------------------------------------------ game.py:102-120
    _BCTS_WEIGHTS = np.array([-24.04, -19.77, -13.08, -12.63, -10.49, -9.22, 6.6, -1.61])

    def get_best_policy(self):
        self._check_feature_type()
        _, _, fa, na = self._env.get_after_states(include_terminal=True)
        feats = fa[0, :int(na[0])].cpu().numpy()  # ALL afterstates, terminal included (game.py:103)
        fitness = np.array([self._fitness_of(f) for f in feats])
        best = (fitness == fitness.max()).astype(float)
        return best / best.sum()

    @staticmethod
    def _fitness_of(f):
        # same left-to-right float32*float64 accumulation as game.py:109-118
        return (f[0] * -24.04 + f[1] * -19.77 + f[2] * -13.08 + f[3] * -12.63 +
                f[4] * -10.49 + f[5] * -9.22 + f[6] * 6.6 + f[7] * -1.61)

    def fitness(self, state):
        return self._fitness_of(state.get_features())

    # ------------------------------------------------------------------ game.py:122-127
    def render(self):
        print(print_board_to_string(self.current_state))
        print(self.current_tetromino)

    def get_state(self):
        self._check_feature_type()
        return self.current_state.get_features(direct_by=self.feature_directions)

    # ------------------------------------------------------------------ game.py:129-160
    def _restore(self, state, tetromino):
        self._push_state(self._env, state, tetromino)
        self.current_state, self.current_tetromino = state, tetromino
        self._token += 1

    def single_rollout(self, action, policy_function, length):
        reset_state, reset_tetromino = self.current_state, self.current_tetromino
        if self.is_game_over(reset_state):
            return -1
        if self._after_token != self._token:
            self.get_after_states()
        _, _, done, _ = self.step(action)
        if done:
            self._restore(reset_state, reset_tetromino)
            return -1
        rollout_return = 0
        for _ in range(length - 1):
            act = policy_function(self.current_state, self.get_after_states(include_terminal=True)[0])
            _, reward, done, _ = self.step(act)
            rollout_return += reward
            if done:
                rollout_return = -1
                break
        self._restore(reset_state, reset_tetromino)
        return rollout_return

    def perform_rollouts(self, actions, policy_function, length=5, n=5):
        rollout_actions, rollout_returns = [], []
        for action in range(len(actions)):
            returns = [self.single_rollout(action, policy_function, length) for _ in range(n)]
            rollout_actions.append(actions[action])
            rollout_returns.append(np.mean(returns))
        return rollout_actions, rollout_returns


class _AfterState:
    """Entry of ``Tetris.afterstates``: the k-th non-terminal placement.  Features come
    from the batched kernel; the board is materialised on first access by playing the
    action on a scratch env."""

    def __init__(self, game, k):
        self._game, self._k = game, k
        self.terminal_state = False
        self._state = None

    @property
    def features(self):
        return self._game._after_feats[self._k]

    def get_features(self, direct_by=None, **_):
        f = self.features
        return f if direct_by is None else f * direct_by

    def _materialise(self):
        if self._state is None:
            g = self._game
            sc = g._get_scratch()
            g._push_state(sc, g._after_parent[0], g._after_parent[1])
            sc._cursor.zero_()
            obs, _, _, lines = sc.step(torch.tensor([self._k], dtype=torch.int32))
            rep = sc.boards()[0].cpu().numpy().astype(np.int_)
            self._state = State(rep, features=obs[0].cpu().numpy().copy(), n_cleared_lines=int(lines[0]))
        return self._state

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._materialise(), name)
