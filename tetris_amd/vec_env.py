"""VecTetris: a batch of placement-level Tetris envs resident in HBM.

The batched counterpart of the reference's ``game.Tetris`` (game.py:8-100):
``reset() / get_after_states() / step(actions)`` for B envs in lockstep.  All
state lives in PyTorch-ROCm tensors; every method enqueues one HIP kernel of
``libtetris_hip.so`` on the current stream through the C-ABI
(include/tetris_hip.h).  PyTorch is plumbing (memory + streams) only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .tetromino import CATALOGUE, Tetromino, resolve_pieces


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class VecTetris:
    """B independent envs with the semantics of game.Tetris.

    Parameters mirror ``Tetris(num_columns, num_rows, feature_directions=None)``
    (game.py:21-23) plus the batch knobs:

    batch_size   number of envs
    device       a CUDA (ROCm) device
    pieces       'default' (game.py:38-39), 'standard7' (game.py:41-47) or names
    auto_reset   reset finished envs inside the step kernel (game.py:53-63)
    seed         seed of the counter-based device bag
    piece_stream optional uint8 [L, B]: replay these list indices instead of the
                 device bag (parity runs against a recorded NumPy stream)
    numpy_seeds  optional [B] ints: env i draws the pieces of a reference game seeded with
                 ``np.random.seed(numpy_seeds[i])`` -- the reference's sampler on NumPy's MT19937
                 stream, generated on the device (``stream_len`` draws per env, replay mode)
    env_offset   global index of env 0 (shards of one logical batch draw the
                 same pieces as the unsharded batch)
    compute_obs  False: step() skips the BCTS observation (obs stays zero); for agents that
                 already hold get_after_states() features, whose row `action` is that observation
    afterstate_layout  storage of the get_after_states matrices: "env_major" (contiguous
                 [B, a_max, 8], default, fastest on MI355X) or "action_major"
                 ([a_max, B, 8] storage returned as a [B, a_max, 8] view)


    ``step`` returns views of buffers that the next ``step`` overwrites.
    """

    def __init__(self, num_columns, num_rows, batch_size, device="cuda", pieces="default", auto_reset=False,
                 seed=0, feature_directions=None, piece_stream=None, env_offset=0, afterstate_layout="env_major",
                 compute_obs=True, numpy_seeds=None, stream_len=4096):
        self._lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != self._lib.device_type:
            raise ValueError("VecTetris needs a %s device (got %s): the env only exists as HIP kernels"
                             % (self._lib.device_type, self.device))
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_columns, self.num_rows, self.batch_size = int(num_columns), int(num_rows), int(batch_size)
        self.piece_names = resolve_pieces(pieces)
        self.tetrominos = [Tetromino(n, i, self.num_columns) for i, n in enumerate(self.piece_names)]
        self._generation = 0  # bumped whenever something a bound step call / captured graph froze changes
        self._step_call = None
        self._auto_reset = bool(auto_reset)
        self._seed = int(seed)
        self._env_offset = int(env_offset)
        self.feature_directions = feature_directions
        if afterstate_layout not in ("action_major", "env_major"):
            raise ValueError("afterstate_layout must be 'action_major' or 'env_major'")
        self.afterstate_layout = afterstate_layout
        self._compute_obs = bool(compute_obs)
        self.loss_reward, self.timestep_reward = -100, -1  # game.py:34-35 (baked into the kernel)

        ids = (ctypes.c_int32 * len(self.piece_names))(*[CATALOGUE.index(n) for n in self.piece_names])
        dirs = None
        if feature_directions is not None:
            fd = np.asarray(feature_directions, dtype=np.float32)
            if fd.shape != (8,):
                raise ValueError("feature_directions must have 8 entries")
            dirs = (ctypes.c_float * 8)(*fd.tolist())
        self.desc = _lib.TetrisDesc()
        self._lib.check(self._lib.desc_init(ctypes.byref(self.desc), self.num_columns, self.num_rows, ids,
                                            len(self.piece_names), dirs), "tetris_hip_desc_init")
        if self.batch_size <= 0:
            raise ValueError("batch_size must be positive")
        self.a_max = int(self.desc.a_max)
        self.stored_rows = self.num_rows + 4
        self.word_dtype = torch.int32 if self.desc.word_bytes == 4 else torch.int64

        B, dev = self.batch_size, self.device
        with torch.device(dev):
            # tile-major column bitboards [ceil(B/64), n_planes, 64], opaque: one plane per column, or
            # bit-packed four columns to three words when the stored rows fit three quarters of the word
            # (tetris_hip_n_planes / tetris_hip_board_words)
            self.n_planes = int(self._lib.n_planes(ctypes.byref(self.desc)))
            if self.n_planes <= 0:
                self._lib.check(self.n_planes, "tetris_hip_n_planes")
            self.n_tiles = (B + 63) // 64
            assert int(self._lib.board_words(ctypes.byref(self.desc), B)) == self.n_tiles * self.n_planes * 64
            self.cols = torch.zeros((self.n_tiles, self.n_planes, 64), dtype=self.word_dtype)
            self.meta = torch.zeros(B, dtype=torch.int64)
            self._obs_buf = torch.zeros((B, 8), dtype=torch.float32)
            self._reward_buf = torch.zeros(B, dtype=torch.int32)
            self._done_buf = torch.zeros(B, dtype=torch.uint8)
            self._lines_buf = torch.zeros(B, dtype=torch.uint8)
            self.n_valid = torch.zeros(B, dtype=torch.uint8)
            self.piece = torch.zeros(B, dtype=torch.uint8)
            self.status = torch.zeros(int(self._lib.status_words(B)), dtype=torch.int32)  # [n_waves, 4]
            self._action_buf = torch.zeros(B, dtype=torch.int32)  # actions drawn by the built-in policy
        self._own_views()
        self._stream = None
        self._cursor = None
        if numpy_seeds is not None:
            if piece_stream is not None:
                raise ValueError("give either piece_stream or numpy_seeds")
            piece_stream = self.numpy_piece_stream(numpy_seeds, len(self.piece_names), stream_len, dev)
        if piece_stream is not None:
            ps = torch.as_tensor(piece_stream, dtype=torch.uint8)
            if ps.dim() != 2 or ps.shape[1] != B:
                raise ValueError("piece_stream must be [L, batch_size]")
            self._stream = ps.to(dev).contiguous()
            self._cursor = torch.zeros(B, dtype=torch.int32, device=dev)
        self._feats = None
        self._feats_all = None
        self._n_all = None
        self.step_idx = 0
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) if self.device.type == "cuda" else None
        self.reset(init_bag=True)

    MAX_REPLAY_STREAM_BYTES = 1 << 30  # one byte per (draw, env)

    @staticmethod
    def numpy_piece_stream(seeds, n_pieces, length, device="cuda"):
        """uint8 [length, B]: column i = what the reference's ``TetrominoSampler`` (tetromino.py:12-22)
        returns call after call in a process seeded ``np.random.seed(seeds[i])``, computed on the device
        (``tetris_hip_numpy_bag_stream``: MT19937 + NumPy's legacy permutation)."""
        lib = _lib.load()
        device = torch.device(device)
        n_bytes = int(length) * int(np.asarray(seeds).size)
        if n_bytes > VecTetris.MAX_REPLAY_STREAM_BYTES:
            raise ValueError("a replay stream of %d draws x %d envs is %.1f GiB (limit %.1f GiB): NumPy-exact replay "
                             "is a set-up mode for moderate batches -- pass a shorter stream_len, or use the "
                             "counter-based device bag (no numpy_seeds) for large batches"
                             % (int(length), int(np.asarray(seeds).size), n_bytes / 2.0**30,
                                VecTetris.MAX_REPLAY_STREAM_BYTES / 2.0**30))
        seeds32 = torch.from_numpy(np.asarray(seeds, dtype=np.uint32).view(np.int32).copy()).to(device)
        out = torch.empty((int(length), seeds32.numel()), dtype=torch.uint8, device=device)
        stream = None
        if device.type == "cuda":
            stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        lib.check(lib.numpy_bag_stream(_ptr(seeds32), int(n_pieces), int(length), _ptr(out), seeds32.numel(), stream),
                  "tetris_hip_numpy_bag_stream")
        return out

    # seed / auto_reset / compute_obs / env_offset are baked into the bound step call (and into every HIP
    # graph captured from it): assigning one drops the bound call, so that step(), step_many(), reset() and
    # random_actions() keep agreeing, and invalidates captured graphs (StepGraph.replay raises).
    def _frozen(name):  # noqa: N805
        def get(self):
            return getattr(self, "_" + name)

        def set_(self, value):
            setattr(self, "_" + name, type(getattr(self, "_" + name))(value))
            self._invalidate_bound_call()
        return property(get, set_)

    seed = _frozen("seed")
    auto_reset = _frozen("auto_reset")
    compute_obs = _frozen("compute_obs")
    env_offset = _frozen("env_offset")
    del _frozen

    def _invalidate_bound_call(self):
        self._step_call = None
        self._generation += 1

    # -- plumbing -----------------------------------------------------------------
    def _own_views(self):
        """Point the per-step attributes (obs, reward, done, lines, action) at this env's own buffers;
        after step_many they are views of the last row of its trajectory buffers instead."""
        self.obs, self.reward, self._done, self.lines = self._obs_buf, self._reward_buf, self._done_buf, self._lines_buf
        self.action = self._action_buf
        self.done = self._done.view(torch.bool)
        self._views_own = True

    def _hip_stream(self):
        if self.device.type == "cuda":
            return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    def _stream_args(self):
        if self._stream is None:
            return None, None, 0
        return _ptr(self._stream), _ptr(self._cursor), int(self._stream.shape[0])

    # -- Tetris.reset (game.py:53-63) -----------------------------------------------
    def reset(self, mask=None, init_bag=False):
        """Reset every env (or those where ``mask`` is true).  The bag is kept
        unless ``init_bag`` (the reference bag survives reset: game.py:50)."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.batch_size,):
                raise ValueError("mask must be [batch_size]")
        s, c, n = self._stream_args()
        rc = self._lib.reset(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), _ptr(m), _ptr(self.piece),
                             _ptr(self.n_valid), s, c, n, _ptr(self.status), int(bool(init_bag)), self.seed,
                             self.step_idx, self.env_offset, self.batch_size, self._hip_stream())
        self._lib.check(rc, "tetris_hip_reset")

    # -- Tetris.get_after_states (game.py:67-80) --------------------------------------
    def get_after_states(self, include_terminal=False):
        """BCTS features of every placement of the current piece.

        Returns ``(features [B, a_max, 8] float32, n_valid [B] uint8)``; row k of
        env i is its k-th non-terminal placement (= action k), rows >= n_valid
        are zero.  With ``include_terminal`` also ``(features_all, n_all)`` in
        raw enumeration order (game.py:74-78)."""
        B, A = self.batch_size, self.a_max
        am = self.afterstate_layout == "action_major"
        shape = (A, B, 8) if am else (B, A, 8)
        env_stride, row_stride = (8, B * 8) if am else (A * 8, 8)
        if self._feats is None:
            self._feats = torch.empty(shape, dtype=torch.float32, device=self.device)
            self._nv_after = torch.empty(B, dtype=torch.uint8, device=self.device)
        fa = na = None
        if include_terminal:
            if self._feats_all is None:
                self._feats_all = torch.empty(shape, dtype=torch.float32, device=self.device)
                self._n_all = torch.empty(B, dtype=torch.uint8, device=self.device)
            fa, na = self._feats_all, self._n_all
        rc = self._lib.afterstates(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), _ptr(self._feats),
                                   _ptr(self._nv_after), _ptr(fa), _ptr(na), env_stride, row_stride, B,
                                   self._hip_stream())
        self._lib.check(rc, "tetris_hip_afterstates")
        view = (lambda t: t.permute(1, 0, 2)) if am else (lambda t: t)
        if include_terminal:
            return view(self._feats), self._nv_after, view(fa), na
        return view(self._feats), self._nv_after

    # -- Tetris.step (game.py:82-92) ------------------------------------------------------
    def gather_payload(self):
        """Buffers for one done/reset gather (``step(..., gather=payload)``): ``done_bits`` uint8
        [8 * ceil(B / 64)] -- the done bitmask, bit i % 8 of byte i // 8 -- and ``counters`` int32
        [n_waves, 4] -- the per-wave (invalid, episodes, lines, steps) slots as of that step.  The step
        kernel writes both from its epilogue; nothing else touches them, so a collective can read them
        on another stream (``tetris_hip_stream_link``) while the env keeps stepping."""
        return dict(done_bits=torch.zeros(((self.batch_size + 63) // 64) * 8, dtype=torch.uint8, device=self.device),
                    counters=torch.zeros_like(self.status).view(-1, 4))

    def step(self, action=None, gather=None):
        """``action`` [B] = index into each env's non-terminal placements; ``None`` = every env
        plays a uniform random valid action drawn inside the kernel (readable afterwards in
        ``self.action``; identical to ``step(random_actions())``).  ``gather``: a
        :meth:`gather_payload` this step fills in (done bitmask + counter snapshot).

        Returns ``(obs [B,8] f32, reward [B] i32, done [B] bool, lines [B] u8)``.
        Out-of-range actions leave that env untouched and are counted; call
        :meth:`check` to turn them into the reference's IndexError."""
        if not self._views_own:
            self._own_views()
        a = None
        if action is not None:
            a = torch.as_tensor(action, device=self.device)
            if a.shape != (self.batch_size,):
                raise ValueError("action must be [batch_size]")
            if a.dtype != torch.int32 or not a.is_contiguous():
                a = a.to(torch.int32).contiguous()
        # bound call (tetris_hip_step_call_*): pointers, geometry and the placement table are prepared once
        call = self._step_call
        if call is None:
            call = self._bind_step_call()
        if self._raw_stream is not None:
            stream = self._raw_stream(self.device.index)
        else:
            stream = self._hip_stream()
        if gather is None:
            rc = self._lib.step_call_run(call, None if a is None else a.data_ptr(), self.step_idx, stream)
        else:
            rc = self._lib.step_call_run_gather(call, None if a is None else a.data_ptr(), self.step_idx,
                                                gather["done_bits"].data_ptr(), gather["counters"].data_ptr(), stream)
        if rc:
            self._lib.check(rc, "tetris_hip_step_call_run")
        self.step_idx += 1
        return self.obs, self.reward, self.done, self.lines

    # -- HIP graph of K steps -------------------------------------------------------------------
    def capture_steps(self, n_steps, action_fn=None):
        """Capture ``n_steps`` consecutive steps in ONE HIP graph and return it (``StepGraph``); each
        ``replay()`` then advances the env by ``n_steps`` steps with a single graph launch.  For small
        batches, where one step kernel (5-8 us) is about as long as its host enqueue, this takes the
        host off the critical path.  ``action_fn(env, k)`` (optional) is called at capture time before
        step k and must enqueue, on the current stream, whatever produces the ``[B]`` int32 action
        tensor it returns (it is captured too); without it every env plays the built-in uniform
        random policy.  The step index lives in device memory (``tetris_hip_step_call_run_counted``), so
        replays are bit-identical to the same number of ``step()`` calls."""
        return StepGraph(self, int(n_steps), action_fn)

    def _bind_step_call(self):
        size = int(self._lib.step_call_size())
        buf = ctypes.create_string_buffer(size + 16)
        addr = (ctypes.addressof(buf) + 15) & ~15
        s, c, n = self._stream_args()
        rc = self._lib.step_call_init(addr, ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta),
                                      _ptr(self._action_buf), s, c, n,
                                      _ptr(self._obs_buf) if self.compute_obs else None, _ptr(self._reward_buf),
                                      _ptr(self._done_buf), _ptr(self._lines_buf), _ptr(self.n_valid), _ptr(self.piece),
                                      _ptr(self.status), int(self.auto_reset), self.seed, self.env_offset,
                                      self.batch_size)
        self._lib.check(rc, "tetris_hip_step_call_init")
        self._step_call_buf = buf  # keeps the memory alive
        self._step_call = addr
        return addr

    def step_many(self, n_steps, policy="random", weights=None, out=None):
        """``n_steps`` consecutive steps in ONE kernel launch with an in-kernel policy ("random":
        uniform valid action; "greedy": linear fitness on ``weights``): boards stay in registers
        between steps, every step's outputs are written.  Returns a dict of trajectory tensors
        ``obs [K,B,8] f32, reward [K,B] i32, done [K,B] bool, lines [K,B] u8, action [K,B] i32,
        n_valid [K,B] u8, piece [K,B] u8``; with policy "random" it is bit-identical to
        ``n_steps`` calls of ``step()``.  Pass the previous result as ``out`` to reuse buffers."""
        K, B, dev = int(n_steps), self.batch_size, self.device
        pol = {"random": 0, "greedy": 1}[policy]
        if self._stream is not None:
            raise ValueError("step_many draws pieces from the device bag (no replay stream)")
        if out is None or out["reward"].shape[0] != K:
            out = dict(obs=torch.zeros((K, B, 8), dtype=torch.float32, device=dev),
                       reward=torch.empty((K, B), dtype=torch.int32, device=dev),
                       _done=torch.empty((K, B), dtype=torch.uint8, device=dev),
                       lines=torch.empty((K, B), dtype=torch.uint8, device=dev),
                       action=torch.empty((K, B), dtype=torch.int32, device=dev),
                       n_valid=torch.empty((K, B), dtype=torch.uint8, device=dev),
                       piece=torch.empty((K, B), dtype=torch.uint8, device=dev))
            out["done"] = out["_done"].view(torch.bool)
        w = (ctypes.c_float * 8)(*(self.BCTS_WEIGHTS if weights is None else [float(x) for x in weights]))
        rc = self._lib.step_many(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), K, pol, w,
                                 _ptr(out["action"]), _ptr(out["obs"]) if self.compute_obs else None,
                                 _ptr(out["reward"]), _ptr(out["_done"]), _ptr(out["lines"]), _ptr(out["n_valid"]),
                                 _ptr(out["piece"]), _ptr(self.status), int(self.auto_reset), self.seed,
                                 self.step_idx, self.env_offset, B, self._hip_stream())
        self._lib.check(rc, "tetris_hip_step_many")
        self.step_idx += K
        # the per-step attributes keep describing the latest step: n_valid / piece (inputs of the next
        # reset / random_actions) are copied, the outputs become views of the last trajectory row
        self.n_valid.copy_(out["n_valid"][K - 1])
        self.piece.copy_(out["piece"][K - 1])
        if self.compute_obs:
            self.obs = out["obs"][K - 1]
        self.reward, self._done, self.lines = out["reward"][K - 1], out["_done"][K - 1], out["lines"][K - 1]
        self.action = out["action"][K - 1]
        self.done = self._done.view(torch.bool)
        self._views_own = False
        return out

    BCTS_WEIGHTS = (-24.04, -19.77, -13.08, -12.63, -10.49, -9.22, 6.6, -1.61)  # game.py:111-118

    def greedy_actions(self, weights=None, include_fitness=False):
        """Tetris.get_best_policy / fitness (game.py:102-120) for every env, without materialising
        the feature matrix: returns ``(best_action int32 [B], best_value float32 [B])`` -- the first
        non-terminal action of maximal linear fitness (-1 when the env has none) -- and with
        ``include_fitness`` also ``fitness_all float32 [B, a_max]`` over ALL placements in raw order
        (terminal included, the domain of the reference's policy vector)."""
        w = (ctypes.c_float * 8)(*(self.BCTS_WEIGHTS if weights is None else [float(x) for x in weights]))
        B = self.batch_size
        if not hasattr(self, "_best_action"):
            self._best_action = torch.empty(B, dtype=torch.int32, device=self.device)
            self._best_value = torch.empty(B, dtype=torch.float32, device=self.device)
            self._fitness_all = None
        fa = None
        if include_fitness:
            if self._fitness_all is None:
                self._fitness_all = torch.empty((B, self.a_max), dtype=torch.float32, device=self.device)
            fa = self._fitness_all
        rc = self._lib.policy_greedy(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), w,
                                     _ptr(self._best_action), _ptr(self._best_value), _ptr(fa), B,
                                     self._hip_stream())
        self._lib.check(rc, "tetris_hip_policy_greedy")
        if include_fitness:
            return self._best_action, self._best_value, fa
        return self._best_action, self._best_value

    def rollouts(self, length=5, n=5, policy="random", weights=None, pieces=None):
        """Tetris.perform_rollouts (game.py:150-160) for every env and every valid first action:
        ``returns float64 [B, a_max]`` = mean rollout return over ``n`` rollouts of ``length`` steps
        (NaN where the action does not exist).  ``policy`` = "random" or "greedy" (linear fitness
        on ``weights``, default the BCTS weights of game.py:111-118).  ``pieces`` (optional uint8
        ``[B, a_max, n, length]``): the list index every step of every rollout draws -- a recorded run
        of the reference's sampler; without it every rollout draws from its own fork of the env's bag.
        The envs are not modified."""
        pol = {"random": 0, "greedy": 1}[policy]
        w = (ctypes.c_float * 8)(*(self.BCTS_WEIGHTS if weights is None else [float(x) for x in weights]))
        out = torch.empty((self.batch_size, self.a_max), dtype=torch.float64, device=self.device)
        fed = None
        if pieces is not None:
            fed = torch.as_tensor(pieces, dtype=torch.uint8).to(self.device).contiguous()
            if tuple(fed.shape) != (self.batch_size, self.a_max, int(n), int(length)):
                raise ValueError("pieces must be [batch_size, a_max, n, length]")
            if int(fed.max()) >= len(self.piece_names):
                raise ValueError("pieces holds an index outside the piece list")
        rc = self._lib.rollouts(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), _ptr(out), int(length),
                                int(n), pol, w, _ptr(fed), self.seed, self.step_idx, self.env_offset, self.batch_size,
                                self._hip_stream())
        self._lib.check(rc, "tetris_hip_rollouts")
        return out

    def random_actions(self, out=None):
        """Uniform random valid action per env (the random-rollout policy)."""
        if not self._views_own:
            self._own_views()
        out = self.action if out is None else out
        rc = self._lib.policy_random(_ptr(self.n_valid), _ptr(out), self.seed, self.step_idx, self.env_offset,
                                     self.batch_size, self._hip_stream())
        self._lib.check(rc, "tetris_hip_policy_random")
        return out

    # -- inspection ---------------------------------------------------------------------------
    def boards(self):
        """State.representation for every env: int8 [B, R+4, C], row 0 = bottom."""
        cells = torch.empty((self.batch_size, self.stored_rows, self.num_columns), dtype=torch.int8,
                            device=self.device)
        rc = self._lib.decode(ctypes.byref(self.desc), _ptr(self.cols), _ptr(cells), None, self.batch_size,
                              self._hip_stream())
        self._lib.check(rc, "tetris_hip_decode")
        return cells

    def heights(self):
        """State.lowest_free_rows for every env: int32 [B, C]."""
        h = torch.empty((self.batch_size, self.num_columns), dtype=torch.int32, device=self.device)
        rc = self._lib.decode(ctypes.byref(self.desc), _ptr(self.cols), None, _ptr(h), self.batch_size,
                              self._hip_stream())
        self._lib.check(rc, "tetris_hip_decode")
        return h

    def set_boards(self, cells, piece=None):
        """Overwrite the boards (int8 [B, R+4, C]) and optionally the current
        pieces (list indices [B]); the valid masks are recomputed.

        The boards must be states a game can be in: a board with a cell in the overflow rows
        (row >= num_rows) is a terminal State (state.py:33-36), which the reference never steps
        from (game.py:69 keeps non-terminal afterstates only); the kernels' placement mask and
        10-row feature tables assume there is none, so such a board is refused here."""
        cells = torch.as_tensor(cells, device=self.device).to(torch.int8).contiguous()
        if cells.shape != (self.batch_size, self.stored_rows, self.num_columns):
            raise ValueError("cells must be [B, num_rows + 4, num_columns]")
        if bool((cells[:, self.num_rows:, :] != 0).any()):
            raise ValueError("boards with cells in the overflow rows (row >= num_rows) are terminal states "
                             "and cannot be set as current boards")
        rc = self._lib.encode(ctypes.byref(self.desc), _ptr(cells), _ptr(self.cols), self.batch_size,
                              self._hip_stream())
        self._lib.check(rc, "tetris_hip_encode")
        if piece is not None:
            p = torch.as_tensor(piece, device=self.device).to(torch.int64)
            if p.shape != (self.batch_size,):
                raise ValueError("piece must be [batch_size]")
            keep = ~(torch.tensor(15, dtype=torch.int64, device=self.device) << 48)
            self.meta.copy_((self.meta & keep) | (p << 48))
            self.piece.copy_(p.to(torch.uint8))
        self.refresh()

    def columns(self):
        """The boards as one bitboard per column, int64 [C, B] (bit r = cell (row r, column c)):
        the unpacked view of ``cols`` (debugging / tests; the kernels work on ``cols``)."""
        C, W = self.num_columns, 8 * self.desc.word_bytes
        planes = self.cols.permute(1, 0, 2).reshape(self.n_planes, -1)[:, :self.batch_size].to(torch.int64)
        if W == 32:
            planes = planes & 0xFFFFFFFF
        if self.n_planes == C:
            return planes
        F = W * 3 // 4
        full = (1 << W) - 1 if W < 64 else -1
        mask = (1 << F) - 1
        out = []

        def shr(x, n):  # logical shift right of a W-bit word held in int64
            if W == 64:
                return (x >> n) & ((1 << (64 - n)) - 1) if n else x
            return x >> n

        for c in range(C):
            b, k = 3 * (c // 4), c % 4
            if k == 0:
                v = planes[b]
            elif k == 1:
                v = shr(planes[b], F) | (planes[b + 1] << (W - F))
            elif k == 2:
                v = shr(planes[b + 1], 2 * F - W) | (planes[b + 2] << (2 * W - 2 * F))
            else:
                v = shr(planes[b + 2], 3 * F - 2 * W)
            out.append(v & mask)
        return torch.stack(out)

    def refresh(self):
        """Recompute valid masks / n_valid from (cols, piece) after a manual edit."""
        rc = self._lib.refresh(ctypes.byref(self.desc), _ptr(self.cols), _ptr(self.meta), _ptr(self.n_valid),
                               self.batch_size, self._hip_stream())
        self._lib.check(rc, "tetris_hip_refresh")

    def totals(self):
        """int64 [4] device tensor: (invalid, episodes, lines, steps) summed over the per-wave slots."""
        return (self.status.view(-1, 4).to(torch.int64) & 0xFFFFFFFF).sum(dim=0)

    def stats(self):
        """Counters accumulated by the step kernel (synchronises)."""
        s = self.totals().cpu().numpy()
        return dict(invalid=int(s[0]), episodes=int(s[1]), lines=int(s[2]), steps=int(s[3]))

    def check(self):
        """Raise IndexError if any step so far received an out-of-range action
        (game.py:83 raises it immediately; the batch path reports lazily)."""
        n = self.stats()["invalid"]
        if n:
            raise IndexError("%d out-of-range actions were passed to step()%s" % (
                n, " (or the replay piece_stream ran out of rows)" if self._stream is not None else ""))

    # -- checkpoint / snapshot --------------------------------------------------------------------
    STATE_FORMAT = 3  # 1: plane-major cols, C-bit mask fields; 2: tile-major cols, 12-bit mask fields (unversioned)

    def state_dict(self):
        d = dict(format=self.STATE_FORMAT, abi=_lib.ABI_VERSION, cols=self.cols.clone(), meta=self.meta.clone(), n_valid=self.n_valid.clone(),
                 piece=self.piece.clone(), status=self.status.clone(), step_idx=self.step_idx, seed=self.seed,
                 num_columns=self.num_columns, num_rows=self.num_rows, pieces=list(self.piece_names))
        if self._cursor is not None:
            d["cursor"] = self._cursor.clone()
        return d

    def load_state_dict(self, d):
        if (d["num_columns"], d["num_rows"], list(d["pieces"])) != (self.num_columns, self.num_rows,
                                                                     list(self.piece_names)):
            raise ValueError("state_dict belongs to a different env configuration")
        if d.get("format") != self.STATE_FORMAT:
            raise ValueError("state_dict has storage format %r, this build reads format %d (the layout of `cols` / "
                             "`meta` changed between them; re-create the env and set_boards() from decoded cells)"
                             % (d.get("format"), self.STATE_FORMAT))
        for k in ("cols", "meta", "n_valid", "piece", "status"):
            if tuple(d[k].shape) != tuple(getattr(self, k).shape) or d[k].dtype != getattr(self, k).dtype:
                raise ValueError("state_dict[%r] is %s %s, this env holds %s %s (different batch size?)"
                                 % (k, tuple(d[k].shape), d[k].dtype, tuple(getattr(self, k).shape), getattr(self, k).dtype))
        for k in ("cols", "meta", "n_valid", "piece", "status"):
            getattr(self, k).copy_(d[k])
        if self._cursor is not None and "cursor" in d:
            self._cursor.copy_(d["cursor"])
        self.step_idx = int(d["step_idx"])
        self.seed = int(d["seed"])  # (drops the bound step call and invalidates captured graphs)


class StepGraph:
    """``n_steps`` steps of one VecTetris as a HIP graph (``VecTetris.capture_steps``)."""

    def __init__(self, env, n_steps, action_fn=None):
        if n_steps < 1:
            raise ValueError("n_steps must be >= 1")
        self.env, self.n_steps = env, n_steps
        lib = env._lib
        self._counter = torch.zeros(1, dtype=torch.int64, device=env.device)
        self._counter.fill_(env.step_idx)
        self._expected = env.step_idx
        call = env._step_call if env._step_call is not None else env._bind_step_call()
        self._generation = env._generation
        self._call_buf = env._step_call_buf  # the bound call's memory must outlive every launch enqueued from here

        def launch_stream():  # the stream being captured / the current stream, exactly as step() picks it
            if env._raw_stream is not None:
                return env._raw_stream(env.device.index)
            return env._hip_stream()

        def enqueue():
            if not env._views_own:
                env._own_views()
            for k in range(n_steps):
                a = None
                if action_fn is not None:
                    a = action_fn(env, k)
                    if a.dtype != torch.int32 or not a.is_contiguous() or a.shape != (env.batch_size,):
                        raise ValueError("action_fn must return a contiguous int32 [batch_size] tensor")
                    self._keep.append(a)
                rc = lib.step_call_run_counted(call, None if a is None else a.data_ptr(), self._counter.data_ptr(), k,
                                               launch_stream())
                lib.check(rc, "tetris_hip_step_call_run_counted")
            lib.check(lib.counter_add(self._counter.data_ptr(), n_steps, launch_stream()), "tetris_hip_counter_add")

        self._keep = []
        self._graph = None
        self._enqueue = enqueue
        if env.device.type == "cuda":
            torch.cuda.synchronize(env.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                enqueue()
            self._graph = g

    def replay(self):
        """Advance the env by ``n_steps`` steps (returns the per-step views of the last step)."""
        env = self.env
        if env._generation != self._generation:
            raise RuntimeError("this StepGraph was captured before the env's seed / auto_reset / compute_obs / "
                               "env_offset changed (or load_state_dict ran): its launches hold the old values -- "
                               "capture a new one with env.capture_steps()")
        if env.step_idx != self._expected:  # plain step() calls in between: re-seat the device counter
            self._counter.fill_(env.step_idx)
        if not env._views_own:
            env._own_views()
        if self._graph is not None:
            self._graph.replay()
        else:  # no graph support on this device type (CPU test harness): the same launches, eagerly
            self._enqueue()
        env.step_idx += self.n_steps
        self._expected = env.step_idx
        return env.obs, env.reward, env.done, env.lines
