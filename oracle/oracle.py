"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module (it is the checker, never the product).
The C restatement it loads is documented in ``oracle/tetris_oracle.h``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtetris_oracle.so")

# catalogue ids: class order of /root/reference/tetromino.py:33-576
CATALOGUE = ["Straight", "Square", "SnakeR", "ThreeLine", "ThreeL", "SnakeL", "T", "RCorner", "LCorner"]
PIECE_SETS = {
    "default": ["ThreeL", "ThreeLine"],  # game.py:38-39
    "standard7": ["Straight", "RCorner", "LCorner", "Square", "SnakeR", "SnakeL", "T"],  # game.py:41-47
}


class OrcDesc(ctypes.Structure):
    _fields_ = [
        ("num_columns", ctypes.c_int32),
        ("num_rows", ctypes.c_int32),
        ("n_pieces", ctypes.c_int32),
        ("piece_ids", ctypes.c_int32 * 16),
    ]


def build(force=False):
    """Build the oracle if it is missing (or sources are newer and we are in the authoring tree)."""
    src = [os.path.join(_HERE, f) for f in ("tetris_oracle.c", "tetris_oracle.h")]
    stale = os.path.exists(_SO) and any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or not os.path.exists(_SO) or (stale and os.environ.get("TETRIS_ORACLE_NO_REBUILD") != "1"):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtetris_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_step_batch.restype = ctypes.c_int64
        _lib.orc_hash32.restype = ctypes.c_uint32
        _lib.orc_hash32.argtypes = [ctypes.c_uint64] * 3
        _lib.orc_mt_next.restype = ctypes.c_uint32
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def make_desc(num_columns, num_rows, pieces="default"):
    names = PIECE_SETS[pieces] if isinstance(pieces, str) else list(pieces)
    d = OrcDesc()
    d.num_columns, d.num_rows, d.n_pieces = num_columns, num_rows, len(names)
    for i, n in enumerate(names):
        d.piece_ids[i] = CATALOGUE.index(n)
    return d


def n_placements(piece_name, C):
    return lib().orc_n_placements(CATALOGUE.index(piece_name), C)


def a_max(desc):
    return max(lib().orc_n_placements(desc.piece_ids[i], desc.num_columns) for i in range(desc.n_pieces))


def placements(desc, cells, piece_name):
    """All raw placements of a catalogue piece on one board.

    cells: int8 [(R+4), C].  Returns dict of arrays with leading dim n.
    """
    C, rows = desc.num_columns, desc.num_rows + 4
    cells = np.ascontiguousarray(cells, dtype=np.int8).reshape(rows, C)
    n_max = 64
    out = dict(
        cells=np.zeros((n_max, rows, C), np.int8),
        heights=np.zeros((n_max, C), np.int32),
        n_cleared=np.zeros(n_max, np.int32),
        terminal=np.zeros(n_max, np.int32),
        anchor_row=np.zeros(n_max, np.int32),
        anchor_col=np.zeros(n_max, np.int32),
        feats=np.zeros((n_max, 8), np.float32),
    )
    n = lib().orc_placements_flat(
        ctypes.byref(desc), _p(cells), CATALOGUE.index(piece_name), _p(out["cells"]), _p(out["heights"]),
        _p(out["n_cleared"]), _p(out["terminal"]), _p(out["anchor_row"]), _p(out["anchor_col"]), _p(out["feats"]))
    return {k: v[:n] for k, v in out.items()}


def board_features(desc, cells):
    cells = np.ascontiguousarray(cells, dtype=np.int8)
    f = np.zeros(8, np.float32)
    lib().orc_board_features_flat(ctypes.byref(desc), _p(cells), _p(f))
    return f


class OracleVecEnv:
    """Batch of independent oracle envs with the build's sampler modes.

    Mirrors tetris_amd.VecTetris so parity tests can drive both in lockstep.
    """

    def __init__(self, num_columns, num_rows, batch_size, pieces="default", auto_reset=False, seed=0,
                 piece_stream=None, env_offset=0, nthreads=1):
        self.desc = make_desc(num_columns, num_rows, pieces)
        self.C, self.R, self.B = num_columns, num_rows, batch_size
        self.rows = num_rows + 4
        self.auto_reset, self.seed, self.env_offset, self.nthreads = int(auto_reset), seed, env_offset, nthreads
        self.cells = np.zeros((batch_size, self.rows, num_columns), np.int8)
        self.piece = np.zeros(batch_size, np.int32)
        self.bag = np.zeros(batch_size, np.uint16)
        self.n_valid = np.zeros(batch_size, np.uint8)
        self.stream = None if piece_stream is None else np.ascontiguousarray(piece_stream, dtype=np.uint8)
        self.cursor = np.zeros(batch_size, np.int32)
        self.step_idx = 0
        self.obs = np.zeros((batch_size, 8), np.float32)
        self.reward = np.zeros(batch_size, np.int32)
        self.done = np.zeros(batch_size, np.uint8)
        self.lines = np.zeros(batch_size, np.uint8)
        self.invalid = np.zeros(batch_size, np.uint8)
        self.action = np.zeros(batch_size, np.int32)
        self.reset(init_bag=True)

    def reset(self, init_bag=False):
        slen = 0 if self.stream is None else self.stream.shape[0]
        lib().orc_reset_batch(
            ctypes.byref(self.desc), _p(self.cells), _p(self.piece), _p(self.bag), _p(self.stream),
            _p(self.cursor), ctypes.c_int64(slen), _p(self.n_valid), int(init_bag), ctypes.c_uint64(self.seed),
            ctypes.c_uint64(self.step_idx), ctypes.c_int64(self.env_offset), ctypes.c_int64(self.B))

    def step(self, action=None):
        """action None = the build's uniform random policy (recorded in self.action)."""
        if action is not None:
            action = np.ascontiguousarray(action, dtype=np.int32)
        slen = 0 if self.stream is None else self.stream.shape[0]
        n_bad = lib().orc_step_batch(
            ctypes.byref(self.desc), _p(self.cells), _p(self.piece), _p(self.bag), _p(action), _p(self.action),
            _p(self.stream),
            _p(self.cursor), ctypes.c_int64(slen), _p(self.obs), _p(self.reward), _p(self.done), _p(self.lines),
            _p(self.n_valid), _p(self.invalid), self.auto_reset, ctypes.c_uint64(self.seed),
            ctypes.c_uint64(self.step_idx), ctypes.c_int64(self.env_offset), ctypes.c_int64(self.B),
            self.nthreads)
        self.step_idx += 1
        return self.obs, self.reward, self.done, self.lines, n_bad

    def rollouts(self, length=5, n=5, policy="random", weights=(-24.04, -19.77, -13.08, -12.63, -10.49, -9.22, 6.6,
                                                                  -1.61)):
        am = a_max(self.desc)
        out = np.zeros((self.B, am), np.float64)
        w = np.asarray(weights, np.float32)
        lib().orc_rollouts_batch(
            ctypes.byref(self.desc), _p(self.cells), _p(self.piece), _p(self.bag), _p(out), am, int(length), int(n),
            {"random": 0, "greedy": 1}[policy], _p(w), ctypes.c_uint64(self.seed), ctypes.c_uint64(self.step_idx),
            ctypes.c_int64(self.env_offset), ctypes.c_int64(self.B), self.nthreads)
        return out

    def afterstates(self, include_terminal=False):
        am = a_max(self.desc)
        feats = np.zeros((self.B, am, 8), np.float32)
        nv = np.zeros(self.B, np.uint8)
        fall = np.zeros((self.B, am, 8), np.float32) if include_terminal else None
        nall = np.zeros(self.B, np.uint8) if include_terminal else None
        lib().orc_afterstates_batch(
            ctypes.byref(self.desc), _p(self.cells), _p(self.piece), am, _p(feats), _p(nv), _p(fall), _p(nall),
            ctypes.c_int64(self.B), self.nthreads)
        return feats, nv, fall, nall


# ---- NumPy legacy MT19937 restatement ---------------------------------------

class NumpyLegacyRNG:
    """np.random.seed(s) / np.random.permutation(n) on the oracle's own MT19937."""

    def __init__(self, seed):
        self.state = np.zeros(625, np.uint32)
        lib().orc_mt_seed(_p(self.state), ctypes.c_uint32(seed))

    def permutation(self, n):
        out = np.zeros(n, np.int32)
        lib().orc_np_permutation(_p(self.state), n, _p(out))
        return out


class BagSampler:
    """tetromino.py:12-22 on top of NumpyLegacyRNG (bag survives reset)."""

    def __init__(self, rng, n_pieces):
        self.rng, self.n = rng, n_pieces
        self.batch = list(rng.permutation(n_pieces))  # tetromino.py:15

    def next(self):
        if len(self.batch) == 0:  # tetromino.py:18-19
            self.batch = list(self.rng.permutation(self.n))
        return int(self.batch.pop(0))  # tetromino.py:20-21


# ---- codec between the reference layout and the product's column bitboards ---

def cells_to_cols(cells):
    """int8 [..., rows, C] -> uint64 [..., C] (bit r of word c = cell (r, c))."""
    cells = np.asarray(cells).astype(np.uint64)
    rows = cells.shape[-2]
    w = (np.uint64(1) << np.arange(rows, dtype=np.uint64))[:, None]
    return (cells * w).sum(axis=-2).astype(np.uint64)


def cols_to_cells(cols, rows):
    cols = np.asarray(cols).astype(np.uint64)
    r = np.arange(rows, dtype=np.uint64)[:, None]
    return ((cols[..., None, :] >> r) & np.uint64(1)).astype(np.int8)
