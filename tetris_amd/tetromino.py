"""Piece catalogue and samplers (host side).

Mirrors the roles of /root/reference/tetromino.py: the nine piece classes
(:33-576) become catalogue entries whose placement tables live in the HIP
library (csrc/tetris_table.hpp); the bag sampler (:12-22) is kept verbatim in
behaviour because "same seed" in the reference means NumPy's global legacy
stream.
"""
import numpy as np

# catalogue id = class order of tetromino.py:33-576 (== TETRIS_* in include/tetris_hip.h)
CATALOGUE = ("Straight", "Square", "SnakeR", "ThreeLine", "ThreeL", "SnakeL", "T", "RCorner", "LCorner")

PIECE_SETS = {
    # game.py:38-39 -- the list that is live in the reference
    "default": ("ThreeL", "ThreeLine"),
    # game.py:41-47 -- the commented-out seven tetrominoes, in that order
    "standard7": ("Straight", "RCorner", "LCorner", "Square", "SnakeR", "SnakeL", "T"),
}

# (width, bottom offsets b, cell counts n) per orientation, grouped in the
# reference's loops; only used host-side to describe / draw placements.  The
# authoritative copy is csrc/tetris_table.hpp (tests check they agree).
ORIENTATIONS = {
    "Straight": (((1, (0,), (4,)),), ((4, (0, 0, 0, 0), (1, 1, 1, 1)),)),
    "Square": (((2, (0, 0), (2, 2)),),),
    "SnakeR": (((3, (0, 0, 1), (1, 2, 1)),), ((2, (1, 0), (2, 2)),)),
    "ThreeLine": (((1, (0,), (3,)),), ((3, (0, 0, 0), (1, 1, 1)),)),
    "ThreeL": (((2, (0, 0), (1, 2)), (2, (0, 1), (2, 1))), ((2, (1, 0), (1, 2)), (2, (0, 0), (2, 1)))),
    "SnakeL": (((3, (1, 0, 0), (1, 2, 1)),), ((2, (0, 1), (2, 2)),)),
    "T": (((3, (0, 0, 0), (1, 2, 1)), (3, (1, 0, 1), (1, 2, 1))), ((2, (1, 0), (1, 3)), (2, (0, 1), (3, 1)))),
    "RCorner": (((3, (0, 0, 0), (1, 1, 2)), (3, (0, 1, 1), (2, 1, 1))), ((2, (2, 0), (1, 3)), (2, (0, 0), (3, 1)))),
    "LCorner": (((3, (0, 0, 0), (2, 1, 1)), (3, (1, 1, 0), (1, 1, 2))), ((2, (0, 2), (3, 1)), (2, (0, 0), (1, 3)))),
}


def resolve_pieces(pieces):
    """'default' | 'standard7' | iterable of catalogue names -> tuple of names."""
    names = PIECE_SETS[pieces] if isinstance(pieces, str) else tuple(pieces)
    for n in names:
        if n not in CATALOGUE:
            raise ValueError("unknown piece %r (catalogue: %s)" % (n, ", ".join(CATALOGUE)))
    if not 1 <= len(names) <= 12:
        raise ValueError("a piece set holds 1..12 pieces")
    return tuple(names)


def n_placements(name, num_columns):
    """Raw placement count (all loops x columns x orientations)."""
    total = 0
    for loop in ORIENTATIONS[name]:
        w = loop[0][0]
        total += max(0, num_columns - w + 1) * len(loop)
    return total


def placement_of_slot(name, num_columns, slot):
    """Bit index of the valid mask (four 12-bit fields, field 2*loop + orientation at bit 12 * field,
    bit = left column) -> (loop, column, orientation index)."""
    k, c = divmod(slot, 12)
    return k >> 1, c, k & 1


_B = "██"
# what `print(env.current_tetromino)` shows upstream (tetromino.py __repr__ of each class; ThreeLine's is
# misspelt `__repr` there, so upstream prints the default object repr for it -- here its cells are drawn)
_UPSTREAM_REPR = {
    "Straight": "\n" + " ".join([_B] * 4),
    "Square": "\n%s %s \n%s %s" % (_B, _B, _B, _B),
    "SnakeR": "\n   %s %s \n%s %s" % (_B, _B, _B, _B),
    "ThreeL": " %s %s \n%s%s" % (_B, _B, " " * 19, _B),
    "SnakeL": "\n%s %s \n   %s %s" % (_B, _B, _B, _B),
    "T": "\n   %s\n%s %s %s" % (_B, _B, _B, _B),
    "RCorner": "\n%s %s %s\n%s" % (_B, _B, _B, _B),
    "LCorner": "\n%s %s %s\n      %s" % (_B, _B, _B, _B),
}


class Tetromino:
    """One piece of the set; the stand-in for the reference's piece objects
    (``env.current_tetromino``).  ``tet_ind`` is the index in the env's piece
    list (the reference only defines it on ThreeLine / ThreeL, reversed:
    tetromino.py:160,205 -- not reproduced)."""

    def __init__(self, name, list_index, num_columns):
        self.name = name
        self.catalogue_id = CATALOGUE.index(name)
        self.list_index = list_index
        self.num_columns = num_columns

    def cells(self, loop=None, orientation=0):
        loops = ORIENTATIONS[self.name]
        if loop is None:
            loop = len(loops) - 1
        w, b, n = loops[loop][orientation]
        return [(b[j] + k, j) for j in range(w) for k in range(n[j])]

    def __repr__(self):
        # the text upstream prints for this piece (tests/golden/g8_render.npz), quirks included
        if self.name in _UPSTREAM_REPR:
            return _UPSTREAM_REPR[self.name]
        cells = self.cells()
        hgt = max(r for r, _ in cells) + 1
        wid = max(c for _, c in cells) + 1
        rows = []
        for r in range(hgt - 1, -1, -1):
            rows.append(" ".join("██" if (r, c) in cells else "  " for c in range(wid)).rstrip())
        return "\n" + "\n".join(rows)


class TetrominoSampler:
    """Bag sampler with the reference's exact semantics (tetromino.py:12-22):
    a fresh ``np.random.permutation`` whenever the bag is empty, drawn from
    NumPy's GLOBAL legacy stream, surviving ``Tetris.reset`` (game.py:50)."""

    def __init__(self, tetrominos):
        self.tetrominos = tetrominos
        self._bag = self._refill()  # tetromino.py:15: a bag exists from construction on

    def _refill(self):
        return [int(i) for i in np.random.permutation(len(self.tetrominos))]

    @property
    def current_batch(self):
        """Indices still in the bag, front first (the reference's attribute name)."""
        return np.array(self._bag, dtype=np.int64)

    def next_index(self):
        if not self._bag:  # tetromino.py:18-19
            self._bag = self._refill()
        return self._bag.pop(0)  # tetromino.py:20-21

    def next_tetromino(self):
        return self.tetrominos[self.next_index()]
