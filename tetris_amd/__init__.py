"""tetris_amd -- MI355X-native vectorised placement-level Tetris.

Drop-in for the step()/reset()/get_after_states() surface of s0phia-/tetris
(game.py / state.py / tetromino.py), computed by hand-written HIP kernels for
gfx950 behind the C-ABI of include/tetris_hip.h.
"""
from .tetromino import CATALOGUE, PIECE_SETS, Tetromino, TetrominoSampler  # noqa: F401
from .vec_env import VecTetris  # noqa: F401
from .game import Tetris  # noqa: F401
from .state import State  # noqa: F401

__all__ = ["VecTetris", "Tetris", "State", "CATALOGUE", "PIECE_SETS", "Tetromino", "TetrominoSampler"]
