"""State: host-side view of one env's board, with the attribute surface of the
reference's ``state.State`` (state.py:5-38) that callers read.

The reference State *computes* (line clear, terminal test, features) in its
constructor; here all of that already happened inside the HIP kernels, and a
State is a decoded snapshot: a ``(R+4) x C`` int64 ``representation`` (row 0 =
bottom), ``lowest_free_rows`` and the BCTS feature vector.
"""
import numpy as np


class State:
    def __init__(self, representation, lowest_free_rows=None, anchor_col=0, anchor_row=0, n_cleared_lines=0,
                 landing_height_bonus=0.0, features=None, terminal_state=False, num_features=8,
                 feature_type="bcts"):
        self.representation = np.asarray(representation, dtype=np.int_)
        self.num_rows, self.num_columns = self.representation.shape  # state.py:27-28 (num_rows = R + 4)
        self.n_legal_rows = self.num_rows - 4                         # state.py:30
        if lowest_free_rows is None:                                  # state.py:22-23,162-172
            filled = self.representation != 0
            top = self.num_rows - np.argmax(filled[::-1], axis=0)
            lowest_free_rows = np.where(filled.any(axis=0), top, 0)
        self.lowest_free_rows = np.asarray(lowest_free_rows, dtype=np.int_)
        self.anchor_col = int(anchor_col)
        self.anchor_row = int(anchor_row)
        self.n_cleared_lines = int(n_cleared_lines)
        self.landing_height_bonus = float(landing_height_bonus)
        self.num_features = num_features
        self.feature_type = feature_type
        self.features = None if features is None else np.asarray(features, dtype=np.float32)
        self.terminal_state = bool(terminal_state)
        self.reward = 0 if self.terminal_state else self.n_cleared_lines  # state.py:37
        self.value_estimate = 0.0

    def get_features(self, direct_by=None, order_by=None, standardize_by=None, addRBF=False):
        """state.py:43-55: cached float32 features, optionally times ``direct_by``
        (NumPy promotion applies: an int64 direction vector yields float64)."""
        if self.feature_type != "bcts":
            raise ValueError("Only 'bcts' features implemented.")  # state.py:91-95
        if self.features is None:
            raise ValueError("this State carries no feature vector (it was not produced by the env)")
        features = self.features
        if direct_by is not None:
            features = features * direct_by
        return features

    def print_board_to_string(self):
        """state.py:69-81: the R legal rows, top first."""
        out = "\n"
        for r in range(self.n_legal_rows - 1, -1, -1):
            out += "|" + "".join("██" if v else "  " for v in self.representation[r]) + "|\n"
        return out

    def print_board(self):
        for r in range(self.n_legal_rows - 1, -1, -1):
            print("| " + " ".join("██" if v else "  " for v in self.representation[r]) + " |")

    def __repr__(self):
        return self.print_board_to_string()


def print_board_to_string(state):
    """utils.print_board_to_string (utils.py:179-191): all R+4 stored rows, top first."""
    out = "\n"
    for r in range(state.num_rows - 1, -1, -1):
        out += "|" + "".join("██" if v else "  " for v in state.representation[r]) + "|\n"
    return out
