"""ctypes binding of libtetris_hip.so (the C-ABI declared in include/tetris_hip.h).

There is no CPU fallback: if the shared library is missing this module raises,
and every call maps a non-zero return code to an exception.
"""
import ctypes
import os

from . import build as _build

MAX_PIECES = 12
ABI_VERSION = 6


class TetrisDesc(ctypes.Structure):
    """Mirror of ``struct TetrisDesc`` (include/tetris_hip.h)."""
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("num_columns", ctypes.c_int32),
        ("num_rows", ctypes.c_int32),
        ("word_bytes", ctypes.c_int32),
        ("n_pieces", ctypes.c_int32),
        ("piece_ids", ctypes.c_int32 * MAX_PIECES),
        ("a_max", ctypes.c_int32),
        ("has_direct_by", ctypes.c_int32),
        ("direct_by", ctypes.c_float * 8),
    ]


class TetrisHipError(RuntimeError):
    pass


_vp, _i32, _i64, _u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
_dp = ctypes.POINTER(TetrisDesc)

# name -> argtypes; every function returns int (0 = ok)
SIGNATURES = {
    "tetris_hip_version": [],
    "tetris_hip_supported_columns": [_vp, ctypes.c_int],
    "tetris_hip_desc_init": [_dp, _i32, _i32, _vp, _i32, _vp],
    "tetris_hip_n_placements": [_i32, _i32],
    "tetris_hip_status_words": [_i64],
    "tetris_hip_n_planes": [_dp],
    "tetris_hip_board_words": [_dp, _i64],
    "tetris_hip_reset": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _u64, _u64, _i64, _i64, _vp],
    "tetris_hip_step": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _u64, _u64,
                        _i64, _i64, _vp],
    "tetris_hip_step_call_size": [],
    "tetris_hip_step_call_init": [_vp, _dp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _u64,
                                  _i64, _i64],
    "tetris_hip_step_call_run": [_vp, _vp, _u64, _vp],
    "tetris_hip_step_call_run_gather": [_vp, _vp, _u64, _vp, _vp, _vp],
    "tetris_hip_stream_link": [_vp, _vp],
    "tetris_hip_step_call_run_counted": [_vp, _vp, _vp, ctypes.c_uint32, _vp],
    "tetris_hip_counter_add": [_vp, _u64, _vp],
    "tetris_hip_pack_done_bits": [_vp, _vp, _i64, _vp],
    "tetris_hip_step_many": [_dp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _u64, _u64,
                             _i64, _i64, _vp],
    "tetris_hip_afterstates": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "tetris_hip_policy_greedy": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "tetris_hip_rollouts": [_dp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _u64, _u64, _i64, _i64, _vp],
    "tetris_hip_policy_random": [_vp, _vp, _u64, _u64, _i64, _i64, _vp],
    "tetris_hip_numpy_bag_stream": [_vp, _i32, _i64, _vp, _i64, _vp],
    "tetris_hip_decode": [_dp, _vp, _vp, _vp, _i64, _vp],
    "tetris_hip_encode": [_dp, _vp, _vp, _i64, _vp],
    "tetris_hip_refresh": [_dp, _vp, _vp, _vp, _i64, _vp],
}
EXPORTS = list(SIGNATURES) + ["tetris_hip_error_string", "tetris_hip_source_hash"]


class _Binding:
    """Typed view of a library exporting the C-ABI (prefix selects the symbol family)."""

    def __init__(self, cdll, prefix="tetris_hip_", device_type="cuda"):
        self.cdll = cdll
        self.device_type = device_type
        for name, argtypes in SIGNATURES.items():
            sym = prefix + name[len("tetris_hip_"):]
            if not hasattr(cdll, sym):
                continue
            fn = getattr(cdll, sym)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int64 if name.endswith(("status_words", "step_call_size", "board_words")) else ctypes.c_int
            setattr(self, name[len("tetris_hip_"):], fn)
        if hasattr(cdll, prefix + "error_string"):
            self._errstr = getattr(cdll, prefix + "error_string")
            self._errstr.argtypes = [ctypes.c_int]
            self._errstr.restype = ctypes.c_char_p
        else:
            self._errstr = None
        self._srchash = None
        if hasattr(cdll, prefix + "source_hash"):
            self._srchash = getattr(cdll, prefix + "source_hash")
            self._srchash.argtypes = []
            self._srchash.restype = ctypes.c_char_p

    def source_hash(self):
        """Hash of the kernel sources the loaded library was built from ("unknown" for ad-hoc builds)."""
        return self._srchash().decode() if self._srchash is not None else "unknown"

    def error_string(self, code):
        if self._errstr is not None:
            return self._errstr(code).decode()
        return "error %d" % code

    def check(self, code, what):
        if code != 0:
            raise TetrisHipError("%s failed: %s (code %d)" % (what, self.error_string(code), code))


_BINDING = None


def load():
    """Load libtetris_hip.so.  A missing library is built once if hipcc is available (an existing
    one is never rebuilt implicitly: N ranks start at once and file times do not survive copies;
    use `python -m tetris_amd.build` or TETRIS_AMD_REBUILD=1 after editing csrc/)."""
    global _BINDING
    if _BINDING is not None:
        return _BINDING
    path = _build.SO_PATH
    if not os.path.exists(path) or os.environ.get("TETRIS_AMD_REBUILD") == "1":
        try:
            _build.build_hip(force=True)
        except Exception as exc:  # no hipcc on this machine
            if not os.path.exists(path):
                raise ImportError(
                    "tetris_amd needs its HIP extension %s and it could not be built: %s. "
                    "Run `python -m tetris_amd.build` on a machine with hipcc." % (path, exc))
    cdll = ctypes.CDLL(path)
    b = _Binding(cdll)
    if b.version() != ABI_VERSION:
        raise ImportError("libtetris_hip.so ABI %d != expected %d: rebuild with `python -m tetris_amd.build`"
                          % (b.version(), ABI_VERSION))
    _BINDING = b
    return b


def _install_test_backend(binding):
    """TEST-SUITE HOOK ONLY (tests/harness): swap the binding so the host logic
    can be exercised on CPU tensors.  Never called by the package itself."""
    global _BINDING
    old = _BINDING
    _BINDING = binding
    return old
