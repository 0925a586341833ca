"""CPU-only checks of the boundary: the C-ABI library loads without a GPU and exports every
symbol include/tetris_hip.h declares; argument errors come back as codes before any launch;
the host-side tables agree with the library."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from tetris_amd import _lib
    old = _lib._install_test_backend(None)  # make sure the real library is what load() returns
    try:
        b = _lib.load()
    finally:
        _lib._install_test_backend(old)
    return b


def test_header_symbols_are_exported(hip_lib):
    header = open(os.path.join(ROOT, "include", "tetris_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(tetris_hip_\w+)\s*\(", header)))
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(hip_lib.cdll, name), "libtetris_hip.so does not export %s" % name
    from tetris_amd import _lib
    assert set(_lib.EXPORTS) == set(declared)
    assert hip_lib.version() == _lib.ABI_VERSION


def test_desc_init_and_argument_errors(hip_lib):
    from tetris_amd import _lib
    d = _lib.TetrisDesc()
    ids = (ctypes.c_int32 * 2)(4, 3)
    assert hip_lib.desc_init(ctypes.byref(d), 10, 20, ids, 2, None) == 0
    assert (d.word_bytes, d.a_max, d.n_pieces) == (4, 36, 2)
    assert hip_lib.desc_init(ctypes.byref(d), 10, 40, ids, 2, None) == 0 and d.word_bytes == 8
    assert hip_lib.desc_init(ctypes.byref(d), 13, 20, ids, 2, None) == -3     # TETRIS_E_COLUMNS (5..12 are built)
    assert hip_lib.desc_init(ctypes.byref(d), 4, 20, ids, 2, None) == -3
    assert hip_lib.desc_init(ctypes.byref(d), 10, 3, ids, 2, None) == -4      # TETRIS_E_ROWS
    assert hip_lib.desc_init(ctypes.byref(d), 10, 20, ids, 0, None) == -5     # TETRIS_E_PIECES
    bad = (ctypes.c_int32 * 1)(9)
    assert hip_lib.desc_init(ctypes.byref(d), 10, 20, bad, 1, None) == -5
    assert hip_lib.desc_init(ctypes.byref(d), 10, 20, ids, 2, None) == 0
    # NULL pointers / bad batch are refused before anything is launched (no GPU needed)
    assert hip_lib.step(ctypes.byref(d), None, None, None, None, None, None, 0, None, None, None, None, None, None,
                        None, 0, 0, 0, 0, 16, None) == -1
    assert hip_lib.reset(ctypes.byref(d), None, None, None, None, None, None, None, 0, None, 1, 0, 0, 0, 16, None) == -1
    z = _lib.TetrisDesc()
    assert hip_lib.refresh(ctypes.byref(z), None, None, None, 16, None) == -2  # uninitialised descriptor
    assert "NULL" in hip_lib.error_string(-1)
    cols = (ctypes.c_int32 * 16)()
    n = hip_lib.supported_columns(cols, 16)
    assert list(cols)[:n] == [5, 6, 7, 8, 9, 10, 11, 12]
    from tetris_amd import build
    assert list(build._COLUMNS) == list(cols)[:n]  # the per-column translation units of the in-tree build
    assert hip_lib.status_words(1 << 20) == 4 * ((1 << 20) // 64)


def test_host_tables_match_library(hip_lib):
    from tetris_amd.tetromino import CATALOGUE, ORIENTATIONS, n_placements
    want = dict(Straight=17, Square=9, SnakeR=17, ThreeLine=18, ThreeL=36, SnakeL=17, T=34, RCorner=34, LCorner=34)
    for i, name in enumerate(CATALOGUE):
        for C in (5, 6, 7, 8, 9, 10):
            assert hip_lib.n_placements(i, C) == n_placements(name, C)
        assert n_placements(name, 10) == want[name]
        for loop in ORIENTATIONS[name]:
            for w, b, n in loop:
                assert len(b) == len(n) == w and min(b) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without the HIP extension (and without hipcc) the package refuses to work."""
    from tetris_amd import _lib, build
    old = _lib._install_test_backend(None)
    try:
        monkeypatch.setattr(build, "SO_PATH", str(tmp_path / "nope.so"))
        monkeypatch.setattr(build, "_hipcc", lambda: (_ for _ in ()).throw(RuntimeError("no hipcc")))
        with pytest.raises(ImportError):
            _lib.load()
    finally:
        _lib._install_test_backend(old)


def test_vec_env_rejects_cpu_device_without_test_backend():
    from tetris_amd import VecTetris, _lib
    old = _lib._install_test_backend(None)
    try:
        with pytest.raises(ValueError):
            VecTetris(10, 20, 4, device="cpu")
    finally:
        _lib._install_test_backend(old)


def test_done_bit_packing_roundtrip():
    import torch
    from tetris_amd.distributed import pack_done_bits, shard_range, unpack_done_bits
    d = torch.rand(1003) < 0.1
    bits = pack_done_bits(d)
    assert bits.numel() == 126 and torch.equal(unpack_done_bits(bits, 1003), d)
    assert shard_range(1 << 23, 3, 8) == (3 << 20, 4 << 20)
    with pytest.raises(ValueError):
        shard_range(10, 0, 3)
