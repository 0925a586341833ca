"""Test-only: run tetris_amd's host logic on CPU tensors by binding the g++
build of the per-lane device code (tests/harness).  The product never does
this -- see tetris_amd._lib._install_test_backend."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "harness")
_SO = os.path.join(_DIR, "libtetris_core_host.so")


def build():
    subprocess.check_call(["make", "-C", _DIR, "libtetris_core_host.so"], stdout=subprocess.DEVNULL)
    return _SO


def binding():
    from tetris_amd import _lib
    build()
    return _lib._Binding(ctypes.CDLL(_SO), prefix="tetris_host_", device_type="cpu")
