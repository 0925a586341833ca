"""In-tree build of libtetris_hip.so (hipcc, gfx950 only)."""
import os
import shutil
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO_PATH = os.path.join(_CSRC, "libtetris_hip.so")
_SOURCES = ["tetris_kernels.hip", "tetris_core.hpp", "tetris_table.hpp", "tetris_feature_lut.inc", "tetris_feature_lut10.inc"]
_HEADER = os.path.join(os.path.dirname(_CSRC), "..", "include", "tetris_hip.h")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtetris_hip.so cannot be built (set HIPCC=/path/to/hipcc)")


def is_stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(_CSRC, s) for s in _SOURCES] + [_HEADER]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """Compile the HIP kernels + C-ABI for gfx950 into tetris_amd/csrc/libtetris_hip.so."""
    if not force and not is_stale():
        return SO_PATH
    tmp = "%s.%d.tmp" % (SO_PATH, os.getpid())  # build aside, then rename: never a half-written .so
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
           os.path.join(_CSRC, "tetris_kernels.hip"), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, SO_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return SO_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
