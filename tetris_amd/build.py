"""In-tree build of libtetris_hip.so (hipcc, gfx950 only)."""
import os
import shutil
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO_PATH = os.path.join(_CSRC, "libtetris_hip.so")
_SOURCES = ["tetris_kernels.hip", "tetris_core.hpp", "tetris_table.hpp", "tetris_feature_lut.inc", "tetris_feature_lut10.inc",
            "tetris_after_lut.inc"]
_HEADER = os.path.join(os.path.dirname(_CSRC), "..", "include", "tetris_hip.h")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtetris_hip.so cannot be built (set HIPCC=/path/to/hipcc)")


def source_hash():
    """Content hash of the kernel sources (what bench.py calls csrc_hash): compiled into the library as
    tetris_hip_source_hash() so that a measurement can be tied to the binary that produced it."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(_CSRC)):
        if name.endswith((".hip", ".hpp", ".inc", ".h")):
            h.update(name.encode())
            with open(os.path.join(_CSRC, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(_CSRC, s) for s in _SOURCES] + [_HEADER]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


_COLUMNS = (5, 6, 7, 8, 9, 10, 11, 12)  # == TET_COLUMNS in csrc/tetris_table.hpp (checked by the CPU tests)
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def build_hip(force=False, verbose=False, jobs=None):
    """Compile the HIP kernels + C-ABI for gfx950 into tetris_amd/csrc/libtetris_hip.so.

    The kernels are templates on the column count; the source is compiled once per column count
    (-DTET_PART=<C>) plus once as the main unit (-DTET_SPLIT_MAIN, the C-ABI), in parallel hipcc
    processes, and the objects are linked (see "translation units" in tetris_kernels.hip)."""
    if not force and not is_stale():
        return SO_PATH
    import tempfile
    hipcc = _hipcc()
    src = os.path.join(_CSRC, "tetris_kernels.hip")
    jobs = jobs or max(1, min(len(_COLUMNS) + 1, (os.cpu_count() or 2)))
    tmp = "%s.%d.tmp" % (SO_PATH, os.getpid())  # build aside, then rename: never a half-written .so
    with tempfile.TemporaryDirectory(prefix="tetris_build_") as d:
        units = [("main", ["-DTET_SPLIT_MAIN", '-DTET_SRC_HASH="%s"' % source_hash()])] + \
                [("c%d" % c, ["-DTET_PART=%d" % c]) for c in _COLUMNS]
        objs, running = [], []
        pending = list(units)

        def reap(block):
            for pr, name in list(running):
                if pr.poll() is None and not block:
                    continue
                if pr.wait() != 0:
                    for other, _ in running:
                        if other.poll() is None:
                            other.kill()
                    raise subprocess.CalledProcessError(pr.returncode, "hipcc -c (%s)" % name)
                running.remove((pr, name))
                if block:
                    return

        while pending or running:
            while pending and len(running) < jobs:
                name, defs = pending.pop(0)
                obj = os.path.join(d, name + ".o")
                cmd = [hipcc] + _FLAGS + defs + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                running.append((subprocess.Popen(cmd), name))
                objs.append(obj)
            if running:
                reap(block=True)
        try:
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, SO_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return SO_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
