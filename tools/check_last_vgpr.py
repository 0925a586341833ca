#!/usr/bin/env python3
"""Gate against a gfx950 hardware hazard found in round 3 (tools/ubench/shift64_last_vgpr.hip):

    v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 whose 32-bit shift amount is the LAST vector register
    of the wave's allocation (v255 of 256, v167 of 168, ...) shift by (v0 & 63) instead whenever another
    wave shares the SIMD.  hipcc (ROCm 7.2) allocates that register like any other.

This script disassembles every kernel of a built libtetris_hip.so (or any HIP shared library / code
object) and lists the instructions that have the pattern.  Exit status 1 if there is one.
   check_last_vgpr.py [library.so]"""
import os
import re
import subprocess
import sys
import tempfile

LL = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
GRANULE = 8  # VGPR allocation granule of gfx90a / gfx94x / gfx950 (unified register file)
SHIFTS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")


def code_objects(path, tmp):
    """Device code objects (gfx950) inside a HIP shared library, object file or bare code object."""
    data = open(path, "rb").read()
    if data[:4] == b"\x7fELF" and b".hip_fatbin" in data:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "-j", ".hip_fatbin", path, fat])
        data = open(fat, "rb").read()
    elif data[:4] == b"\x7fELF":
        return [path]
    outs = []
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), data)]
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(data)
        part = os.path.join(tmp, "bundle%d.bin" % n)
        open(part, "wb").write(data[a:b])
        out = os.path.join(tmp, "co%d.elf" % n)
        r = subprocess.run([LL + "/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + part, "--output=" + out, "--unbundle"], capture_output=True)
        if r.returncode == 0 and os.path.getsize(out) > 0:
            outs.append(out)
    return outs


def kernel_vgprs(elf):
    """kernel symbol -> .vgpr_count from the code object's metadata note."""
    txt = subprocess.run([LL + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        v = re.search(r"\.vgpr_count:\s+(\d+)", blk)
        a = re.search(r"^\s*(\d+)", blk)
        if name and v:
            out[name.group(1)] = (int(v.group(1)), int(a.group(1)) if a else 0)
    return out


def check(path):
    bad, n_kernels, n_shifts = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for elf in code_objects(path, tmp):
            vg = kernel_vgprs(elf)
            dis = subprocess.run([LL + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
            cur, last = None, None
            for ln in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    if cur in vg:
                        n_kernels += 1
                        total, agpr = vg[cur]
                        arch = total - agpr  # unified file: .vgpr_count counts both; shift amounts are arch VGPRs
                        last = ((arch + GRANULE - 1) // GRANULE) * GRANULE - 1 if arch else None
                    else:
                        last = None
                    continue
                if last is None:
                    continue
                t = ln.strip()
                if t.startswith(SHIFTS):
                    n_shifts += 1
                    ops = t.split(None, 1)[1].split(",")
                    amount = ops[1].strip()
                    if amount == "v%d" % last:
                        bad.append((cur, t.split("//")[0].strip(), vg[cur]))
    return bad, n_kernels, n_shifts


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tetris_amd", "csrc", "libtetris_hip.so")
    bad, nk, ns = check(lib)
    print("%s: %d kernels, %d 64-bit shifts, %d with the shift amount in the last allocated VGPR" % (lib, nk, ns, len(bad)))
    seen = {}
    for k, ins, vg in bad:
        seen.setdefault(k, []).append(ins)
    for k, ins in seen.items():
        short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print("  %s\n      %d site(s), e.g. %s" % (short[:110], len(ins), ins[0]))
    sys.exit(1 if bad else 0)
