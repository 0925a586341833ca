#!/usr/bin/env python3
"""Builds a variant of the library from EDITED device assembly (no recompilation: seconds per variant).
   asm_variant.py prepare                 # once: device assembly of the 12-column single-unit build -> build_variants/asm/dev.s
   asm_variant.py count <kernel-substring>
   asm_variant.py nops <name> <kernel-substring> <first> <last> [pad]
        s_nop <pad> after every instruction with index first..last of that kernel -> build_variants/libasm_<name>.so
Used to bisect the co-residency fault by instruction range.  NOTE: these variants are assembled from
the compiler's UNPATCHED assembly on purpose (the product and tools/ablate.py go through
tetris_amd.build.patch_last_vgpr_shifts)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "build_variants", "asm")
LL = "/opt/rocm/lib/llvm/bin"
SRC = os.path.join(ROOT, "tetris_amd", "csrc", "tetris_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-DTET_COLUMNS(X)=X(12)"]


def link(asm_path, name):
    o, out, fb, host = (os.path.join(D, name + e) for e in (".o", ".out", ".hipfb", ".host.o"))
    subprocess.check_call([LL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm_path, "-o", o])
    subprocess.check_call([LL + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", o, "-o", out])
    subprocess.check_call([LL + "/clang-offload-bundler", "-type=o", "-bundle-align=4096",
                           "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null",
                           "-input=" + out, "-output=" + fb])
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb,
                           "-c", SRC, "-o", host], stderr=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "build_variants", "libasm_%s.so" % name)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", host, "-o", lib])
    for f in (o, out, fb, host):
        os.remove(f)
    return lib


def kernel_span(lines, sub):
    start = next(i for i, ln in enumerate(lines) if re.match(r"^\S*%s\S*:" % re.escape(sub), ln))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return start, end


def is_instr(ln):
    t = ln.split(";")[0].strip()
    return bool(t) and not t.startswith(".") and not t.endswith(":")


if __name__ == "__main__":
    os.makedirs(D, exist_ok=True)
    if sys.argv[1] == "prepare":
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", SRC, "-o", os.path.join(D, "dev.s")],
                              stderr=subprocess.DEVNULL)
        print(link(os.path.join(D, "dev.s"), "base"))
    elif sys.argv[1] == "base":
        print(link(os.path.join(D, "dev.s"), "base"))
    elif sys.argv[1] == "count":
        lines = open(os.path.join(D, "dev.s")).read().split("\n")
        a, b = kernel_span(lines, sys.argv[2])
        print(sum(is_instr(ln) for ln in lines[a:b]))
    elif sys.argv[1] == "nops":
        name, sub, first, last = sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
        pad = sys.argv[6] if len(sys.argv) > 6 else "0"
        lines = open(os.path.join(D, "dev.s")).read().split("\n")
        a, b = kernel_span(lines, sub)
        out, k, in_asm = [], 0, False
        for i, ln in enumerate(lines):
            out.append(ln)
            if a <= i < b:
                if "#ASMSTART" in ln:
                    in_asm = True
                if "#ASMEND" in ln:
                    in_asm = False
                if is_instr(ln):
                    if first <= k <= last and not in_asm:
                        out.append("\ts_nop %s" % pad)
                    k += 1
        p = os.path.join(D, name + ".s")
        open(p, "w").write("\n".join(out))
        print(link(p, name), "(%d instructions in the kernel)" % k)
        os.remove(p)
