#!/usr/bin/env python3
"""Static per-kernel report from the gfx950 assembly of tetris_kernels.hip: instruction counts
(total / VALU / LDS / scratch) and the registers, LDS and spill bytes each kernel was given.
usage: kernel_resources.py [substring-filter] [-- extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
flt = args[0] if args else "Li10E"
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", asm] + extra +
                          [os.path.join(ROOT, "tetris_amd", "csrc", "tetris_kernels.hip")], stderr=subprocess.DEVNULL)
    s = open(asm).read()
for name in re.findall(r"\.amdhsa_kernel (\S+)", s):
    if flt not in name:
        continue
    i = s.index(name + ":")
    j = s.index(".amdhsa_kernel", i)
    ins = [ln.strip() for ln in s[i:j].split("\n")
           if ln.strip() and not ln.strip().startswith((".", ";", "_")) and not ln.strip().endswith(":")]
    c = collections.Counter(ln.split()[0] for ln in ins)
    k = s.index(".amdhsa_kernel " + name)
    r = dict(re.findall(r"\.amdhsa_(group_segment_fixed_size|next_free_vgpr|private_segment_fixed_size) (\d+)",
                        s[k:k + 3000]))
    short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(anonymous namespace\)::|\(.*", "", short)
    print("%-46s instr %5d VALU %5d LDS %4d scratch %3d | vgpr %3s lds %6s spill %4s B" % (
        short[:46], len(ins), sum(v for q, v in c.items() if q.startswith("v_")),
        sum(v for q, v in c.items() if q.startswith("ds_")), sum(v for q, v in c.items() if q.startswith("scratch_")),
        r.get("next_free_vgpr"), r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size")))
