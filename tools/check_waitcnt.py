#!/usr/bin/env python3
"""Path-sensitive check of one kernel's gfx9-family assembly: on EVERY path from a load (LDS, scalar,
global, scratch) to the first instruction that reads or overwrites its destination registers there must
be an s_waitcnt that covers it.  Counters are modelled as hipcc models them on this target: vmcnt and
lgkmcnt retire in issue order, except that scalar loads may return out of order (only lgkmcnt(0) covers
them, and while one is pending no LDS result is covered by a partial wait either).
   check_waitcnt.py kernel.s [substring of the kernel's symbol]"""
import re
import sys

src = open(sys.argv[1]).read()
if len(sys.argv) > 2:
    m = re.search(r"^(\S*%s\S*):" % re.escape(sys.argv[2]), src, re.M)
    start = m.start()
    end = src.index(".amdhsa_kernel", start)
    src = src[start:end]
lines = [ln.split(";")[0].rstrip() for ln in src.split("\n")]
ins, labels = [], {}
for ln in lines:
    t = ln.strip()
    if not t or t.startswith("."):
        m = re.match(r"(\.L\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
        continue
    if re.match(r"\S+:$", t):
        continue
    ins.append(t)


def regs(tok):
    out = set()
    for kind, a, b, c in re.findall(r"\b([vs])\[(\d+):(\d+)\]|\b([vs]\d+)\b", tok):
        if c:
            out.add(c)
        else:
            out.update("%s%d" % (kind, r) for r in range(int(a), int(b) + 1))
    return out


LOAD_VM = re.compile(r"(global_load|scratch_load|buffer_load|flat_load)")
STORE_VM = re.compile(r"(global_store|scratch_store|buffer_store|flat_store|global_atomic)")
LOAD_LDS = re.compile(r"ds_(read|bpermute|permute|swizzle|consume|append)")
STORE_LDS = re.compile(r"ds_(write|or_b|add_u|and_b|xor_b|min_|max_)")
LOAD_SM = re.compile(r"s_(load|buffer_load)")
viol = {}
seen = set()
stack = [(0, (), ())]
steps = 0
while stack:
    pc, vm, lg = stack.pop()
    while True:
        key = (pc, vm, lg)
        if key in seen or pc >= len(ins):
            break
        seen.add(key)
        steps += 1
        t = ins[pc]
        op, _, rest = t.partition(" ")
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                vm = vm[len(vm) - n:] if n else ()
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    lg = ()
                elif not any(k == "sm" for k, _ in lg):
                    lg = lg[len(lg) - n:] if n < len(lg) else lg
            if not re.search(r"cnt\(", t) and re.search(r"\b0\b", rest):
                vm, lg = (), ()
            pc += 1
            continue
        touched = regs(rest)
        for q in (vm, lg):
            for kind, dst in q:
                if dst and touched & set(dst):
                    viol.setdefault((pc, t), set()).add(kind)
        if LOAD_VM.match(op):
            vm = vm + (("vm", tuple(sorted(regs(rest.split(",")[0])))),)
        elif STORE_VM.match(op):
            vm = vm + (("vmst", ()),)
        elif LOAD_LDS.match(op):
            lg = lg + (("lds", tuple(sorted(regs(rest.split(",")[0])))),)
        elif STORE_LDS.match(op):
            lg = lg + (("ldsst", ()),)
        elif LOAD_SM.match(op):
            lg = lg + (("sm", tuple(sorted(regs(rest.split(",")[0])))),)
        vm, lg = vm[-24:], lg[-24:]
        if op == "s_endpgm":
            break
        if op == "s_branch":
            pc = labels[rest.strip()]
            continue
        if op.startswith("s_cbranch"):
            stack.append((labels[rest.strip()], vm, lg))
        pc += 1
print("instructions %d, states visited %d, violations %d" % (len(ins), steps, len(viol)))
for (pc, t), kinds in sorted(viol.items())[:40]:
    print("  #%d  %s   <- pending %s" % (pc, t, sorted(kinds)))
