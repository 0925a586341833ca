#!/usr/bin/env python3
"""Assembly-level instrumentation of one kernel WITHOUT touching its register allocation or schedule (every
source-level probe of the round-3 fault made it vanish): at chosen instruction indices the named VGPRs
(or the lane bit of an SGPR-pair mask, "m<sgpr>") are stored to scratch slots beyond the kernel's own frame, and
at the end of the kernel the slots are written behind best_value[B] (16 dwords per env); tools/coresidency_instr.py
reads them.  It is how the fault was traced to `v_lshlrev_b64 v[178:179], v255, -1`.
   asm_variant.py prepare ; asm_instrument.py "2522:v178 v179 v255 v240;2912:v126 v127"
(indices are instruction-line offsets inside the kernel's text in build_variants/asm/dev.s; the script is tied to
greedy_kernel<unsigned long, 12, 4> of that build: label .LBB35_605, free registers v200-v202 / s96-s101.)"""
import re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import asm_variant as av
SITES = {}
slot = 0
for spec in sys.argv[1].split(";"):
    idx, regs = spec.split(":")
    SITES[int(idx)] = [(r, slot + k) for k, r in enumerate(regs.split())]
    slot += len(regs.split())
lines=open(os.path.join(av.D, 'dev.s')).read().split('\n')
start=next(i for i,l in enumerate(lines) if re.match(r'^\S*greedy_kernelImLi12ELi4E\S*:',l))
end=next(i for i in range(start,len(lines)) if lines[i].strip().startswith('s_endpgm'))
out=[]
nsite=0
for i,l in enumerate(lines):
    if i==start+2:
        out += ["\ts_mov_b64 s[100:101], s[0:1]", "\ts_mov_b32 s98, 0"]
    if start<=i<end and l.strip()==".LBB35_605:":
        out.append(l)
        continue
    out.append(l)
    if start<=i<end and (i-start) in SITES:
        nsite+=1
        for r,k in SITES[i-start]:
            if r.startswith("m"):   # lane bit of an SGPR-pair mask, through v240 (must be dead at the site)
                n=int(r[1:])
                out += ["\tv_cndmask_b32_e64 v240, 0, 1, s[%d:%d]" % (n, n+1), "\tscratch_store_dword off, v240, off offset:%d" % (512+4*k)]
            else:
                out += ["\tscratch_store_dword off, %s, off offset:%d" % (r, 512+4*k)]
    if start<=i<end and i>0 and lines[i-1].strip()==".LBB35_605:" and "s_or_b64 exec, exec, s[4:5]" in l:
        K=16
        out += ["\ts_load_dwordx2 s[96:97], s[100:101], 0x18", "\ts_load_dwordx2 s[98:99], s[100:101], 0x28", "\ts_waitcnt vmcnt(0) lgkmcnt(0)",
                "\ts_lshl_b64 s[98:99], s[98:99], 2", "\ts_add_u32 s96, s96, s98", "\ts_addc_u32 s97, s97, s99",
                "\tv_lshlrev_b64 v[200:201], 6, v[0:1]", "\tv_lshl_add_u64 v[200:201], v[200:201], 0, s[96:97]"]
        for k in range(K):
            out += ["\tscratch_load_dword v202, off, off offset:%d" % (512+4*k), "\ts_waitcnt vmcnt(0)", "\tglobal_store_dword v[200:201], v202, off offset:%d" % (4*k)]
        out += ["\ts_waitcnt vmcnt(0)"]
print("sites instrumented:", nsite)
s='\n'.join(out)
m=re.search(r"(\.amdhsa_kernel _ZN12_GLOBAL__N_113greedy_kernelImLi12ELi4E.*?\.amdhsa_next_free_sgpr )96", s, re.S)
s=s[:m.end()-2]+"102"+s[m.end():]
m=re.search(r"(\.amdhsa_kernel _ZN12_GLOBAL__N_113greedy_kernelImLi12ELi4E.*?\.amdhsa_private_segment_fixed_size )252", s, re.S)
s=s[:m.end()-3]+"1024"+s[m.end():]
# metadata (msgpack notes) also carries the private segment size the runtime allocates from
s=re.sub(r"(\.name:\s+_ZN12_GLOBAL__N_113greedy_kernelImLi12ELi4EEEvNS_12GreedyParamsE\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+)252", r"\g<1>1024", s)
p=os.path.join(av.D, 'instr.s')
open(p,'w').write(s)
print(av.link(p,'instr'))
