"""The gfx950 hazard of round 3 (a 64-bit shift whose shift amount sits in the last allocated VGPR reads v0
instead when another wave shares the SIMD: tools/ubench/shift64_last_vgpr.hip): the build's
assembly patch, and a disassembly check of the library that was actually built.  CPU only."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ASM = """
\t.text
k1:
\tv_lshlrev_b64 v[10:11], v7, v[2:3]
\tv_lshlrev_b64 v[10:11], v15, -1
\tv_lshrrev_b64 v[4:5], v15, v[4:5]
\tv_ashrrev_i64 v[6:7], v15, v[8:9]
\tv_lshlrev_b32_e32 v1, v15, v2
\ts_endpgm
\t.amdhsa_kernel k1
\t\t.amdhsa_next_free_vgpr 16
\t\t.amdhsa_accum_offset 16
\t.end_amdhsa_kernel
k2:
\tv_lshlrev_b64 v[10:11], v15, -1
\ts_endpgm
\t.amdhsa_kernel k2
\t\t.amdhsa_next_free_vgpr 19
\t\t.amdhsa_accum_offset 20
\t.end_amdhsa_kernel
"""


def test_patch_moves_the_amount_off_the_last_register():
    from tetris_amd import build
    out, n = build.patch_last_vgpr_shifts(ASM)
    assert n == 3
    body = out.split("k2:")[0]
    assert "v_lshlrev_b64 v[10:11], v7, v[2:3]" in body                                   # not the last register: untouched
    assert "v_mov_b32_e32 v10, v15" in body and "v_lshlrev_b64 v[10:11], v10, -1" in body  # through the destination's low half
    assert "v_swap_b32 v0, v15" in body and "v_lshrrev_b64 v[4:5], v0, v[4:5]" in body     # in place: through v0
    assert "v_mov_b32_e32 v6, v15" in body and "v_ashrrev_i64 v[6:7], v6, v[8:9]" in body
    assert "v_lshlrev_b32_e32 v1, v15, v2" in body                                         # 32-bit shifts are not affected
    assert "v_lshlrev_b64 v[10:11], v15, -1" in out.split("k2:")[1]                        # 19 registers -> 24 allocated: v15 is not the last
    assert "v_lshlrev_b64 v[10:11], v15, -1" not in body


@pytest.mark.timeout(300)
def test_built_library_has_no_shift_amount_in_the_last_vgpr():
    from tetris_amd import build
    import check_last_vgpr
    if not os.path.exists(build.SO_PATH):
        pytest.skip("library not built")
    if not os.path.exists(check_last_vgpr.LL + "/llvm-objdump"):
        pytest.skip("no llvm-objdump on this machine")
    bad, n_kernels, n_shifts = check_last_vgpr.check(build.SO_PATH)
    assert n_kernels > 300 and n_shifts > 10000, (n_kernels, n_shifts)  # the scan really saw the kernels
    assert bad == [], bad[:3]
