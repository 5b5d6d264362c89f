"""
The inline-asm DPP FMAs of ``tbmodels_amd/csrc/tbk_dpp.h`` are outside LLVM's hazard recogniser; the built gfx950
objects are checked instead (``tools/dpp_hazard_lint.py``): no VALU write of a DPP source register less than 2 wait
states, and no VALU write of EXEC less than 5, ahead of a ``v_fmac_f64_dpp``.
"""

import glob
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dpp_hazard_lint as lint  # noqa: E402  pylint: disable=wrong-import-position

CLEAN = """
0000000000001900 <kernel_a>:
	v_mov_b32_e32 v4, v9                                       // 000000001900: 7E080309
	ds_read_b64 v[10:11], v3                                   // 000000001904: D8EC0000
	s_waitcnt lgkmcnt(0)                                       // 000000001908: BF8CC07F
	v_add_f64 v[20:21], v[0:1], v[2:3]                         // 00000000190C: D2800014
	v_mul_f64 v[22:23], v[0:1], v[2:3]                         // 000000001910: D2810016
	v_fmac_f64_dpp v[6:7], v[10:11], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf // 000000001914: 080C18FA
	v_fmac_f64_dpp v[6:7], -v[10:11], v[14:15] row_newbcast:4 row_mask:0xf bank_mask:0xf // 00000000191C: 080C1CFA
"""

HAZARD_VGPR = CLEAN.replace("v_mul_f64 v[22:23], v[0:1], v[2:3]", "v_mul_f64 v[10:11], v[0:1], v[2:3]")
HAZARD_ONE_APART = CLEAN.replace("v_add_f64 v[20:21], v[0:1], v[2:3]", "v_add_f64 v[10:11], v[0:1], v[2:3]")
SPACED_BY_NOP = HAZARD_VGPR.replace(
    "	v_fmac_f64_dpp v[6:7], v[10:11]", "	s_nop 1                                                    // 0: BF800001\n	v_fmac_f64_dpp v[6:7], v[10:11]", 1)
HAZARD_EXEC = CLEAN.replace("v_mov_b32_e32 v4, v9", "v_cmpx_gt_i32_e32 v4, v9")


def test_lint_recognises_the_hazards_it_is_there_for():
    assert lint.count_dpp(CLEAN) == 2
    assert lint.lint(CLEAN) == []
    assert len(lint.lint(HAZARD_VGPR)) == 2 and "v_mul_f64" in lint.lint(HAZARD_VGPR)[0][2]  # both FMAs are within 2 states
    assert len(lint.lint(HAZARD_ONE_APART)) == 1  # one instruction in between is one wait state: still too close
    assert lint.lint(SPACED_BY_NOP) == []
    assert any("EXEC" in p[2] for p in lint.lint(HAZARD_EXEC))


def test_built_objects_have_no_dpp_hazard(tmp_path):
    objects = sorted(glob.glob(os.path.join(ROOT, "tbmodels_amd", "csrc", "*.o")))
    tools = [shutil.which("objcopy"), os.path.join(lint.LLVM, "clang-offload-bundler"), os.path.join(lint.LLVM, "llvm-objdump")]
    if not objects or not all(t and os.path.exists(t) for t in tools):
        pytest.skip("no built objects / no LLVM binutils here")
    seen = 0
    for obj in objects:
        if os.path.basename(obj) not in ("tbk_eig_small.o", "tbk_eig_band.o"):
            continue  # the files that include tbk_dpp.h's FMAs
        text = lint.disassemble(obj, str(tmp_path))
        assert text is not None, obj
        seen += lint.count_dpp(text)
        problems = lint.lint(text)
        assert problems == [], problems[:5]
    assert seen > 1000  # the check looked at the real thing
