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
# a swap writes BOTH operands: the DPP source is the swap's second operand here
HAZARD_SWAP_SECOND = CLEAN.replace("v_mul_f64 v[22:23], v[0:1], v[2:3]", "v_permlane16_swap_b32_e32 v30, v10")
SWAP_ELSEWHERE = CLEAN.replace("v_mul_f64 v[22:23], v[0:1], v[2:3]", "v_permlane16_swap_b32_e32 v30, v31")

# a loop whose body STARTS with the DPP FMA: the producer sits at the bottom of the loop, in front of the back edge;
# plain `llvm-objdump -d` prints no labels and a straight-line scan never meets it
LOOP_BACK_EDGE = """
0000000000002000 <kernel_b>:
	v_mov_b32_e32 v4, v9                                       // 000000002000: 7E080309
	v_add_f64 v[20:21], v[0:1], v[2:3]                         // 000000002004: D2800014
	v_add_f64 v[24:25], v[0:1], v[2:3]                         // 000000002008: D2800014

000000000000200c <L0>:
	v_fmac_f64_dpp v[6:7], v[10:11], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf // 00000000200C: 080C18FA
	s_add_i32 s4, s4, 1                                        // 000000002014: 81048104
	s_cmp_lt_i32 s4, s5                                        // 000000002018: BF040504
	v_mul_f64 v[10:11], v[6:7], v[2:3]                         // 00000000201C: D2810016
	s_cbranch_scc1 L0                                          // 000000002020: BF85FFFA
	s_endpgm                                                   // 000000002024: BF810000
"""
LOOP_SPACED = LOOP_BACK_EDGE.replace("	s_cbranch_scc1 L0", "	s_nop 0                                                    // 0: BF800000\n	s_cbranch_scc1 L0")
# a forward branch over the spacing instructions: the taken path arrives one state after the write
BRANCH_OVER_SPACING = """
0000000000003000 <kernel_c>:
	v_mul_f64 v[10:11], v[6:7], v[2:3]                         // 000000003000: D2810016
	s_cbranch_vccz L1                                          // 000000003008: BF86000A
	v_add_f64 v[20:21], v[0:1], v[2:3]                         // 00000000300C: D2800014
	v_add_f64 v[24:25], v[0:1], v[2:3]                         // 000000003010: D2800014

0000000000003014 <L1>:
	v_fmac_f64_dpp v[6:7], v[10:11], v[12:13] row_newbcast:3 row_mask:0xf bank_mask:0xf // 000000003014: 080C18FA
	s_endpgm                                                   // 00000000301C: BF810000
"""


def test_lint_recognises_the_hazards_it_is_there_for():
    assert lint.count_dpp(CLEAN) == 2
    assert lint.lint(CLEAN) == []
    assert len(lint.lint(HAZARD_VGPR)) == 2 and "v_mul_f64" in lint.lint(HAZARD_VGPR)[0][2]  # both FMAs are within 2 states
    assert len(lint.lint(HAZARD_ONE_APART)) == 1  # one instruction in between is one wait state: still too close
    assert lint.lint(SPACED_BY_NOP) == []
    assert any("EXEC" in p[2] for p in lint.lint(HAZARD_EXEC))
    assert len(lint.lint(HAZARD_SWAP_SECOND)) == 2 and "swap" in lint.lint(HAZARD_SWAP_SECOND)[0][2]
    assert lint.lint(SWAP_ELSEWHERE) == []


def test_lint_follows_branches():
    # producer at the bottom of a loop, DPP FMA at its head: one wait state (the branch) over the back edge
    found = lint.lint(LOOP_BACK_EDGE)
    assert len(found) == 1 and "v_mul_f64" in found[0][2] and "1 wait state" in found[0][2]
    assert lint.lint(LOOP_SPACED) == []  # s_nop + branch = 2 states
    # the fall-through path has three instructions in between, the taken branch only the branch itself
    found = lint.lint(BRANCH_OVER_SPACING)
    assert len(found) == 1 and "1 wait state" in found[0][2]


def _built_objects():
    objects = sorted(glob.glob(os.path.join(ROOT, "tbmodels_amd", "csrc", "*.o")))
    tools = [shutil.which("objcopy"), os.path.join(lint.LLVM, "clang-offload-bundler"), os.path.join(lint.LLVM, "llvm-objdump"),
             os.path.join(lint.LLVM, "llvm-readelf")]
    if not objects or not all(t and os.path.exists(t) for t in tools):
        pytest.skip("no built objects / no LLVM binutils here")
    return objects


def test_built_objects_have_no_dpp_hazard(tmp_path):
    seen, with_dpp = 0, []
    for obj in _built_objects():  # every object: the files that use tbk_dpp.h's FMAs are found, not listed
        text = lint.disassemble(obj, str(tmp_path))
        if text is None:
            continue
        n = lint.count_dpp(text)
        if n == 0:
            continue
        with_dpp.append(os.path.basename(obj))
        seen += n
        problems = lint.lint(text)
        assert problems == [], (obj, problems[:5])
    assert seen > 1000 and "tbk_eig_small.o" in with_dpp and "tbk_eig_band.o" in with_dpp  # the check looked at the real thing


# Kernels allowed to use scratch memory (``.private_segment_fixed_size`` != 0), by substring of the mangled name.  EMPTY:
# a register spill on a hot path has cost 30-90 % wherever it was measured, and nothing else in the
# suite notices when a kernel starts spilling.
SCRATCH_ALLOWED = ()


def test_no_product_kernel_spills(tmp_path):
    found = [entry for entry in lint.spills(_built_objects(), str(tmp_path))
             if not any(allowed in entry[1] for allowed in SCRATCH_ALLOWED)]
    assert found == [], "kernels with scratch memory (object, kernel, bytes per thread, VGPRs): %r" % (found,)
    # the metadata reader sees the real thing: every object with kernels reports their register counts
    res = lint.kernel_resources(os.path.join(ROOT, "tbmodels_amd", "csrc", "tbk_eig_band.o"), str(tmp_path))
    assert any("band_reduce_kernel" in name and r["vgpr"] > 64 for name, r in res.items())
