#!/usr/bin/env python3
"""
Static check of the BUILT gfx950 code objects for the data hazards of inline-asm DPP instructions.

``tbmodels_amd/csrc/tbk_dpp.h`` issues ``v_fmac_f64_dpp ... row_newbcast`` through inline asm.  LLVM's hazard
recogniser does not look inside asm blocks, and gfx9 hardware does not interlock:

* a VALU instruction that writes a VGPR, followed by a DPP instruction that reads that VGPR through the DPP path
  (its first source), needs 2 wait states in between;
* a VALU instruction that writes EXEC, followed by any DPP instruction, needs 5.

Putting ``s_nop`` into the asm string costs +10 % on the n <= 64 reduction (measured), so the built objects are
checked instead: every ``*.o`` of ``tbmodels_amd/csrc`` is unbundled (``objcopy`` of ``.hip_fatbin`` +
``clang-offload-bundler``), disassembled (``llvm-objdump -d``), and each DPP FMA's predecessors are inspected.  Wait
states are counted conservatively as instructions: one per preceding instruction, ``s_nop N`` counting N + 1.

    python tools/dpp_hazard_lint.py [object files...]      exit status 1 and a listing when a hazard is found
"""

import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("TBK_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def _vgprs(operand):
    """The VGPR numbers an operand like ``v[12:13]``, ``-v[4:5]`` or ``v7`` names."""
    found = _REG.search(operand)
    if not found:
        return set()
    if found.group(3) is not None:
        return {int(found.group(3))}
    return set(range(int(found.group(1)), int(found.group(2)) + 1))


def disassemble(obj_path, workdir):
    """llvm-objdump text of the gfx950 code object inside a host object / shared library, or None."""
    fatbin = os.path.join(workdir, "x.fatbin")
    code = os.path.join(workdir, "x.co")
    for path in (fatbin, code):
        if os.path.exists(path):
            os.unlink(path)
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj_path, fatbin], check=True)
    if not os.path.exists(fatbin) or os.path.getsize(fatbin) == 0:
        return None
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fatbin,
                    "--targets=" + TARGET, "--output=" + code], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code], check=True, stdout=subprocess.PIPE,
                          universal_newlines=True).stdout


def parse(text):
    """[(kernel, [(mnemonic, [operands])...])]: instructions per function, branch targets marked as ('<label>', [])."""
    kernels = []
    current = None
    for line in text.splitlines():
        head = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if head:
            name = head.group(1)
            if name.startswith("L") or name.startswith(".L"):  # a local label inside a function: a control-flow merge
                if current is not None:
                    current[1].append(("<label>", []))
            else:
                current = (name, [])
                kernels.append(current)
            continue
        if current is None or "//" not in line:
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.split(None, 1)
        operands = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        current[1].append((parts[0], operands))
    return kernels


def lint(text):
    """List of (kernel, index, message) for every DPP FMA with a too-close VALU producer."""
    problems = []
    for kernel, instrs in parse(text):
        for idx, (mnemonic, operands) in enumerate(instrs):
            if not (mnemonic.startswith("v_fmac_f64_dpp") or mnemonic.startswith("v_fmac_f64") and any("row_newbcast" in o for o in operands)):
                continue
            # operands: dst, src0 (the DPP-read one, may carry a neg modifier), src1 + dpp controls
            if len(operands) < 2:
                continue
            dpp_src = _vgprs(operands[1].split()[0])
            states = 0
            for back in range(idx - 1, max(-1, idx - 8), -1):
                prev, prev_ops = instrs[back]
                if prev == "<label>":
                    break  # predecessors on other paths are not visible here: the compiler's own scheduling applies
                if prev == "s_nop":
                    states += int(prev_ops[0], 0) + 1 if prev_ops else 1
                    continue
                if prev.startswith("v_") and prev_ops:
                    written = _vgprs(prev_ops[0].split()[0])
                    if states < 2 and written & dpp_src and not prev.startswith("v_cmp"):
                        problems.append((kernel, idx, "%s writes %s %d wait state(s) before %s reads it through DPP"
                                         % (prev, prev_ops[0], states, mnemonic)))
                    if states < 5 and (prev.startswith("v_cmpx") or prev_ops[0].split()[0] in ("exec", "exec_lo", "exec_hi")):
                        problems.append((kernel, idx, "%s writes EXEC %d wait state(s) before %s" % (prev, states, mnemonic)))
                states += 1
                if states >= 5:
                    break
    return problems


def count_dpp(text):
    return sum(1 for _, instrs in parse(text) for m, ops in instrs if m.startswith("v_fmac_f64") and any("row_newbcast" in o for o in ops))


def main(argv):
    objects = argv or sorted(glob.glob(os.path.join(ROOT, "tbmodels_amd", "csrc", "*.o")))
    if not objects:
        print("no object files (build first: make -C tbmodels_amd/csrc)")
        return 2
    total, bad = 0, 0
    with tempfile.TemporaryDirectory() as workdir:
        for obj in objects:
            text = disassemble(obj, workdir)
            if text is None:
                continue
            n = count_dpp(text)
            problems = lint(text)
            total += n
            bad += len(problems)
            print("%-28s %6d DPP FMAs, %d hazard(s)" % (os.path.basename(obj), n, len(problems)))
            for kernel, idx, message in problems[:20]:
                print("    %s [%d]: %s" % (kernel[:60], idx, message))
    print("total: %d DPP FMAs, %d hazard(s)" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
