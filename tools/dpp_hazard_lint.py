#!/usr/bin/env python3
"""
Static check of the BUILT gfx950 code objects for the data hazards of inline-asm DPP instructions.

``tbmodels_amd/csrc/tbk_dpp.h`` issues ``v_fmac_f64_dpp ... row_newbcast`` through inline asm.  LLVM's hazard
recogniser does not look inside asm blocks, and gfx9 hardware does not interlock:

* a VALU instruction that writes a VGPR, followed by a DPP instruction that reads that VGPR through the DPP path
  (its first source), needs 2 wait states in between;
* a VALU instruction that writes EXEC, followed by any DPP instruction, needs 5.

The walk follows control flow: local labels come from ``llvm-objdump --symbolize-operands``, a label's predecessors are
the fall-through instruction and every branch that names it (loop back edges included), and ``v_permlane16_swap`` /
``v_permlane32_swap`` / ``v_swap_b32`` count as writers of BOTH their operands.  The same pass reads the code objects'
metadata and reports every kernel with a private segment (``.private_segment_fixed_size`` != 0: register spills).

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


def unbundle(obj_path, workdir):
    """Path of the gfx950 code object inside a host object / shared library, or None."""
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
    return code


def disassemble(obj_path, workdir):
    """llvm-objdump text of the gfx950 code object inside a host object / shared library, or None.  Branch targets are
    symbolised (``--symbolize-operands``: ``s_cbranch_scc1 L7`` and a ``<L7>:`` line at the target) -- plain ``-d`` prints
    no local labels at all, and the scan then runs straight through branches and never sees a loop's back edge."""
    code = unbundle(obj_path, workdir)
    if code is None:
        return None
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", code], check=True,
                          stdout=subprocess.PIPE, universal_newlines=True).stdout


def kernel_resources(obj_path, workdir):
    """{kernel symbol: {"scratch": private_segment_fixed_size, "vgpr": .vgpr_count, "sgpr": .sgpr_count}} from the code
    object's metadata notes."""
    code = unbundle(obj_path, workdir)
    if code is None:
        return {}
    text = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code], check=True, stdout=subprocess.PIPE,
                          universal_newlines=True).stdout
    out, fields = {}, {}
    for line in text.splitlines():
        found = re.match(r"\s*-?\s*\.(name|private_segment_fixed_size|vgpr_count|sgpr_count|agpr_count):\s*(\S+)", line)
        if not found:
            if line.strip().startswith("- .") and fields.get("name"):  # next kernel record
                pass
            continue
        key, value = found.group(1), found.group(2)
        if key == "name" and "name" in fields and "private_segment_fixed_size" in fields:
            out[fields["name"]] = fields
            fields = {}
        fields[key] = value if key == "name" else int(value)
        if all(k in fields for k in ("name", "private_segment_fixed_size", "vgpr_count", "sgpr_count")):
            out[fields["name"]] = {"scratch": fields["private_segment_fixed_size"], "vgpr": fields["vgpr_count"],
                                   "sgpr": fields["sgpr_count"], "agpr": fields.get("agpr_count", 0)}
            fields = {}
    return out


_UNCONDITIONAL = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")
_TWO_DESTINATIONS = ("v_permlane16_swap", "v_permlane32_swap", "v_swap_b32")


def parse(text):
    """[(kernel, instrs, labels)]: instrs = [(mnemonic, [operands])...]; labels = {name: index of the instruction the label
    stands in front of}."""
    kernels = []
    current = None
    for line in text.splitlines():
        head = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if head:
            name = head.group(1)
            if re.match(r"^\.?L\d+$", name):  # a local label inside a function: a branch target
                if current is not None:
                    current[2][name] = len(current[1])
            else:
                current = (name, [], {})
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


def _written_vgprs(mnemonic, operands):
    """VGPRs a VALU instruction writes: its first operand -- and the second one too for the swaps (v_permlane16_swap /
    v_permlane32_swap / v_swap_b32 exchange lanes of BOTH registers)."""
    if not operands:
        return set()
    written = _vgprs(operands[0].split()[0])
    if mnemonic.startswith(_TWO_DESTINATIONS) and len(operands) > 1:
        written |= _vgprs(operands[1].split()[0])
    return written


def _is_dpp_fma(mnemonic, operands):
    return mnemonic.startswith("v_fmac_f64_dpp") or (mnemonic.startswith("v_fmac_f64") and any("row_newbcast" in o for o in operands))


def lint(text):
    """List of (kernel, index, message) for every DPP FMA with a too-close VALU producer on ANY path into it: the walk
    goes backwards through fall-through predecessors and through every branch that targets a label on the way."""
    problems = []
    for kernel, instrs, labels in parse(text):
        targets = {}  # instruction index -> indices of the branches that jump there
        for idx, (mnemonic, operands) in enumerate(instrs):
            if (mnemonic.startswith("s_cbranch") or mnemonic == "s_branch") and operands and operands[0] in labels:
                targets.setdefault(labels[operands[0]], []).append(idx)

        def predecessors(idx):
            preds = list(targets.get(idx, []))
            if idx > 0 and not instrs[idx - 1][0].startswith(_UNCONDITIONAL):
                preds.append(idx - 1)
            return preds

        for idx, (mnemonic, operands) in enumerate(instrs):
            if not _is_dpp_fma(mnemonic, operands) or len(operands) < 2:
                continue
            # operands: dst, src0 (the DPP-read one, may carry a neg modifier), src1 + dpp controls
            dpp_src = _vgprs(operands[1].split()[0])
            seen = set()
            stack = [(p, 0) for p in predecessors(idx)]
            while stack:
                at, states = stack.pop()
                if (at, states) in seen or states >= 5:
                    continue
                seen.add((at, states))
                prev, prev_ops = instrs[at]
                if prev == "s_nop":
                    after = states + (int(prev_ops[0], 0) + 1 if prev_ops else 1)
                else:
                    if prev.startswith("v_") and prev_ops:
                        if states < 2 and _written_vgprs(prev, prev_ops) & dpp_src and not prev.startswith("v_cmp"):
                            problems.append((kernel, idx, "%s writes %s %d wait state(s) before %s reads it through DPP"
                                             % (prev, ", ".join(prev_ops[:2]), states, mnemonic)))
                        if prev.startswith("v_cmpx") or prev_ops[0].split()[0] in ("exec", "exec_lo", "exec_hi"):
                            problems.append((kernel, idx, "%s writes EXEC %d wait state(s) before %s" % (prev, states, mnemonic)))
                    after = states + 1
                for p in predecessors(at):
                    stack.append((p, after))
    return sorted(set(problems), key=lambda item: (item[0], item[1], item[2]))


def count_dpp(text):
    return sum(1 for _, instrs, _ in parse(text) for m, ops in instrs if _is_dpp_fma(m, ops))


def spills(objects, workdir):
    """[(object, kernel, scratch bytes, vgprs)] for every kernel with a private segment (spills or indexed local arrays)."""
    found = []
    for obj in objects:
        for name, res in sorted(kernel_resources(obj, workdir).items()):
            if res["scratch"] != 0:
                found.append((os.path.basename(obj), name, res["scratch"], res["vgpr"]))
    return found


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
        spilled = spills(objects, workdir)
    print("total: %d DPP FMAs, %d hazard(s)" % (total, bad))
    for obj, name, scratch, vgpr in spilled:
        print("scratch: %s %s: %d B per thread (%d VGPRs)" % (obj, name, scratch, vgpr))
    return 1 if bad or spilled else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
