#!/usr/bin/env python3
"""Where a kernel's spilled registers are touched: every scratch_load / scratch_store of the kernel's disassembly, with the
innermost loop (backward branch) it lies in -- "are the spills in the hot loop or at the cold ends?" answered from the
code object instead of by assertion.
usage: spill_sites.py file.o|file.so kernel-name-filter      (e.g. 'k_sdtw_q<8, 25, 0, true>')"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import LLVM, code_objects, demangle        # noqa: E402


def main():
    path, flt = sys.argv[1], sys.argv[2]
    for co in code_objects(path):
        syms = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True,
                              text=True).stdout
        cur, body = None, {}
        for line in syms.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                body[cur] = []
                continue
            if cur and line.strip():
                body[cur].append(line)
        for sym, lines in body.items():
            name = re.sub(r"\(.*\)$", "", demangle(sym).replace("(anonymous namespace)::", "")).replace("void ", "")
            if flt not in name or sym.endswith(".kd"):
                continue
            ins = []                                             # (address, text)
            for ln in lines:
                m = re.match(r"\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
                if m:
                    ins.append((int(m.group(2), 16), m.group(1)))
            addr = {a: i for i, (a, _) in enumerate(ins)}
            loops = []                                           # (first index, last index) of every backward branch
            for i, (a, t) in enumerate(ins):
                m = re.match(r"s_cbranch\w*\s+(\S+)|s_branch\s+(\S+)", t)
                if m:
                    tgt = m.group(1) or m.group(2)
                    mm = re.search(r"<\S+\+0x([0-9a-fA-F]+)>", ln) if False else None
                    # objdump prints the offset as a signed immediate; resolve through the comment-less form
                    try:
                        off = int(tgt, 0)
                    except ValueError:
                        continue
                    off = off - (1 << 16) if off >= (1 << 15) else off
                    ta = a + 4 + 4 * off
                    if ta in addr and ta <= a:
                        loops.append((addr[ta], i))
            print("%s: %d instructions, %d backward branches (loops)" % (name, len(ins), len(loops)))
            sites = [(i, t) for i, (a, t) in enumerate(ins) if t.startswith(("scratch_load", "scratch_store", "buffer_load_dword off", "buffer_store_dword off")) or "scratch_" in t]
            if not sites:
                print("  no scratch instruction")
            for i, t in sites:
                inner = [lp for lp in loops if lp[0] <= i <= lp[1]]
                if inner:
                    lp = min(inner, key=lambda q: q[1] - q[0])
                    where = "INSIDE a loop of %d instructions (%d..%d)" % (lp[1] - lp[0] + 1, lp[0], lp[1])
                else:
                    where = "outside every loop"
                print("  #%-6d %-44s %s" % (i, t[:44], where))
            big = sorted(loops, key=lambda q: q[0] - q[1])[:4]
            print("  largest loops (instruction index ranges): %s" % ", ".join("%d..%d" % q for q in big))


if __name__ == "__main__":
    main()
