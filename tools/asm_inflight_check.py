#!/usr/bin/env python3
"""Static check for kernels that issue global loads from inline asm (attention.hip's att_load: hipcc does not count those loads,
so nothing may read, write or spill their destination registers before the kernel's own `s_waitcnt vmcnt(0)`).
usage: asm_inflight_check.py <file.s> <kernel-name substring>   -> prints "asm loads N bad M"; exit status 1 if M > 0.
tests/test_build_asm.py compiles attention.hip to assembly and runs this on every kernel that uses att_load."""
import re, sys


def check(path, pat):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^\S*" + pat + r"\S*:", l)][0]
    end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
    infl, bad, inasm, nasm = set(), [], False, 0
    for l in lines[start:end]:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        regs = []
        for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", t):
            regs += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
        if inasm and t.startswith("global_load"):
            m = re.search(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", t)   # first operand = destination
            infl.update(range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))])
            nasm += 1
            continue
        if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            infl = set()
            continue
        if any(r in infl for r in regs):
            bad.append(t)
    return nasm, bad


if __name__ == "__main__":
    n, bad = check(sys.argv[1], sys.argv[2])
    for t in bad[:20]:
        print("TOUCH", t)
    print("asm loads", n, "bad", len(bad))
    sys.exit(1 if bad else 0)
