#!/usr/bin/env python3
"""Audit for kernels that hide global loads in inline asm (conv_pointwise.hip): between an asm `global_load_dwordx4` and the next
`s_waitcnt vmcnt` no instruction may touch the load's destination registers (hipcc counts an asm load as landed at ;;#ASMEND and is free
to copy the registers — cdna_hip_programming.md §5.7 item 1).  usage: tools/audit_asm_loads.py file.s  (exit 1 on a finding)"""
import re
import sys

txt = open(sys.argv[1]).read()
bad_total = 0
for k in re.split(r"\n(?=_Z[\w]+:)", txt)[1:]:
    name = k.split(":")[0]
    pend, bad = {}, []
    for i, l in enumerate(k.split("\n")):
        m = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\]", l)
        if m:
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                pend[r] = i
            continue
        if "s_waitcnt vmcnt" in l:
            pend = {}
            continue
        t = l.strip()
        if not pend or not t or t[0] in ";.":
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", l):
            regs.update(range(int(a), int(b) + 1))
        regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", l))
        if regs & set(pend):
            bad.append((i, t))
    if bad:
        print(name, ":", len(bad), "instructions touch a load destination before its wait, e.g.", bad[0])
    bad_total += len(bad)
print("audit:", "FAILED" if bad_total else "clean")
sys.exit(1 if bad_total else 0)
