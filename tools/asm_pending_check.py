#!/usr/bin/env python3
"""Static check of a kernel's assembly for hand-counted asm loads (csrc: attn_fwd8_kernel, deep256_kernel): between an inline-asm
`global_load_dwordx4 vDST, vADDR, off` and the next `s_waitcnt vmcnt(...)` written in inline asm, no instruction may name a register of
vDST -- the compiler believes the value is there already, and a copy or re-use in that window reads / clobbers data in flight.
Linear scan (branches ignored), counted waits retire the oldest loads.
    hipcc ... -save-temps ; python tools/asm_pending_check.py file.s kernel_symbol_substring"""
import re, sys

def main(path, sym):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    L = lines[start:end]
    pend, bad = [], 0            # list of (set(regs), line) in issue order
    for i, l in enumerate(L):
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        in_asm = i > 0 and L[i - 1].strip() == ";;#ASMSTART"
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if m:
            keep = int(m.group(1))
            pend = pend[max(len(pend) - keep, 0):] if keep else []   # (a compiler-written wait counts its own, younger loads too: conservative)
            continue
        regs = set()
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
            regs |= set(range(int(a), int(b) + 1))
        regs |= {int(a) for a in re.findall(r"\bv(\d+)\b", t)}
        m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v", t)
        allp = set().union(*[r for r, _ in pend]) if pend else set()
        if m and in_asm:
            d = set(range(int(m.group(1)), int(m.group(2)) + 1))
            if (regs - d) & allp:
                print(f"line {start + i + 1}: address uses a pending register: {t}"); bad += 1
            pend.append((d, i))
            continue
        if regs & allp:
            print(f"line {start + i + 1}: touches pending {sorted(regs & allp)[:4]}: {t}"); bad += 1
    print(f"{sym}: {bad} violations")
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
