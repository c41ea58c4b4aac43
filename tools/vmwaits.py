#!/usr/bin/env python3
"""Order of the vector-memory operations, their waits and the barriers inside the innermost loop of one kernel of an ISA listing
(hipcc -S --cuda-device-only ...): on gfx9 loads AND stores count in vmcnt, so a wait for a load that sits behind output stores
drains those stores -- the pattern this tool makes visible.
    python tools/vmwaits.py gpurun_out/tmp/lin.s _ZN4vsde15lin_rows_kernelILi256ELi1ELi1ELi4EEEvNS_9LinParamsE"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
a = s.index(name + ":")
lines = s[a:s.index(".Lfunc_end", a)].splitlines()
labels = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)]
hdr = [l for l in lines if "Inner Loop Header" in l][0].split(":")[0]
st = [i for i, l in enumerate(lines) if l.startswith(hdr + ":")][0]
inloop = [i for i in labels if "Header=" + hdr.replace(".LBB", "BB") in lines[i]]
nxt = [i for i in labels if i > inloop[-1]]
loop = lines[st:nxt[0] if nxt else len(lines)]
out, mf = [], 0
for i, l in enumerate(loop):
    t = l.strip()
    if "v_mfma" in t:
        mf += 1
        continue
    if re.search(r"global_store|global_load|buffer_|s_waitcnt vmcnt|s_barrier", t):
        if mf:
            out.append(f"      [{mf} mfma]")
            mf = 0
        out.append(f"{i:5d} {' '.join(t.split()[:2]) if 'waitcnt' in t else t.split()[0]}")
print("\n".join(out))
print(len(loop), "lines in the loop")
