#!/usr/bin/env python3
"""Where a kernel's register spills sit: scratch loads / stores and MFMAs per basic block of the gfx950 assembly.
    python tools/spills.py viforsdes_amd/csrc/vsde_attn.hip attn_fwd_kernelILb1ELi0"""
import os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
os.makedirs("gpurun_out/tmp", exist_ok=True)
out = "gpurun_out/tmp/spills.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", src, "-o", out], check=True,
               stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M):
    name = m.group(1)
    body = s[m.end():s.index(".Lfunc_end", m.end())].split("\n")
    print(name, len(body), "lines")
    blk, stats, order, depth = "entry", {}, [], {}
    for ln in body:
        lm = re.match(r"^(\.LBB\d+_\d+):(.*)", ln) or re.match(r"^; (%bb\.\d+):(.*)", ln)   # (fall-through blocks carry no label)
        if lm:
            blk = lm.group(1)
            dm = re.search(r"Depth=(\d+)", lm.group(2))
            depth[blk] = int(dm.group(1)) if dm else 0
        if blk not in stats:
            stats[blk] = [0, 0, 0, 0]; order.append(blk)
        st = stats[blk]
        st[0] += "scratch_store" in ln; st[1] += "scratch_load" in ln; st[2] += "v_mfma" in ln; st[3] += 1
    for b in order:
        st = stats[b]
        if st[0] or st[1] or st[2]:
            print(f"  {b:12s} loop depth {depth.get(b, 0)} lines {st[3]:5d} mfma {st[2]:3d} scratch_store {st[0]:3d} scratch_load {st[1]:3d}")
