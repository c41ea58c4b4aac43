#!/usr/bin/env python3
"""s_waitcnt vmcnt(N) inside the loops of a kernel (a vmcnt(0) in a pipelined loop drains every load in flight):
    python tools/waits.py viforsdes_amd/csrc/vsde_linear.hip lin_rows_kernelILi256ELi1ELi1ELi4"""
import os, re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
os.makedirs("gpurun_out/tmp", exist_ok=True)
out = "gpurun_out/tmp/waits.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", src, "-o", out],
               check=True, stderr=subprocess.DEVNULL)
s = open(out).read()
for m in re.finditer(r"^(_Z\w*" + re.escape(pat) + r"\w*):", s, re.M):
    body = s[m.end():s.index(".Lfunc_end", m.end())].split("\n")
    print(m.group(1))
    blk, depth, info, order = "entry", {"entry": 0}, {}, []
    for ln in body:
        lm = re.match(r"^(\.LBB\d+_\d+):(.*)", ln) or re.match(r"^; (%bb\.\d+):(.*)", ln)   # (fall-through blocks carry no label)
        if lm:
            blk = lm.group(1)
            dm = re.search(r"Depth=(\d+)", lm.group(2))
            depth[blk] = int(dm.group(1)) if dm else 0
        if blk not in info:
            info[blk] = {"mfma": 0, "vm": [], "gl": 0, "gs": 0, "n": 0}; order.append(blk)
        d = info[blk]; d["n"] += 1
        d["mfma"] += "v_mfma" in ln
        d["gl"] += bool(re.search(r"\b(global_load|buffer_load|scratch_load)", ln))
        d["gs"] += bool(re.search(r"\b(global_store|buffer_store|scratch_store)", ln))
        wm = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", ln)
        if wm:
            d["vm"].append(int(wm.group(1)))
    for b in order:
        d = info[b]
        if depth.get(b, 0) > 0 and (d["vm"] or d["mfma"]):
            print(f"  {b:12s} depth {depth[b]} lines {d['n']:4d} mfma {d['mfma']:3d} loads {d['gl']:2d} stores {d['gs']:2d} vmcnt waits {d['vm']}")
