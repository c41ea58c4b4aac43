#!/usr/bin/env python3
"""Instruction mix of every loop (backward branch) of every kernel in a gfx950 assembly file:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only x.hip -o x.s; tools/loopmix.py x.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
for k, (pos, name) in enumerate(starts):
    if flt and flt not in name:
        continue
    body = s[pos:starts[k + 1][0] if k + 1 < len(starts) else len(s)]
    body = body.split('.Lfunc_end')[0]
    lines = body.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    cnt_all = lambda pat: sum(1 for x in lines if re.search(pat, x))
    print(name, 'lines', len(lines), 'mfma', cnt_all('v_mfma'), 'scratch', cnt_all('scratch_'))
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = lines[labels[m.group(1)]:i]
            cnt = lambda pat: sum(1 for x in seg if re.search(pat, x))
            print('  loop', m.group(1), 'len', len(seg), 'mfma', cnt('v_mfma'), 'scratch', cnt('scratch_'), 'ds_read', cnt('ds_read'),
                  'ds_write', cnt('ds_write'), 'valu', cnt(r'^\s+v_(?!mfma)'), 'accvgpr', cnt('v_accvgpr'), 'trans', cnt(r'v_(exp|rcp|rsq|log)'),
                  'nop', cnt('s_nop'), 'waitcnt', cnt('s_waitcnt'), 'global', cnt(r'global_|buffer_'))
