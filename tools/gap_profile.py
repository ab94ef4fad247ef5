#!/usr/bin/env python3
"""Instruction mix between consecutive MFMAs in the hot blocks of a kernel's ISA (-save-temps .s).
usage: tools/gap_profile.py FILE.s KERNEL_SUBSTRING [min_mfma]"""
import re, sys, collections
src = open(sys.argv[1]).read()
sub = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for m in re.finditer(r'^(_Z\S+):(.*?)\n\.Lfunc_end', src, re.S | re.M):
    if sub not in m.group(1):
        continue
    print(m.group(1))
    for b in re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2)):
        lines = [l.strip() for l in b.split('\n')[1:] if l.strip() and not l.strip().startswith(('.', ';'))]
        if sum('mfma' in x for x in lines) < min_mfma:
            continue
        gaps, cur = [], []
        for l in lines:
            if 'mfma' in l:
                gaps.append(cur); cur = []
            else:
                cur.append(l)
        gaps.append(cur)
        out = []
        for i, g in enumerate(gaps):
            c = collections.Counter('V' if x.startswith('v_') else 'D' if x.startswith('ds') else 'N' if x.startswith('s_nop')
                                    else 'W' if x.startswith('s_waitcnt') else 'G' if x.startswith(('global', 'scratch')) else 'S' for x in g)
            out.append(f"{i}:" + ''.join(f"{k}{v}" for k, v in sorted(c.items())))
        print(' ', b.split('\n')[0].strip(), ' '.join(out))
        print('   v_mov:', sum(1 for l in lines if l.startswith('v_mov')), ' scratch:', sum(1 for l in lines if 'scratch' in l),
              ' s_nop:', sum(1 for l in lines if l.startswith('s_nop')))
