#!/usr/bin/env python3
"""Order of VMEM loads and vmcnt waits of one kernel between two source lines (from hipcc -S -gline-tables-only).
usage: tools/isa_waits.py file.s kernel-substring first_line last_line"""
import re, sys
path, kern, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
infn = False; line = 0; n = 0
for l in open(path):
    t = l.strip()
    if t.startswith('.type') and '@function' in t:
        infn = kern in t; continue
    if not infn: continue
    m = re.match(r'\.loc\s+\d+\s+(\d+)', t)
    if m: line = int(m.group(1)); continue
    if not t or t[0] in '.;': continue
    if t.endswith(':') or ': ' in t[:20] and t.startswith('.LBB'):
        continue
    n += 1
    if not (lo <= line <= hi): continue
    op = t.split()[0]
    if op.startswith(('global_load', 'buffer_load', 'scratch_load')) or ('vmcnt' in t) or op.startswith('s_cbranch') and False:
        print(f"{n:6d} L{line:5d} {t[:100]}")
