#!/usr/bin/env python3
"""Static instruction mix of one kernel by source line, from `hipcc -S -gline-tables-only --cuda-device-only`.
usage: tools/isa_lines.py file.s kernel-substring [first_line last_line]"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 10**9
cnt = collections.defaultdict(lambda: collections.Counter())
infn = False; line = 0
def cls(op):
    if op.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_sleep')): return 'wait'
    if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc', 's_endpgm')): return 'br'
    if op.startswith(('s_load', 's_buffer_load', 's_store', 's_memtime', 's_dcache')): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'vlane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_')): return 'vmem'
    if op.startswith('scratch_'): return 'scratch'
    return 'other'
for l in open(path):
    t = l.strip()
    if t.startswith('.type') and '@function' in t:
        infn = kern in t
        continue
    if not infn: continue
    m = re.match(r'\.loc\s+\d+\s+(\d+)', t)
    if m: line = int(m.group(1)); continue
    if not t or t[0] in '.;' or t.endswith(':'): continue
    op = t.split()[0]
    cnt[line][cls(op)] += 1
tot = collections.Counter()
rows = []
for ln, c in cnt.items():
    if lo <= ln <= hi:
        tot.update(c); rows.append((ln, c))
print('total', dict(tot))
rows.sort(key=lambda r: -(r[1]['salu'] + r[1]['vlane']))
src = open('/root/repo/xz_amd/csrc/lzma_kernels.hip').read().split('\n') if len(sys.argv) <= 5 else []
for ln, c in rows[:70]:
    s = src[ln - 1].strip()[:90] if src and ln - 1 < len(src) else ''
    print(f"{ln:5d} salu {c['salu']:4d} vlane {c['vlane']:3d} valu {c['valu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} scr {c['scratch']:3d} br {c['br']:3d} wait {c['wait']:3d} | {s}")
