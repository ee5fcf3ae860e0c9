"""Instruction mix of the (first) inner loop of a kernel in a hipcc -S listing.
usage: python tools/isa_loop.py file.s <substring of the mangled kernel name> [print]"""
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(lines) if l.startswith('_Z') and name in l and l.rstrip().endswith(')') is False and ':' in l][0]
end = [i for i in range(start, len(lines)) if 's_endpgm' in lines[i]][0]
body = lines[start:end]
hdrs = [i for i, l in enumerate(body) if 'Loop Header' in l and 'Depth=1' in l]
best = None
for hdr in hdrs:                      # the loop with the most MFMAs
    lab = body[hdr].split(':')[0]
    backs = [i for i, l in enumerate(body) if i > hdr and l.strip().startswith(('s_branch', 's_cbranch')) and l.split()[-1] == lab]
    if not backs:
        continue
    n = sum('mfma' in l for l in body[hdr:backs[-1] + 1])
    if best is None or n > best[0]:
        best = (n, hdr, backs[-1], lab)
_, hdr, back, lab = best
loop = [l.strip() for l in body[hdr:back + 1] if l.strip() and not l.strip().startswith((';', '.'))]
c = Counter()
for l in loop:
    op = l.split()[0]
    k = ('mfma' if 'mfma' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write') else
         'vmem' if op.startswith(('buffer_', 'global_')) else 'waitcnt' if op == 's_waitcnt' else 'nop' if op == 's_nop' else
         'branch' if op.startswith(('s_cbranch', 's_branch')) else 'salu' if op.startswith('s_') else
         'accvgpr' if 'accvgpr' in op else 'valu' if op.startswith('v_') else op)
    c[k] += 1
print(name, ':', len(loop), 'instructions in loop', lab, dict(c))
print('  waits:', [l for l in loop if l.startswith(('s_waitcnt', 's_nop'))])
if len(sys.argv) > 3:
    print('\n'.join(loop))
