#!/usr/bin/env python3
"""Count instructions of the regular check-node body with N consecutive ds_read_u8 in a kernel's ISA dump."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
N = int(sys.argv[2])
i = 0
while i < len(lines):
    if 'ds_read_u8' in lines[i]:
        j = i
        while j < len(lines) and 'ds_read_u8' in lines[j]: j += 1
        if j - i == N:
            # walk back to the block start and forward to N-th ds_write_b8
            a = i
            while a > 0 and not lines[a].startswith('.LBB') and 's_cbranch' not in lines[a] and '; %bb' not in lines[a]: a -= 1
            b = j; w = 0
            while b < len(lines) and w < N:
                if 'ds_write_b8' in lines[b]: w += 1
                b += 1
            ops = [l.split()[0] for l in lines[a:b] if re.match(r'^\s+[vsd][a-z_]', l)]
            c = collections.Counter(ops)
            v = sum(n for o, n in c.items() if o.startswith('v_')); sc = sum(n for o, n in c.items() if o.startswith('s_'))
            print(f"lines {a}-{b}: total {len(ops)} VALU {v} SALU/other-s {sc} ds {sum(n for o,n in c.items() if o.startswith('ds_'))}")
            print('  ', ', '.join(f"{o}:{n}" for o, n in c.most_common(24)))
            break
        i = j
    else:
        i += 1
