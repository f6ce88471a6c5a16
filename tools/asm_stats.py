#!/usr/bin/env python3
"""Instruction statistics of one kernel in a hipcc -S listing.

usage: asm_stats.py file.s kernel-substring
Prints per-basic-block instruction counts (largest blocks first) and the opcode mix of
the hottest block, which for the fused kernels is the Picard iteration body.
"""
import re, sys
from collections import Counter

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(tuple([':'])) or (l.startswith('_Z') and key in l and ': ' in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
    blocks, cur, name = [], [], 'entry'
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t or t.startswith(';'):
            continue
        if re.match(r'^\.LBB[\w_]+:', t):
            blocks.append((name, cur)); cur = []; name = t.split(':')[0]
            continue
        if t.startswith('.'):
            continue
        cur.append(t.split()[0])
    blocks.append((name, cur))
    tot = sum(len(b) for _, b in blocks)
    print(f'total instructions: {tot}')
    for n, b in sorted(blocks, key=lambda x: -len(x[1]))[:6]:
        c = Counter(b)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        f64 = sum(v for k, v in c.items() if k.endswith('_f64'))
        print(f'\nblock {n}: {len(b)} instr, VALU {valu}, f64 {f64}, salu {sum(v for k,v in c.items() if k.startswith("s_"))}, ds {sum(v for k,v in c.items() if k.startswith("ds_"))}')
        print('  ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(24)))

main()
