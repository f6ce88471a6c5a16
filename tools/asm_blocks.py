#!/usr/bin/env python3
"""Per-basic-block opcode summary of one kernel in a hipcc -S listing (every block of >= MIN instructions, in program order).
usage: asm_blocks.py file.s kernel-substring [MIN]"""
import re, sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and ':' in l)
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
print(len(blocks), "blocks,", sum(len(b) for _, b in blocks), "instructions")
for n, b in blocks:
    if len(b) >= mn:
        c = Counter(b)
        grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(f"{n:14s} {len(b):5d}  readlane {c['v_readlane_b32']:4d} writelane {c['v_writelane_b32']:4d} mfma {grp('v_mfma'):4d} "
              f"f64 {sum(v for k, v in c.items() if '_f64' in k and not k.startswith('v_mfma')):4d} dpp {sum(v for k, v in c.items() if k.endswith('_dpp')):4d} "
              f"acc {grp('v_accvgpr'):4d} s_nop {c['s_nop']:4d} ds {grp('ds_'):3d} scratch {grp('scratch_'):3d} barrier {c['s_barrier']}")
