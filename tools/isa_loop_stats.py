#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -save-temps .s file (tuning aid).

    python tools/isa_loop_stats.py <file.s> <kernel-name-substring>
"""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2]
    start = None
    for m in re.finditer(r'^(\S+):\s*; @(\S+)', text, re.M):
        if pat in m.group(1):
            start = m
            break
    if not start:
        sys.exit("kernel not found")
    end = text.index('.end_amdhsa_kernel', start.end()) if '.end_amdhsa_kernel' in text[start.end():] else len(text)
    nxt = re.search(r'^\s*s_endpgm', text[start.end():], re.M)
    body = text[start.end(): start.end() + nxt.end()] if nxt else text[start.end():end]
    lines = [l.split(';')[0].strip() for l in body.split('\n')]
    lines = [l for l in lines if l and (l.endswith(':') or not l.startswith('.'))]
    print(start.group(1)[:60], len(lines), 'lines')
    lab = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\S+)|s_branch (\S+)', l)
        if not mm:
            continue
        t = mm.group(1) or mm.group(2)
        if t in lab and lab[t] < i:
            seg = lines[lab[t]:i + 1]
            c = Counter()
            for x in seg:
                op = x.split()[0]
                if op.endswith(':'): continue
                if op.startswith('v_mfma'): c['mfma'] += 1
                elif op.startswith('v_'): c['valu'] += 1
                elif op.startswith('s_waitcnt'): c['waitcnt'] += 1
                elif op.startswith('s_'): c['salu'] += 1
                elif op.startswith('ds_'): c['lds'] += 1
                elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['vmem'] += 1
                else: c['other'] += 1
            print('loop', t, 'instructions', len(seg), dict(c))


if __name__ == '__main__':
    main()
