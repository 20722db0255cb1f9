#!/usr/bin/env python3
"""asmblocks.py <file.s> — per-basic-block instruction mix of a gfx950 ISA dump (scratch traffic, FMAs, LDS ops, barriers)."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
blocks = []; cur = None
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        if cur: blocks.append(cur)
        cur = {'n': m.group(1), 'i': 0, 'sl': 0, 'ss': 0, 'fma': 0, 'ds': 0, 'bar': 0, 'gl': 0}
        continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    cur['i'] += 1
    if 'scratch_load' in t: cur['sl'] += 1
    if 'scratch_store' in t: cur['ss'] += 1
    if 'v_fma_f64' in t or 'v_fmac_f64' in t: cur['fma'] += 1
    if t.startswith('ds_'): cur['ds'] += 1
    if 's_barrier' in t: cur['bar'] += 1
    if t.startswith('global_'): cur['gl'] += 1
if cur: blocks.append(cur)
for b in blocks:
    if b['sl'] + b['ss'] >= thr or b['fma'] > 30:
        print(b)
