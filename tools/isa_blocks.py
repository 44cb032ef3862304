#!/usr/bin/env python
"""Per-basic-block instruction mix of one kernel in a gfx950 .s file.
usage: hipcc ... -S --cuda-device-only -o k.s file.hip ; python tools/isa_blocks.py k.s <substring of mangled name>"""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w*:', l) and pat in l][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
body = [l.strip() for l in lines[start + 1:end]]
blocks = [["entry", []]]
for l in body:
    if l.startswith('.LBB') and ':' in l:
        blocks.append([l.split(':')[0], []])
    elif l and not l.startswith((';', '.')):
        blocks[-1][1].append(l.split(';')[0].strip())
tot = 0
for name, b in blocks:
    if not b: continue
    c = collections.Counter(x.split()[0] for x in b)
    f64 = sum(v for k, v in c.items() if 'f64' in k)
    dpp = sum(1 for x in b if 'row_ror' in x or 'quad_perm' in x or '_dpp' in x)
    ds = sum(v for k, v in c.items() if k.startswith('ds_'))
    gl = sum(v for k, v in c.items() if k.startswith(('global_', 'flat_', 'buffer_')))
    sal = sum(v for k, v in c.items() if k.startswith('s_'))
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    tot += len(b)
    print(f"{name:12s} n={len(b):4d} valu={valu:4d} f64={f64:4d} dpp={dpp:3d} ds={ds:3d} vmem={gl:3d} salu={sal:3d} waitcnt={c.get('s_waitcnt',0)}")
print("total", tot)
