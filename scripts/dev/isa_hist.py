"""Dev only: instruction histogram per kernel of a gfx950 .s file (hipcc -save-temps).  usage: isa_hist.py file.s [name-substring]"""
import sys, re, collections
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ''
for f in re.split(r'\n(?=\S+:\s*; @)', s):
    m = re.match(r'(\S+):', f)
    if not m or want not in m.group(1):
        continue
    c = collections.Counter()
    for line in f.split('\n'):
        t = line.strip().split()
        if t and not t[0].startswith(('.', ';')) and not t[0].endswith(':'):
            c[t[0]] += 1
    print(m.group(1)[:80], 'instructions:', sum(c.values()))
    groups = collections.Counter()
    for k, v in c.items():
        g = ('mfma' if 'mfma' in k else 'accvgpr' if 'accvgpr' in k else 'scratch' if k.startswith('scratch') else 'ds_read_tr' if 'tr_b16' in k else
             'ds_read' if k.startswith('ds_read') else 'ds_write' if k.startswith('ds_write') else 'global' if k.startswith(('global', 'buffer')) else
             'waitcnt' if k == 's_waitcnt' else 'barrier' if k == 's_barrier' else 's_nop' if k == 's_nop' else 'salu' if k.startswith('s_') else 'valu')
        groups[g] += v
    print('   ', dict(groups))
    if '-v' in sys.argv:
        for k, v in c.most_common(45):
            print('      ', k, v)
