"""static instruction census of one kernel by source region: asm_regions.py <file.s> <kernel label substring> [line:name ...]
The .s comes from `hipcc -S --cuda-device-only -gline-tables-only`; instructions under helper lines (< first region) are charged to the
region seen last."""
import sys, re, collections
path, label = sys.argv[1], sys.argv[2]
regions = [(int(a.split(':')[0]), a.split(':')[1]) for a in sys.argv[3:]]
regions.sort()
def region_of(line):
    name = None
    for l0, n in regions:
        if line >= l0: name = n
    return name
cnt = collections.defaultdict(lambda: collections.Counter())
inside = False; cur = None; fileno = None
for ln in open(path):
    s = ln.strip()
    if not inside:
        if s.startswith('_Z') and label in s and s.split(':')[0].endswith(tuple('EhPKv')) and ': ' in s: inside = True
        continue
    if s.startswith('.Lfunc_end'): break
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        if int(m.group(1)) == 1:
            r = region_of(int(m.group(2)))
            if r is not None and int(m.group(2)) >= regions[0][0]: cur = r
        continue
    if not s or s.startswith(('.', ';')) or s.endswith(':'): continue
    op = s.split()[0]
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): kind = 'wait'
    if op.startswith('scratch_'): kind = 'scratch'
    cnt[cur][kind] += 1
tot = collections.Counter()
print('%-28s %6s %6s %6s %6s %6s %6s' % ('region', 'valu', 'salu', 'lds', 'vmem', 'scratch', 'wait'))
for r in [n for _, n in regions] + [None]:
    c = cnt.get(r)
    if not c: continue
    print('%-28s %6d %6d %6d %6d %6d %6d' % (r, c['valu'], c['salu'], c['lds'], c['vmem'], c['scratch'], c['wait']))
    tot.update(c)
print('%-28s %6d %6d %6d %6d %6d %6d' % ('total', tot['valu'], tot['salu'], tot['lds'], tot['vmem'], tot['scratch'], tot['wait']))
