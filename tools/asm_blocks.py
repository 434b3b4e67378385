"""Dev tool: per-basic-block instruction-class summary of one kernel in a hipcc -S listing.
usage: python tools/asm_blocks.py file.s kernel_name_regex [--body BLOCK]"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(%s[^:\n]*):' % sys.argv[2], s, re.M)
body = s[m.end():s.index('.Lfunc_end', m.end())]
lines = [l.strip().split(';')[0].strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(';')]
lines = [l for l in lines if l]
if len(sys.argv) > 4 and sys.argv[3] == '--body':
    on = False
    for l in lines:
        if l.endswith(':'):
            on = l[:-1] == sys.argv[4]
        if on:
            print(l)
    sys.exit(0)
cur = None
blocks = []
def new(name): return dict(name=name, n=0, mfma=0, valu=0, gl=0, gs=0, dsr=0, dsw=0, wait=[], br=[])
cur = new('entry')
for l in lines:
    if l.endswith(':') and l.startswith('.LBB'):
        blocks.append(cur); cur = new(l[:-1]); continue
    if l.startswith('.'): continue
    cur['n'] += 1
    op = l.split()[0]
    if op.startswith('v_mfma'): cur['mfma'] += 1
    elif op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('global_load'): cur['gl'] += 1
    elif op.startswith('global_store') or op.startswith('global_atomic'): cur['gs'] += 1
    elif op.startswith('ds_read'): cur['dsr'] += 1
    elif op.startswith('ds_write'): cur['dsw'] += 1
    elif op == 's_waitcnt': cur['wait'].append(l.split(None, 1)[1])
    elif op.startswith('s_cbranch') or op == 's_branch': cur['br'].append(op[2:] + '->' + l.split()[1])
blocks.append(cur)
minn = 12
for c in blocks:
    if c['n'] >= minn:
        print(f"{c['name']:10s} n={c['n']:4d} mfma={c['mfma']:3d} valu={c['valu']:4d} gl={c['gl']:3d} gs={c['gs']:3d} dsr={c['dsr']:3d} dsw={c['dsw']:3d} wait={c['wait'][:5]} br={c['br'][:4]}")
i = s.index('.amdhsa_kernel ' + m.group(1))
print(re.findall(r'\.amdhsa_next_free_vgpr \d+|\.amdhsa_accum_offset \d+|\.amdhsa_group_segment_fixed_size \d+|\.amdhsa_private_segment_fixed_size \d+', s[i:i+3000]))
