"""Static instruction mix per barrier-delimited segment of one kernel (hipcc -save-temps .s file)."""
import sys, re, collections
path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index('\n' + name + ':'); j = s.index('.Lfunc_end', i)
segs = s[i:j].split('s_barrier')
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 'smem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    return 'other'
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for k, p in enumerate(segs):
    if k < first: continue
    c = collections.Counter()
    for l in p.split('\n'):
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.') or l.endswith(':'): continue
        c[cls(l.split()[0])] += 1
    print(k, sum(c.values()), dict(c))
