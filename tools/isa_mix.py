"""Instruction mix of one kernel in a hipcc -S dump: python3 tools/isa_mix.py dump.s kernel_substring [--blocks]"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_ZN3mij\S*' + re.escape(name) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm'))
cnt = collections.Counter()
blocks = []
cur = ["entry", collections.Counter()]
for l in s[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.p2align') or t.startswith('.'):
        if t.startswith('.LBB') and t.endswith(':'):
            pass
        else:
            continue
    if t.endswith(':') or re.match(r'^\.LBB\S+:', t):
        blocks.append(cur)
        cur = [t.split(':')[0], collections.Counter()]
        continue
    op = t.split()[0]
    cnt[op] += 1
    cur[1][op] += 1
blocks.append(cur)
tot = sum(cnt.values())
def klass(op):
    if op.startswith('v_pk_') and 'f32' in op: return 'v_pk_f32'
    if op.startswith('v_') and ('f32' in op) and not op.startswith('v_cvt'): return 'v_f32'
    if op.startswith('v_cvt'): return 'v_cvt'
    if op.startswith('v_'): return 'v_int/other'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'vmem'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_'): return 'salu'
    return 'other'
kc = collections.Counter()
for op, n in cnt.items():
    kc[klass(op)] += n
print("total", tot, dict(kc))
for k, v in cnt.most_common(45):
    print("%6d %s" % (v, k))
if '--blocks' in sys.argv:
    for b in blocks:
        n = sum(b[1].values())
        if n >= 40:
            kc = collections.Counter()
            for op, c in b[1].items():
                kc[klass(op)] += c
            print(b[0], n, dict(kc))
