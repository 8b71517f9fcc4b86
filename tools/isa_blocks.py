"""Basic blocks of one kernel in /tmp/kernels.s (tools/kernel_resources.py --isa): instruction mix and branch targets per block.
usage: isa_blocks.py <substring of the mangled kernel name, e.g. trace_kernelILb0ELb0ELb1E>"""
import re, sys
s = open('/tmp/kernels.s').read()
m = re.search(r'^(_ZN7mi355rt\w*' + re.escape(sys.argv[1]) + r'\w*):', s, re.M)
start = m.end(); end = s.index('s_endpgm', start)
blk = 'entry'; order = [blk]; cnt = {blk: dict(v=0, s=0, m=0, l=0, br=[])}
for ln in s[start:end].split('\n'):
    t = ln.strip()
    mm = re.match(r'^(\.LBB\d+_\d+):', t)
    if mm:
        blk = mm.group(1); order.append(blk); cnt[blk] = dict(v=0, s=0, m=0, l=0, br=[]); continue
    if not t or t[0] in ';.': continue
    op = t.split()[0]; c = cnt[blk]
    if op.startswith('v_'): c['v'] += 1
    elif op.startswith(('s_cbranch', 's_branch')): c['br'].append(t.split()[-1]); c['s'] += 1
    elif op.startswith('s_'): c['s'] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['m'] += 1
    elif op.startswith('ds_'): c['l'] += 1
for b in order:
    c = cnt[b]
    print("%-12s VALU %3d SALU %3d VMEM %2d LDS %2d -> %s" % (b, c['v'], c['s'], c['m'], c['l'], ' '.join(c['br'])))
