"""Instruction mix of the big basic blocks of each kernel in an AMDGPU .s file: python scripts/asm_mix.py file.s [filter]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if flt and flt not in name: continue
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in body.split('\n'):
        s = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', s): cur = [s, []]; blocks.append(cur)
        elif s and not s.startswith((';', '.')): cur[1].append(s)
    tot = sum(len(b[1]) for b in blocks)
    print('==', name[:90], 'instructions', tot)
    for b in blocks:
        if len(b[1]) > 60:
            c = collections.Counter()
            for ins in b[1]:
                op = ins.split()[0]
                k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else
                     'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'scratch' if op.startswith('scratch_') else
                     'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'other')
                c[k] += 1
            print('   ', b[0], len(b[1]), dict(c))
