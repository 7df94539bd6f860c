"""Instruction-type trace of the longest basic block of one kernel in a --save-temps .s file:
M mfma, r ds_read, w ds_write, D LDS-DMA, G other VMEM, W s_waitcnt, B s_barrier, f v_pk_fma, v other
VALU, a accvgpr moves, s SALU, n s_nop.   usage: isa_trace.py file.s <substring of the kernel symbol>"""
import re
import sys


def typ(line):
    op = line.split()[0]
    for pre, t in (('v_mfma', 'M'), ('ds_read', 'r'), ('ds_', 'w'), ('global_load_lds', 'D'),
                   ('global_', 'G'), ('buffer_', 'G'), ('s_waitcnt', 'W'), ('s_barrier', 'B'),
                   ('v_pk_fma', 'f'), ('v_accvgpr', 'a'), ('v_', 'v'), ('s_nop', 'n'), ('s_', 's')):
        if op.startswith(pre):
            return t
    return '?'


def main():
    path, key = sys.argv[1], sys.argv[2]
    s = open(path).read().split('\n')
    starts = [i for i, l in enumerate(s) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') is False
              and ':' in l and not l.startswith('\t')]
    i0 = starts[0]
    end = [i for i in range(i0, len(s)) if '.amdhsa_kernel' in s[i]][0]
    blocks, cur, name = [], [], 'entry'
    for i in range(i0, end):
        l = s[i]
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((name, cur))
            cur, name = [], l.split(':')[0]
        elif l.startswith('\t') and not l.startswith('\t.') and not l.strip().startswith(';'):
            cur.append(l.strip())
    blocks.append((name, cur))
    n, b = max(blocks, key=lambda x: len(x[1]))
    t = ''.join(typ(l) for l in b)
    print(n, len(b), 'instructions;', {c: t.count(c) for c in sorted(set(t))})
    for k in range(0, len(t), 120):
        print(t[k:k + 120])
    for l in s[i0:end + 60]:
        if 'vgpr' in l and ('.set' in l or 'amdhsa_next_free' in l or 'spill' in l.lower()):
            print(l.strip())
    if len(sys.argv) > 3:
        open(sys.argv[3], 'w').write('\n'.join(b))


main()
