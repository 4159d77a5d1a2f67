"""Static instruction mix per phase of a kernel's -save-temps assembly (development aid, no GPU).

Build the team kernel with -DGE2E_MARKS -save-temps (GE2E_PROF(i) becomes an assembly comment "; PHASEMARK i", no
code), cut the kernel out of the .s file and run:   python tools/isa_phase_counts.py kernel.s
Segments are named after the marker they FOLLOW; the scheduler moves code across a marker, so boundaries are
approximate, totals are exact."""
import collections
import re
import sys

KEYS = ['v_f32', 'v_int', 'v_mov', 'v_cvt/mix', 'v_dpp', 'v_trans', 'v_lane', 'v_cnd', 'v_cmp', 'mfma', 'lds', 'vmem', 'salu',
        's_wait', 'barrier']


def cls(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('v_'):
        if 'dpp' in op:
            return 'v_dpp'
        if op.startswith('v_cvt') or 'mix' in op:
            return 'v_cvt/mix'
        if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_permlane')):
            return 'v_trans'
        if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')):
            return 'v_lane'
        if op.startswith('v_cndmask'):
            return 'v_cnd'
        if op.startswith(('v_mov', 'v_accvgpr')):
            return 'v_mov'
        if op.startswith('v_cmp'):
            return 'v_cmp'
        if op.endswith(('_f32_e32', '_f32_e64', '_f32')):
            return 'v_f32'
        return 'v_int'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 's_wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    seg = collections.OrderedDict()
    cur = 'pre'
    seg[cur] = collections.Counter()
    for line in open(sys.argv[1]):
        line = line.strip()
        m = re.match(r'; PHASEMARK (\d+)', line)
        if m:
            cur = 'after' + m.group(1)
            seg.setdefault(cur, collections.Counter())
            continue
        if not line or line.startswith((';', '.')) or line.endswith(':'):
            continue
        line = line.split(';')[0].strip()
        if line:
            seg[cur][cls(line.split()[0])] += 1
    print('%-8s' % 'seg', *['%9s' % k for k in KEYS], '  VALU')
    tot = collections.Counter()
    for k, s in seg.items():
        print('%-8s' % k, *['%9d' % s[x] for x in KEYS], '  %d' % sum(s[x] for x in KEYS[:9]))
        tot.update(s)
    print('%-8s' % 'total', *['%9d' % tot[x] for x in KEYS], '  %d' % sum(tot[x] for x in KEYS[:9]))


if __name__ == '__main__':
    main()
