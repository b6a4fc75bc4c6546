#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in a hipcc -S dump (device-only assembly).
usage: isa_census.py file.s <substring of the mangled kernel name>   -- prints per-block counts by instruction class and branches."""
import collections
import re
import sys


def cls(t):
    op = t.split()[0]
    if op.startswith('v_'):
        if any(k in t for k in ('row_shr', 'row_bcast', 'quad_perm', 'wave_shr', 'row_shl', 'row_ror', 'wave_ror')):
            return 'vdpp'
        if op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
            return 'vlane'
        if op.startswith('v_cmp'):
            return 'vcmp'
        if op.startswith('v_cndmask'):
            return 'vcnd'
        if op.startswith('v_mov') or op.startswith('v_accvgpr'):
            return 'vmov'
        if 'f64' in op:
            return 'vf64'
        return 'vother'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('scratch_') or op.startswith('buffer_'):
        return 'vmem'
    if op.startswith('ds_'):
        return 'lds'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    key = sys.argv[2]
    start = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l.split(':')[0]][0]
    fend = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
    blocks, cur = [], ('entry', [])
    for l in lines[start + 1:fend]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
        else:
            t = l.strip()
            if t and not t.startswith(';') and not t.startswith('.'):
                cur[1].append(t.split(';')[0].strip())
    blocks.append(cur)
    tot = collections.Counter()
    for name, ins in blocks:
        c = collections.Counter(cls(t) for t in ins)
        tot.update(c)
        br = [t.split()[-1] for t in ins if t.startswith('s_cbranch') or t.startswith('s_branch') or t.startswith('s_swappc') or t.startswith('s_setpc')]
        print(f"{name:12s} {len(ins):5d} {dict(sorted(c.items()))} -> {br}")
    print("TOTAL", sum(tot.values()), dict(sorted(tot.items())))


if __name__ == '__main__':
    main()
