#!/usr/bin/env python3
"""(Superseded by tools/isa_vmflow.py, which follows the control flow; kept as the quick linear look.)
Listing check for kernels with hand-counted asynchronous loads (k_sauvola.hip, k_optimise_ws.hip): an instruction
that READS a register between the asm load that targets it and the asm `s_waitcnt vmcnt(N)` covering it copies or spills
data that has not landed.  Linear scan per basic-block order (ignores control flow: a hit is a place to look at, not proof).
usage: isa_inflight.py file.s <substring of the mangled kernel name>"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
inasm, pending, bad = False, [], 0


def regs_of(tok):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


for i in range(i0, i1):
    t = lines[i].strip()
    if t.startswith(';;#ASMSTART'):
        inasm = True
    elif t.startswith(';;#ASMEND'):
        inasm = False
    elif inasm and t.startswith('global_load'):
        pending.append((regs_of(t.split(',')[0]), i))
    elif t.startswith(('global_store', 'buffer_store')) and not t.startswith('global_store_lds'):
        # a store (asm or the compiler's) is a vm operation like a load: vmcnt retires them in issue order, so it takes a
        # place in the queue (k_sauvola.hip's counted stores put a fixed number of them between a load and its wait)
        rs = set()
        if not inasm:
            for s_ in t.split(None, 1)[1].split(','):
                rs |= regs_of(s_)
            for regs, li in pending:
                if rs & regs:
                    bad += 1
                    if bad <= 20:
                        print('line %d: %s   <- registers of the load at line %d' % (i - i0, t, li - i0))
        pending.append((set(), i))
    elif inasm and t.startswith('s_waitcnt') and 'vmcnt' in t:
        n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
        pending = pending[-n:] if n > 0 else []
    elif not inasm and t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        pending = []
    elif not inasm and t and t[0] not in ';.' and ' ' in t:
        op, args = t.split(None, 1)
        parts = args.split(',')
        srcs = parts if op.startswith(('global_store', 'ds_write', 'scratch_store', 'buffer_store')) else parts[1:]
        rs = set()
        for s_ in srcs:
            rs |= regs_of(s_)
        for regs, li in pending:
            if rs & regs:
                bad += 1
                if bad <= 20:
                    print('line %d: %s   <- registers of the load at line %d' % (i - i0, t, li - i0))
        # a register the compiler overwrites holds a new value from here on: later reads are not reads of the load's
        # target (the linear scan runs through loops that never follow each other at run time)
        if not op.startswith(('global_store', 'ds_write', 'scratch_store', 'buffer_store', 's_', 'v_cmp')):
            wr = regs_of(parts[0])
            if wr:
                pending = [(regs - wr, li) for regs, li in pending]
print('reads of registers with an asm load in flight: %d' % bad)
sys.exit(1 if bad else 0)
