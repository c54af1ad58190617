#!/usr/bin/env python3
"""Path-aware listing check for kernels with hand-counted asynchronous loads (k_sauvola.hip).

An asm `global_load` targets registers the compiler believes ready at once; the program covers every use with an asm
`s_waitcnt vmcnt(N)`, N = the vector-memory operations issued after the load (gfx9 retires loads and stores of a wave in
issue order on vmcnt).  That count is a property of the PATH taken between the load and the wait, so this tool builds the
control-flow graph of one kernel from a `hipcc -S --cuda-device-only` listing (labels, s_branch / s_cbranch_*, fall-through)
and runs a forward data-flow analysis over it to a fixpoint:

    state  = for every VGPR that is the target of an asm load: the SMALLEST number of vector-memory operations issued
             since that load over all paths reaching this point (the path on which a wait is weakest)
    vm op  = global_ / buffer_ / flat_ / scratch_ loads, stores and atomics (asm or the compiler's): every entry ages by one
    wait   = s_waitcnt vmcnt(N) (asm or the compiler's): entries with age >= N have landed and leave the state
    merge  = union of the predecessors' entries, minimum age

and reports every instruction that MENTIONS a register whose load may still be in flight on some path:
  * a read copies / spills / computes with data that has not landed (the wait in front of it was too weak on that path);
  * a write is clobbered when the load lands afterwards (e.g. a dead slot reused before the tile's closing vmcnt(0));
  * a second asm load into a register whose first load was never waited for.
Feasibility of edges: hipcc writes an unconditional branch in wave-uniform control flow as `s_cbranch_execnz` (the
fall-through is dead code); the analysis therefore also tracks whether exec is KNOWN to be non-zero (true at entry; lost at
every exec write except restores from a mask saved while exec was known non-zero) and drops the edge such a branch cannot
take.  The other idiom is a loop exit through a flag: `s_mov_b64 sX, -1 ... s_and_b64 vcc, exec, sX; s_cbranch_vccz`; 64-bit
SGPR pairs holding the constants -1 / 0 and what they make of vcc are tracked for it.  Everything else is conservative: an
edge is kept unless it is provably dead.
tools/isa_inflight.py is the older linear scan (no control flow); tests/test_isa_checks.py runs this one over every
Sauvola instantiation of the library.

usage: isa_vmflow.py file.s <substring of the mangled kernel name | --all [substring]>"""
import re
import sys

VM_PREFIX = ('global_load', 'global_store', 'global_atomic', 'buffer_load', 'buffer_store', 'buffer_atomic', 'flat_load',
             'flat_store', 'flat_atomic', 'scratch_load', 'scratch_store')
AGE_CAP = 63        # vmcnt is a 6-bit counter on gfx9: an older entry is covered by any wait


def regs_of(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def kernel_names(lines, sub=''):
    return [m.group(1) for l in lines for m in [re.match(r'\s*\.amdhsa_kernel (\S+)', l)] if m and sub in m.group(1)]


def parse(lines, key):
    """-> list of blocks {label, insts: [(lineno, text, in_asm)], succ: [labels], fall: bool}"""
    i0 = next(i for i, l in enumerate(lines) if l.startswith('_Z') and ':' in l and key in l.split(':')[0])
    i1 = next(i for i in range(i0, len(lines)) if lines[i].strip().startswith('.Lfunc_end'))
    blocks, cur, inasm = [], {'label': 'entry', 'insts': [], 'succ': [], 'fall': True}, False
    for i in range(i0 + 1, i1):
        t = lines[i].strip()
        if t.startswith(';;#ASMSTART') or t.startswith(';#ASMSTART'):
            inasm = True; continue
        if t.startswith(';;#ASMEND') or t.startswith(';#ASMEND'):
            inasm = False; continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            blocks.append(cur)
            cur = {'label': m.group(1), 'insts': [], 'succ': [], 'fall': True}
            continue
        if not t or t[0] in ';.':
            continue
        t = t.split(';')[0].strip()
        if not t:
            continue
        if cur['fall'] is False:      # code after an unconditional branch without a label: unreachable filler
            blocks.append(cur)
            cur = {'label': '_dead%d' % i, 'insts': [], 'succ': [], 'fall': True}
        cur['insts'].append((i - i0, t, inasm))
        op = t.split()[0]
        if op == 's_branch':
            cur['succ'].append(t.split()[1]); cur['fall'] = False
        elif op.startswith('s_cbranch'):
            cur['succ'].append(t.split()[1])
            cur['cond'] = op[len('s_cbranch_'):]
            # a conditional branch ends the block: what follows is a new (unlabelled) block reached by fall-through
            blocks.append(cur)
            cur = {'label': '_ft%d' % i, 'insts': [], 'succ': [], 'fall': True}
        elif op in ('s_endpgm', 's_setpc_b64', 's_swappc_b64'):
            if op != 's_endpgm':
                raise SystemExit('indirect control flow (%s) in %s: not modelled' % (op, key))
            cur['fall'] = False
    blocks.append(cur)
    return blocks


def sregs_of(tok):
    """SGPR numbers named in an operand string (s7, s[4:5]; vcc / exec are not tracked as masks)"""
    out = set()
    for m in re.finditer(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b', tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


class State:
    """pend: VGPR -> (age, line of its asm load); exec_nz: exec is known to be non-zero; nzmask: SGPRs (numbers of the
    low halves of pairs) that hold a copy of exec taken while it was known non-zero"""
    __slots__ = ('pend', 'exec_nz', 'nzmask', 'consts', 'vcc')

    def __init__(self, pend=None, exec_nz=True, nzmask=frozenset(), consts=None, vcc=None):
        self.pend, self.exec_nz, self.nzmask = dict(pend or {}), exec_nz, frozenset(nzmask)
        self.consts = dict(consts or {})       # low SGPR of a pair -> 'ones' | 'zero'
        self.vcc = vcc                         # 'nz' | 'z' | None (unknown)

    def key(self):
        return (tuple(sorted((r, a) for r, (a, _l) in self.pend.items())), self.exec_nz, self.nzmask,
                tuple(sorted(self.consts.items())), self.vcc)


def merge(a, b):
    """union of the loads in flight at their minimum age; what is KNOWN about exec must hold on both paths"""
    pend = dict(a.pend)
    for r, v in b.pend.items():
        if r not in pend or v[0] < pend[r][0]:
            pend[r] = v
    consts = {k: v for k, v in a.consts.items() if b.consts.get(k) == v}
    return State(pend, a.exec_nz and b.exec_nz, a.nzmask & b.nzmask, consts, a.vcc if a.vcc == b.vcc else None)


def transfer(state, insts, report=None):
    st, exec_nz, nz = dict(state.pend), state.exec_nz, set(state.nzmask)
    consts, vcc = dict(state.consts), state.vcc
    for ln, t, inasm in insts:
        op = t.split()[0]
        args = t[len(op):]
        ops = [a.strip() for a in args.split(',')]
        # ---- constants in SGPR pairs and what they make of vcc (the compiler's loop-exit flags) ----
        if not op.startswith(('s_cbranch', 's_branch', 's_waitcnt', 's_nop')):
            if op in ('s_and_b64', 's_andn2_b64') and ops[0] == 'vcc' and 'exec' in ops[1:]:
                other = [o for o in ops[1:] if o != 'exec']
                c = consts.get(min(sregs_of(other[0]))) if other and sregs_of(other[0]) else None
                if op == 's_andn2_b64' and ops[1] != 'exec':
                    c = None                                            # sX & ~exec: not the idiom
                if c is None:
                    vcc = None
                elif (c == 'ones') == (op == 's_and_b64'):
                    vcc = 'nz' if exec_nz else None                     # exec & -1, exec & ~0
                else:
                    vcc = 'z'                                           # exec & 0, exec & ~(-1)
            elif re.search(r'\bvcc(_lo|_hi)?\b', args) and (op.startswith('s_') and ops[0].startswith('vcc') or op.startswith('v_')):
                vcc = None                                              # any other (possible) write of vcc
            if op == 's_mov_b64' and ops[0].startswith('s[') and ops[1] in ('-1', '0'):
                consts[min(sregs_of(ops[0]))] = 'ones' if ops[1] == '-1' else 'zero'
            else:
                written = set()
                if op.startswith('s_') and not op.startswith(('s_cmp', 's_bitcmp', 's_barrier')):
                    written = sregs_of(ops[0]) if ops else set()
                elif op.startswith('v_'):
                    written = sregs_of(','.join(ops[:2]))
                for lo in [k for k in consts if k in written or k + 1 in written]:
                    del consts[lo]
        # ---- what is known about exec ----
        if op.endswith('_saveexec_b64'):                       # sD = exec; exec = f(sS, exec)
            d = sregs_of(ops[0])
            lo = min(d) if d else None
            src_nz = bool(sregs_of(ops[1])) and min(sregs_of(ops[1])) in nz
            nz -= d
            if exec_nz and lo is not None:
                nz.add(lo)
            exec_nz = (exec_nz or src_nz) if op.startswith('s_or_saveexec') else False
        elif op.startswith('s_') and ops and ops[0] == 'exec':
            if op == 's_mov_b64':
                exec_nz = bool(sregs_of(ops[1])) and min(sregs_of(ops[1])) in nz
            elif op == 's_or_b64':
                others = [o for o in ops[1:] if o != 'exec']
                exec_nz = exec_nz and 'exec' in ops[1:] or any(sregs_of(o) and min(sregs_of(o)) in nz for o in others)
            else:
                exec_nz = False
        elif op.startswith('v_cmpx'):
            exec_nz = False
        else:
            # an SGPR that is (possibly) written no longer holds the saved mask: first operand of a scalar instruction,
            # first two of a vector instruction (v_cmp / carry-out / v_readfirstlane / v_mad_u64 destinations)
            if op.startswith('s_') and not op.startswith(('s_cbranch', 's_branch', 's_waitcnt', 's_nop', 's_barrier', 's_cmp', 's_bitcmp')):
                nz -= sregs_of(ops[0]) if ops else set()
            elif op.startswith('v_'):
                nz -= sregs_of(','.join(ops[:2]))
        # ---- loads in flight ----
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                st = {r: v for r, v in st.items() if v[0] < n}
            continue
        is_vm = op.startswith(VM_PREFIX)
        if inasm and op.startswith('global_load'):
            dst = regs_of(ops[0])
            used = regs_of(','.join(ops[1:]))
            if report is not None:
                for r in sorted((dst | used) & set(st)):
                    report(ln, t, r, st[r], 'asm load into / from a register whose earlier load was never waited for')
            st = {r: (min(a + 1, AGE_CAP), l) for r, (a, l) in st.items() if r not in dst}
            for r in dst:
                st[r] = (0, ln)
            continue
        mentioned = regs_of(args)
        if mentioned and report is not None:
            for r in sorted(mentioned & set(st)):
                report(ln, t, r, st[r], 'register mentioned while its asm load may be in flight')
        if is_vm:
            st = {r: (min(a + 1, AGE_CAP), l) for r, (a, l) in st.items()}
    return State(st, exec_nz, nz, consts, vcc)


def analyse(lines, key):
    blocks = parse(lines, key)
    index = {b['label']: i for i, b in enumerate(blocks)}
    for b in blocks:
        missing = [l for l in b['succ'] if l not in index]
        if missing:
            raise SystemExit('branch target %s outside the kernel %s' % (missing, key))

    def edges(i, out):
        b = blocks[i]
        taken = [index[l] for l in b['succ']]
        fall = [i + 1] if b['fall'] and i + 1 < len(blocks) else []
        cond = b.get('cond')
        if out.exec_nz and cond == 'execnz':
            return taken                  # hipcc's unconditional branch of uniform control flow
        if out.exec_nz and cond == 'execz':
            return fall
        if out.vcc is not None and cond in ('vccz', 'vccnz'):
            return taken if (out.vcc == 'z') == (cond == 'vccz') else fall
        return taken + fall

    instate = [None] * len(blocks)
    instate[0] = State()
    work = [0]
    rounds = 0
    while work:
        i = work.pop()
        rounds += 1
        if rounds > 400000:
            raise SystemExit('no fixpoint in %s' % key)
        out = transfer(instate[i], blocks[i]['insts'])
        for j in edges(i, out):
            new = out if instate[j] is None else merge(instate[j], out)
            if instate[j] is None or new.key() != instate[j].key():
                instate[j] = new
                work.append(j)
    hits = []

    def report(ln, t, r, v, why):
        hits.append('line %d: %s   <- v%d: asm load at line %d, only %d vm operations since on some path (%s)'
                    % (ln, t, r, v[1], v[0], why))
    nloads = nwaits = 0
    for i, b in enumerate(blocks):
        if instate[i] is None:
            continue
        left = transfer(instate[i], b['insts'], report)
        for _ln, t, inasm in b['insts']:
            if inasm and t.startswith('global_load'): nloads += 1
            if inasm and t.startswith('s_waitcnt') and 'vmcnt' in t: nwaits += 1
        if b['insts'] and b['insts'][-1][1].startswith('s_endpgm'):      # what is in flight where the program ends
            for r, v in sorted(left.pend.items()):
                hits.append('s_endpgm at line %d with the asm load of line %d into v%d still in flight' % (b['insts'][-1][0], v[1], r))
    return hits, sum(1 for x in instate if x is not None), nloads, nwaits


def main():
    lines = open(sys.argv[1]).read().split('\n')
    if sys.argv[2] == '--all':
        keys = kernel_names(lines, sys.argv[3] if len(sys.argv) > 3 else '')
    else:
        keys = [sys.argv[2]]
    bad = 0
    for key in keys:
        hits, nb, nl, nw = analyse(lines, key)
        print('%s: %d reachable blocks, %d asm loads, %d asm vmcnt waits, %d findings' % (key, nb, nl, nw, len(hits)))
        for h in hits[:20]:
            print('   ', h)
        bad += len(hits)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
