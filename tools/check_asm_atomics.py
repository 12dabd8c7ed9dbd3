#!/usr/bin/env python3
"""k_accum_tiles issues the tile add's returning atomics and the record loop's loads from inline asm and waits for
them itself (all of them, or all but the newest loads); the compiler does not know the results are in flight, so
nothing may touch a result register between its instruction and the wait.  This reads the device assembly
(hipcc --cuda-device-only -S binned.hip, with the flags of the build) and checks exactly that.  The walk follows
program text order (everything in the text counts as executed, which is the feasible path of the pipelined loop from its
second iteration on; a FORWARD branch carries its state to its label, where the state with more in flight continues —
the main path jumps over blocks the compiler placed out of line) and starts afresh after every unconditional
branch (blocks placed out of line are entered from elsewhere); at every BACKWARD branch — a loop's
back-edge, conditional or not — the loop body is walked a second time with the state the first pass ended with, so that what stays in
flight ACROSS iterations (the pipelined loop's second record set) meets the top of the loop again.
(A walk of every path of the control-flow graph was tried: it reports the infeasible path that skips the
`v0 != 0` wait in a later iteration.)
    python tools/check_asm_atomics.py build/binned.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
ATOM = re.compile(r'global_atomic_add_x2 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off sc0')
LOAD = re.compile(r'global_load_dword v(\d+), v(\d+), s\[(\d+):(\d+)\]$')          # the record loads of the pipelined loop (asm: no offset field)
LOAD2 = re.compile(r'global_load_dwordx2 v\[(\d+):(\d+)\], v(\d+), s\[(\d+):(\d+)\]$')      # ... of the packed log: three records per 64-bit word
WAITN = re.compile(r's_waitcnt .*vmcnt\(([1-9]\d*)\)|s_waitcnt vmcnt\(([1-9]\d*)\)')
LABEL = re.compile(r'^(\.LBB\d+_\d+):')
FUNC = re.compile(r'^(_Z\w+):')


def regs_of(t):
    r = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', t):
        r.update(range(int(a), int(b) + 1))
    r.update(int(x) for x in re.findall(r'\bv(\d+)\b', t))
    return r


def parse(t):
    if t.startswith('; acc-guarded-block'):
        return ('marker', None, None, t, None)
    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        return ('wait0', None, None, t, None)
    mw = WAITN.match(t)
    if mw:
        return ('waitn', int(mw.group(1) or mw.group(2)), None, t, None)
    ma, ml, ml2 = ATOM.match(t), LOAD.match(t), LOAD2.match(t)
    if ma or ml or ml2:
        mr = ma or ml2
        d = tuple(range(int(mr.group(1)), int(mr.group(2)) + 1)) if mr else (int(ml.group(1)),)
        return ('atom' if ma else 'load', frozenset(regs_of(t)), d, t, None)
    m = LABEL.match(t)
    if m:
        return ('label', None, None, t, m.group(1))
    if t.startswith('s_branch') or t.startswith('s_endpgm') or t.startswith('s_setpc'):
        return ('reset', None, None, t, t.split()[1] if t.startswith('s_branch') else None)
    if t.startswith('s_cbranch'):
        return ('cjump', None, None, t, t.split()[1])
    return ('op', frozenset(regs_of(t)), None, t, None)


def check_function(name, body):
    """body: instruction / label lines of one function.  Returns (asm ops seen, violations)."""
    ins = [parse(t) for t in body]
    where = {i[4]: k for k, i in enumerate(ins) if i[0] == 'label'}
    nops, bad = set(), set()

    def walk(lo, hi, atoms, loads, second):
        """Text-order walk of ins[lo:hi]; returns the state at the end."""
        atoms, loads = set(atoms), list(loads)
        carried = {}            # label -> state a FORWARD branch brought along (the main path jumps over blocks placed out of line)
        fresh = False           # the state was dropped at an unconditional branch and nothing has happened since
        for k in range(lo, hi):
            kind, regs, dst, t, target = ins[k]
            if kind == 'label':
                c = carried.pop(target, None)
                if c is not None and (fresh or len(c[0]) + len(c[1]) > len(atoms) + len(loads)):
                    atoms, loads = set(c[0]), list(c[1])      # (the state with more in flight: the stricter one)
                fresh = False
                continue
            # A conditional branch around a block that the source marks as guarded ("if this is not the group's first step: wait for the
            # older set, add it") is the first step's path, taken only when nothing of that set is in flight: its state is not carried.
            guarded = False
            if kind == 'cjump' and target in where and where[target] > k:
                for j in range(k + 1, where[target]):
                    if ins[j][0] in ('load', 'atom', 'label', 'waitn', 'wait0'):
                        break
                    if ins[j][0] == 'marker':
                        guarded = True
                        break
            if kind in ('reset', 'cjump') and not guarded and target in where and where[target] > k:
                c = carried.get(target)
                if c is None or len(c[0]) + len(c[1]) < len(atoms) + len(loads):
                    carried[target] = (set(atoms), list(loads))
            fresh = fresh and kind not in ('atom', 'load')
            if kind == 'wait0':
                atoms, loads = set(), []
            elif kind == 'waitn':
                loads = loads[-regs:]                   # the newest n loads stay in flight
            elif kind == 'reset':
                # an unconditional branch BACK is a back-edge too (rotated loops end in one): once more over the body
                if not second and target in where and where[target] < k and (atoms or loads):
                    walk(where[target], k, atoms, loads, True)
                atoms, loads = set(), []
                fresh = True
            elif kind in ('atom', 'load'):
                inflight = {r for a in atoms for r in a} | {r for l in loads for r in l}
                hit = ((regs - set(dst)) & inflight) | (set(dst) & inflight)
                if hit:
                    bad.add('%s: %s touches in-flight v%s%s' % (name, t, sorted(hit), ' (second pass over a loop)' if second else ''))
                (atoms.add(dst) if kind == 'atom' else loads.append(dst))
                loads = loads[-16:]
                nops.add(t)
            elif kind == 'cjump':
                # a loop's back-edge: once more over the body with what is in flight now
                if not second and target in where and where[target] < k and (atoms or loads):
                    walk(where[target], k, atoms, loads, True)
            elif kind == 'marker':
                pass
            elif kind == 'op':
                inflight = {r for a in atoms for r in a} | {r for l in loads for r in l}
                hit = regs & inflight
                if hit:
                    bad.add('%s: "%s" touches v%s while its asm-issued load / atomic is in flight%s'
                            % (name, t, sorted(hit), ' (second pass over a loop)' if second else ''))
        return atoms, loads

    walk(0, len(ins), set(), [], False)
    return len(nops), sorted(bad)


funcs, cur, name = [], [], None
for ln in lines:
    m = FUNC.match(ln)
    if m:
        if name:
            funcs.append((name, cur))
        name, cur = m.group(1), []
        continue
    t = ln.strip()
    if name is not None and t.startswith('; acc-guarded-block'):
        cur.append(t)              # (a marker the kernel source leaves in front of a guarded block, see parse)
        continue
    if name is None or not t or t.startswith(';') or (t.startswith('.') and not LABEL.match(t)):
        continue
    cur.append(t)
if name:
    funcs.append((name, cur))
n = 0
violations = []
for name, body in funcs:
    k, bad = check_function(name, body)
    n += k
    violations += bad
for v in violations[:40]:
    print(v)
print('%d asm-issued returning atomics / record loads checked (loop bodies twice), %d violations' % (n, len(violations)))
sys.exit(1 if violations else 0)
