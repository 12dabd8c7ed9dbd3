#!/usr/bin/env python3
"""k_accum_tiles issues the tile add's returning atomics and the record loop's loads from inline asm and waits for
them itself (all of them, or all but the four newest loads); the compiler does not know the results are in flight, so
nothing may touch a result register between its instruction and the wait.  The walk follows program text order and
starts afresh after every unconditional branch (blocks placed out of line are entered from elsewhere).  This reads the device assembly (hipcc --cuda-device-only -S binned.hip) and checks exactly that.
    python tools/check_asm_atomics.py /tmp/binned.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
bad = n = 0
kern = None
pending = {}
for ln in lines:
    m = re.match(r'^(_Z\w+):', ln)
    if m: kern, pending = m.group(1), {}
    t = ln.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    if t.startswith('s_branch') or t.startswith('s_endpgm') or t.startswith('s_setpc'):
        # what follows in the text is entered by jumps only (out-of-line rare paths): its predecessors' state is
        # not known from text order, so the walk starts afresh there
        pending = {}
        continue
    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        pending = {}
        continue
    regs = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', t): regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r'\bv(\d+)\b', t))
    m = re.match(r'global_atomic_add_x2 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off sc0', t)
    ml = re.match(r'global_load_dword v(\d+), v(\d+), s\[(\d+):(\d+)\]$', t)          # the record loads of the pipelined loop (asm: no offset field)
    mw = re.match(r's_waitcnt vmcnt\(([1-9])\)$', t)
    if mw:
        # the pipelined loop's partial wait: the newest loads stay in flight
        newest = [r for r, what in pending.items() if what.startswith('global_load_dword')][-int(mw.group(1)):]
        pending = {r: pending[r] for r in newest}
        continue
    if m or ml:
        d = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else {int(ml.group(1))}
        use = regs - d
        hit = use & set(pending)
        if hit or (d & set(pending)): bad += 1; print('%s: %s touches in-flight %s' % (kern, t, sorted(hit | (d & set(pending)))))
        for r in d: pending[r] = t
        n += 1
        continue
    hit = regs & set(pending)
    if hit:
        bad += 1
        print('%s: "%s" touches v%s, still in flight from "%s"' % (kern, t, sorted(hit), pending[min(hit)]))
print('%d asm-issued returning atomics / record loads checked, %d violations' % (n, bad))
sys.exit(1 if bad else 0)
