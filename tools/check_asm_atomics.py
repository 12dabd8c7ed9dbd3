#!/usr/bin/env python3
"""The tile add of k_accum_tiles issues its returning atomics from inline asm and waits for all of them once; the
compiler does not know the results are in flight, so nothing may touch a result register between its atomic and the
wait.  This reads the device assembly (hipcc --cuda-device-only -S binned.hip) and checks exactly that.
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
    if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
        pending = {}
        continue
    regs = set()
    for a, b in re.findall(r'v\[(\d+):(\d+)\]', t): regs.update(range(int(a), int(b) + 1))
    regs.update(int(r) for r in re.findall(r'\bv(\d+)\b', t))
    m = re.match(r'global_atomic_add_x2 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off sc0', t)
    if m:
        d = set(range(int(m.group(1)), int(m.group(2)) + 1))
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
print('%d returning global_atomic_add_x2 checked, %d violations' % (n, bad))
sys.exit(1 if bad else 0)
