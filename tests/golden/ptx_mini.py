"""A PTX interpreter just large enough for the two inline-PTX blocks of cuburn/code/iter.py (the packed-cell add of the iterate
kernel, :332-411, and flush_atom, :429-541), for tests/golden/make_golden_ptx.py.  Vectorised over threads (numpy arrays per
register), forward branches through an active mask, warp votes over groups of 32 consecutive threads, memory operations executed
thread by thread in thread order (which makes the atomics atomic).  Not a general tool: unknown opcodes raise."""
import re

import numpy as np

U32, U64, F32 = np.uint32, np.uint64, np.float32
_DT = {'.pred': np.bool_, '.u32': U32, '.b32': U32, '.u64': U64, '.b64': U64, '.f32': F32}


class Memory(object):
    """Flat address space of named numpy arrays; every access goes through a byte view."""
    def __init__(self):
        self.regions = []

    def add(self, base, arr):
        assert arr.flags['C_CONTIGUOUS']
        self.regions.append((int(base), arr.view(np.uint8).reshape(-1)))
        return int(base)

    def _find(self, addr, nbytes):
        for base, b in self.regions:
            if base <= addr and addr + nbytes <= base + b.size:
                return b, addr - base
        raise IndexError('address 0x%x (+%d) is outside every buffer' % (addr, nbytes))

    def load(self, addr, dtype, count=1):
        b, o = self._find(int(addr), np.dtype(dtype).itemsize * count)
        return b[o:o + np.dtype(dtype).itemsize * count].view(dtype).copy()

    def store(self, addr, values):
        values = np.ascontiguousarray(values)
        b, o = self._find(int(addr), values.nbytes)
        b[o:o + values.nbytes] = values.view(np.uint8).reshape(-1)


def _imm(tok, dtype):
    if not re.fullmatch(r'[-+*/()<>.0-9a-fx\s]+', tok):
        raise ValueError('not an immediate: %r' % tok)
    v = eval(tok.replace('/', '/'), {'__builtins__': {}})           # (1.0/255.0), ((1<<18)-1), (256 << 23), 0.97 ...
    return dtype(v)


def _split_operands(s):
    out, depth, cur = [], 0, ''
    for ch in s:
        if ch in '{[(':
            depth += 1
        elif ch in '}])':
            depth -= 1
        if ch == ',' and depth == 0:
            out.append(cur.strip()); cur = ''
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Machine(object):
    def __init__(self, text, n, operands, special, mem, surface=None):
        self.n, self.ops, self.special, self.mem, self.surface = n, operands, special, mem, surface
        self.reg = {}
        self.prog = []
        body = re.sub(r'//[^\n]*', '', text[text.index('{') + 1:text.rindex('}')])
        for raw in body.split(';'):
            raw = raw.strip()
            while True:
                m = re.match(r'^(\w+):\s*', raw)                      # labels in front of a statement (or alone at the end)
                if not m:
                    break
                self.prog.append(m.group(1) + ':')
                raw = raw[m.end():]
            if raw:
                self.prog.append(' '.join(raw.split()))

    # ---- operand access
    def get(self, tok, dtype):
        if tok in self.reg:
            return self.reg[tok]
        if tok in self.special:
            return self.special[tok].astype(dtype)
        m = re.fullmatch(r'%(\d+)', tok)
        if m:
            return self.ops[int(m.group(1))]
        return np.full(self.n, _imm(tok, dtype), dtype)

    def put(self, name, value, mask):
        r = self.reg[name]
        r[mask] = value.astype(r.dtype)[mask] if isinstance(value, np.ndarray) else value

    def run(self):
        n = self.n
        active = np.ones(n, bool)
        waiting = {}
        for stmt in self.prog:
            if stmt.endswith(':'):
                active |= waiting.pop(stmt[:-1], np.zeros(n, bool))
                continue
            if stmt.startswith('.reg'):
                _, ty, names = stmt.split(None, 2)
                for nm in names.split(','):
                    self.reg[nm.strip()] = np.zeros(n, _DT[ty])
                continue
            mask = active
            m = re.match(r'@(!?)(\w+)\s+(.*)', stmt)
            if m:
                p = self.reg[m.group(2)]
                mask = active & (~p if m.group(1) else p)
                stmt = m.group(3)
            op, _, rest = stmt.partition(' ')
            a = _split_operands(rest)
            parts = op.split('.')
            base = parts[0]
            if base == 'bra':
                waiting[a[0]] = waiting.get(a[0], np.zeros(n, bool)) | mask
                active = active & ~mask
                continue
            if not mask.any() and base not in ('vote',):
                continue
            getattr(self, 'op_' + base)(parts, a, mask)
        assert not waiting, waiting

    # ---- integer / conversion
    def op_shl(self, parts, a, mask):
        x, s = self.get(a[1], U32).astype(U64), self.get(a[2], U32).astype(U64)
        self.put(a[0], np.where(s >= 32, U64(0), (x << np.minimum(s, U64(31))) & U64(0xffffffff)).astype(U32), mask)

    def op_shr(self, parts, a, mask):
        x, s = self.get(a[1], U32), self.get(a[2], U32)
        self.put(a[0], np.where(s >= 32, U32(0), x >> np.minimum(s, U32(31))).astype(U32), mask)

    def op_and(self, parts, a, mask):
        self.put(a[0], self.get(a[1], U32) & self.get(a[2], U32), mask)

    def op_add(self, parts, a, mask):
        assert parts[1] == 'u64'
        self.put(a[0], self.get(a[1], U64) + self.get(a[2], U64), mask)

    def op_cvt(self, parts, a, mask):
        kind = '.'.join(parts[1:])
        if kind == 'u64.u32':
            self.put(a[0], self.get(a[1], U32).astype(U64), mask)
        elif kind == 'rn.f32.u32':
            self.put(a[0], self.get(a[1], U32).astype(F32), mask)
        elif kind == 'rni.u32.f32':                                 # to nearest even, saturating, NaN -> 0
            f = self.get(a[1], F32).astype(np.float64)
            r = np.where(np.isnan(f), 0.0, np.clip(np.rint(f), 0.0, 4294967295.0))
            self.put(a[0], r.astype(U64).astype(U32), mask)
        else:
            raise NotImplementedError(kind)

    def op_mov(self, parts, a, mask):
        if parts[1] == 'b64' and a[1].startswith('{'):
            lo, hi = [t.strip() for t in a[1].strip('{}').split(',')]
            self.put(a[0], self.get(lo, U32).astype(U64) | (self.get(hi, U32).astype(U64) << U64(32)), mask)
        elif parts[1] == 'b64' and a[0].startswith('{'):
            lo, hi = [t.strip() for t in a[0].strip('{}').split(',')]
            v = self.get(a[1], U64)
            self.put(lo, (v & U64(0xffffffff)).astype(U32), mask)
            self.put(hi, (v >> U64(32)).astype(U32), mask)
        else:
            self.put(a[0], self.get(a[1], _DT['.' + parts[1]]), mask)

    def op_bfe(self, parts, a, mask):
        x, pos, ln = self.get(a[1], U32), int(_imm(a[2], int)), int(_imm(a[3], int))
        self.put(a[0], (x >> U32(pos)) & U32((1 << ln) - 1), mask)

    def op_bfi(self, parts, a, mask):                                # d = b with its bits [pos, pos+len) replaced by the low bits of a
        ins, b, pos, ln = self.get(a[1], U32), self.get(a[2], U32), int(_imm(a[3], int)), int(_imm(a[4], int))
        fm = U32(((1 << ln) - 1) << pos)
        self.put(a[0], (b & ~fm) | ((ins << U32(pos)) & fm), mask)

    # ---- predicates, votes
    def op_setp(self, parts, a, mask):
        comb = None
        if parts[1] == 'and':
            comb, parts = self.reg[a[3]], [parts[0]] + parts[2:]
        cmp_, ty = parts[1], _DT['.' + parts[2]]
        x, y = self.get(a[1], ty), self.get(a[2], ty)
        r = {'eq': x == y, 'gt': x > y, 'le': x <= y, 'lo': x < y, 'lt': x < y, 'ge': x >= y}[cmp_]
        if comb is not None:
            r = r & comb
        self.put(a[0], r, mask)

    def op_vote(self, parts, a, mask):
        assert parts[1:] == ['ballot', 'b32'] and self.n % 32 == 0
        p = (self.reg[a[1]] & mask).reshape(-1, 32)
        bal = (p.astype(U64) << np.arange(32, dtype=U64)).sum(1).astype(U32)
        self.put(a[0], np.repeat(bal, 32), mask)

    # ---- float
    def op_fma(self, parts, a, mask):                                # (the product of two float32 is exact in double)
        x, y, z = [self.get(t, F32).astype(np.float64) for t in a[1:4]]
        self.put(a[0], (x * y + z).astype(F32), mask)

    def op_mul(self, parts, a, mask):
        self.put(a[0], self.get(a[1], F32) * self.get(a[2], F32), mask)

    # ---- memory (thread by thread, in thread order)
    def _addr(self, tok):
        m = re.fullmatch(r'\[(\w+)(?:\+(\d+))?\]', tok)
        return self.reg[m.group(1)], int(m.group(2) or 0)

    def op_ld(self, parts, a, mask):
        ptr, off = self._addr(a[1])
        if 'v2' in parts or 'v4' in parts:
            names = [t.strip() for t in a[0].strip('{}').split(',')]
            ty = _DT['.' + parts[-1]]
            for t in np.nonzero(mask)[0]:
                v = self.mem.load(int(ptr[t]) + off, ty, len(names))
                for nm, x in zip(names, v):
                    self.reg[nm][t] = x
        else:
            ty = _DT['.' + parts[-1]]
            for t in np.nonzero(mask)[0]:
                self.reg[a[0]][t] = self.mem.load(int(ptr[t]) + off, ty)[0]

    def op_st(self, parts, a, mask):
        ptr, off = self._addr(a[0])
        ty = _DT['.' + parts[-1]]
        if a[1].startswith('{'):
            vals = [self.get(t.strip(), ty) for t in a[1].strip('{}').split(',')]
        else:
            vals = [self.get(a[1], ty)]
        for t in np.nonzero(mask)[0]:
            self.mem.store(int(ptr[t]) + off, np.array([v[t] for v in vals], ty))

    def op_suld(self, parts, a, mask):                               # suld.b.2d.v2.b32.clamp {lo, hi}, [surf, {xbytes, y}]
        names = [t.strip() for t in a[0].strip('{}').split(',')]
        m = re.fullmatch(r'\[(\w+),\s*\{(.+?),\s*(.+?)\}\]', a[1])
        xb, y = self.get(m.group(2), U32), self.get(m.group(3), U32)
        surf = self.surface                                          # [rows][cols] uint64
        for t in np.nonzero(mask)[0]:
            xi = min(max(int(np.int32(xb[t])), 0), surf.shape[1] * 8 - 8) // 8
            yi = min(max(int(np.int32(y[t])), 0), surf.shape[0] - 1)
            v = int(surf[yi, xi])
            self.reg[names[0]][t], self.reg[names[1]][t] = v & 0xffffffff, v >> 32

    def op_red(self, parts, a, mask):
        ptr, off = self._addr(a[0])
        ty = _DT['.' + parts[-1]]
        val = self.get(a[1], ty)
        for t in np.nonzero(mask)[0]:
            cur = self.mem.load(int(ptr[t]) + off, ty)
            self.mem.store(int(ptr[t]) + off, (cur + val[t:t + 1]).astype(ty))

    def op_atom(self, parts, a, mask):
        ptr, off = self._addr(a[1])
        ty = _DT['.' + parts[-1]]
        val = self.get(a[2], ty)
        for t in np.nonzero(mask)[0]:
            cur = self.mem.load(int(ptr[t]) + off, ty)
            new = (cur + val[t:t + 1]).astype(ty) if parts[2] == 'add' else val[t:t + 1].astype(ty)
            self.mem.store(int(ptr[t]) + off, new)
            self.reg[a[0]][t] = cur[0]
