#!/usr/bin/env python3
"""Golden vectors for the packed-cell arithmetic — the add of a sample into the 64-bit packed histogram cell with its overflow drain,
and flush_atom — produced by the REFERENCE's own inline PTX (cuburn/code/iter.py:332-411 and :429-541), for
tests/test_cpu_golden.py::test_packed_cell_add_and_flush_match_reference_ptx.

Both pieces exist only as PTX text inside ``asm volatile`` of the generated iterate module.  The module is rendered by the imported
reference for cfg2 (``mkiterlib``), the two PTX strings are cut out of it and interpreted by tests/golden/ptx_mini.py (a ~250-line
interpreter of exactly the opcodes they use: shifts, bit-field extract / insert, conversions, predicates and forward branches,
warp ballots, loads / stores, ``suld`` on the palette surface, 64-bit and float atomics).  Nothing of the text is kept.

Scenario: 6000 samples, one at a time (the interpreter's thread order = the oracle's sample order), into a 64 x 48 accumulator:
48 cells, one of them taking a quarter of the samples (it drains several times), three cells under hot-pixel multipliers 2 / 8 / 32,
every sample on the checked path (cosel 0.99: the add's result is looked at, as this design does on every add) — then a second
batch on the unchecked path (cosel 0.5, ``red``) into cells that stay far below the threshold; then flush_atom over the whole
grid as cuburn/render.py:361-364 launches it (16 x 16 blocks), with the three hot cells flagged in the hot-pixel map.
Kept (ptx_cells.npz): the samples, the palette, the accumulator (packed and float) after the adds, and after the flush the float
accumulator and the new hot flag of every cell.
    python tests/golden/make_golden_ptx.py          (in the build container: needs /root/reference)
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import make_golden as MG          # noqa: E402
import ptx_mini as PM             # noqa: E402

U32, U64, F32 = np.uint32, np.uint64, np.float32


def main():
    tmp, dst = MG.prepare_reference()
    from cuburn.code import iter as ref_iter
    from cuburn import render
    from cuburn_amd import configs
    gnm, prof = configs.cfg2()
    packer, lib = ref_iter.mkiterlib(gnm)
    blocks = re.findall(r'asm volatile \(("(?:[^"\\]|\\.)*")\s*::(.*?)\);', lib.defs, re.S)
    assert len(blocks) == 2
    ptx_add, ptx_flush = [bytes(lit[1:-1], 'ascii').decode('unicode_escape') for lit, _ in blocks]
    assert 'atom.global.add.u64' in ptx_add and 'vote.ballot.b32' in ptx_flush

    d = render.Framebuffers.calc_dim(36, 20)
    S, AH = int(d.astride), int(d.ah)
    ncell = S * AH
    rs = np.random.RandomState(2026)
    # the packed palette in the format interp_palette_flat writes (cuburn/code/interp.py:424-431): hi = 1 << 22 | y << 4, lo = u << 18 | v
    y8, u8, v8 = [rs.randint(0, 256, (64, 256)).astype(U64) for _ in range(3)]
    palette = np.ascontiguousarray((((U64(1) << U64(22)) | (y8 << U64(4))) << U64(32)) | (u8 << U64(18)) | v8)
    cells = rs.choice(ncell, 48, replace=False).astype(U32)
    hot_cells = {int(cells[1]): 1, int(cells[2]): 2, int(cells[3]): 3}            # cell -> hot flag (multiplier 2, 8, 32)
    n1, n2 = 5400, 600
    pick = rs.choice(48, n1, p=np.r_[0.25, np.full(47, 0.75 / 47)])
    gi1 = cells[pick]
    quiet = rs.choice(np.setdiff1d(np.arange(ncell), cells), 40, replace=False).astype(U32)
    gi2 = quiet[rs.randint(0, 40, n2)]
    gi = np.r_[gi1, gi2].astype(U32)
    n = n1 + n2
    cc = rs.uniform(-0.02, 1.02, n).astype(F32)
    dither = (0.49 * rs.uniform(-1, 1, n)).astype(F32)
    row = rs.randint(0, 64, n).astype(U32)
    cosel = np.r_[np.full(n1, 0.99), np.full(n2, 0.5)].astype(F32)
    mult = np.array([float((1 << (hot_cells.get(int(g), 0) << 1)) >> 1) or 1.0 for g in gi], F32)

    atom = np.zeros(ncell, U64)
    out4 = np.zeros((ncell, 4), F32)
    mem = PM.Memory()
    A, B = mem.add(0x100000000, atom), mem.add(0x200000000, out4)
    for i in range(n):
        ops = [np.array([cc[i]], F32), np.array([dither[i]], F32), np.array([row[i]], U32), np.array([gi[i]], U32),
               np.array([A], U64), np.array([cosel[i]], F32), np.array([B], U64), np.array([mult[i]], F32)]
        PM.Machine(ptx_add, 1, ops, {}, mem, surface=palette).run()
    atom_after_add, out_after_add = atom.copy(), out4.copy()
    print('adds done: drained cells', int((out4[:, 3] > 0).sum()), 'max packed count', int((atom >> U64(54)).max()))

    # flush_atom, blocks of 16 x 16 over astride x aheight; threads in block order, then y, then x: 32 consecutive = a warp
    hot = np.zeros(ncell // 16, U32)
    for g, f in hot_cells.items():
        yi, xi = divmod(g, S)
        hot[((yi >> 4) * S) + (xi & ~15) + (yi & 15)] |= U32(f << ((xi & 15) << 1))
    H = mem.add(0x300000000, hot)
    bx, by, ty, tx = np.meshgrid(np.arange(S // 16), np.arange(AH // 16), np.arange(16), np.arange(16), indexing='ij')
    order = np.lexsort((tx.ravel(), ty.ravel(), bx.ravel(), by.ravel()))
    tx, ty = tx.ravel()[order], ty.ravel()[order]
    xi = (bx.ravel()[order] * 16 + tx).astype(U32)
    yi = (by.ravel()[order] * 16 + ty).astype(U32)
    gidx = (yi * U32(S) + xi).astype(U32)
    hoti = (((yi >> U32(4)) * U32(S)) + (xi & U32(0xfffffff0)) + (yi & U32(0xf))).astype(U32)
    nt = gidx.size
    ops = [gidx, hoti, np.full(nt, A, U64), np.full(nt, B, U64), np.full(nt, H, U64), xi, yi]
    special = {'%tid.x': tx.astype(U32), '%tid.y': ty.astype(U32), '%laneid': ((ty * 16 + tx) % 32).astype(U32)}
    PM.Machine(ptx_flush, nt, ops, special, mem).run()
    assert not atom.any()
    flags = np.zeros(ncell, np.uint8)                                  # per cell, whatever the map's layout
    for g in range(ncell):
        yy, xx = divmod(g, S)
        flags[g] = (int(hot[((yy >> 4) * S) + (xx & ~15) + (yy & 15)]) >> ((xx & 15) << 1)) & 3
    np.savez_compressed(os.path.join(HERE, 'ptx_cells.npz'), astride=np.int32(S), aheight=np.int32(AH), palette=palette,
                        gi=gi, cc=cc, dither=dither, row=row, cosel=cosel, mult=mult,
                        hot_cells=np.array(sorted(hot_cells.items()), np.int32),
                        atom_after_add=atom_after_add, out_after_add=out_after_add, out_after_flush=out4, flags_after_flush=flags)
    print('wrote ptx_cells.npz; cells with flags', np.bincount(flags, minlength=4).tolist())


if __name__ == '__main__':
    main()
