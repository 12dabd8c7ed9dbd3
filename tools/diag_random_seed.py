#!/usr/bin/env python3
"""Why does a random genome's mean colour differ between the device and the CPU game?  Renders the
genome of tests/test_gpu_random_genomes.py (seed argv[1]) with several fuse lengths and sample counts:
a chain that mixes slowly needs a longer fuse than 64 rounds, on either side.
    python tools/diag_random_seed.py 50"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import O, prepare, frame_times
from cuburn_amd import profile, render, _lib
import test_gpu_random_genomes as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.load()
gnm, prof = T.random_genome(seed)
print('xforms:', {k: sorted(v['variations']) for k, v in gnm['xforms'].items()}, 'final' in gnm)
gprof = profile.wrap(prof, gnm)
tc = 0.37
def blocks(a, dim):
    H, W = dim.ah // 16 * 16, dim.astride // 16 * 16
    return a[:, 3].reshape(dim.ah, dim.astride)[:H, :W].reshape(H // 16, 16, W // 16, 16).sum((1, 3))

for nslots in (1024 if seed % 2 == 0 else 1536,):
    m = render.RenderManager(device=0, nslots=nslots, host_seed=42)
    rdr = render.Renderer(gnm, gprof)
    g = rdr._handle(m.fb); m._copy(rdr, gnm)
    dim = m.fb.calc_dim(gprof.width, gprof.height)
    ts, td = frame_times(gprof, tc)
    F = prepare(gnm, prof, tc, nslots=nslots)
    refh, _, _ = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** 26, 64)
    refh = refh.astype(np.float64); br = blocks(refh, dim)
    for fuse in (64, 256, 1024, 4096):
        for n in (2 ** 24,):
            _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
            run = C.c_uint64()
            _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(n), fuse, 1, C.byref(run)))
            front = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32).astype(np.float64)
            bg = blocks(front, dim)
            print('gpu nslots %d fuse %4d n 2^%d: in-frame %.4f colour %s  block L1 vs cpu %.4f' % (nslots, fuse, int(np.log2(n)), front[:, 3].sum() / run.value,
                  np.round(front[:, :3].sum(0) / front[:, 3].sum(), 4), np.abs(bg / bg.sum() - br / br.sum()).sum()))
    for nthreads, n in ((8, 2 ** 24), (64, 2 ** 26)):
        refh, _, _ = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, nthreads)
        refh = refh.astype(np.float64)
        print('cpu nslots %d threads %2d n 2^%d: in-frame %.4f colour %s' % (nslots, nthreads, int(np.log2(n)), refh[:, 3].sum() / n,
              np.round(refh[:, :3].sum(0) / refh[:, 3].sum(), 4)))
    m.fb.free()
