#!/usr/bin/env python3
"""One-off soak: parameter interpolation (fl_interp) of random animated genomes at random frame
times, including the ends of the animation and times outside [0, 1], device against the oracle's
float64 restatement (the comparison of tests/test_gpu_random_genomes.py)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from common import prepare, frame_times
from cuburn_amd import profile, render, _lib
import test_gpu_random_genomes as T

lib = _lib.load()
m = render.RenderManager(device=0, nslots=1024, host_seed=3)
worst = 0.0
nbad = 0
for seed in range(1, 60, 2):                     # odd seeds are the animated ones
    gnm, prof = T.random_genome(seed)
    rs = np.random.RandomState(seed)
    for tc in [0.0, 1.0, 0.5, float(rs.uniform(-0.2, 1.2)), float(rs.uniform(0, 1))]:
        for fw in (0.0, 1.0, 8.0):
            p2 = dict(prof, frame_width=fw)
            gprof = profile.wrap(p2, gnm)
            rdr = render.Renderer(gnm, gprof)
            g = rdr._handle(m.fb); m._copy(rdr, gnm)
            dim = m.fb.calc_dim(gprof.width, gprof.height)
            ts, td = frame_times(gprof, tc)
            _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
            dev = m.fb.read('params', (m.fb.nslots, rdr.packer.pstride), np.float32, g)
            ref = prepare(gnm, p2, tc, nslots=m.fb.nslots)['params']
            names = ['.'.join(n) for n in rdr.packer.packed]
            lastden = names.index('den.' + rdr.packer.xform_keys[-1])
            dev[:, lastden] = ref[:, lastden]
            structural = np.array([n.split('.')[-1].startswith('#') or n.startswith('pad') for n in names])
            err = np.abs(dev - ref)[:, ~structural] / (np.abs(ref[:, ~structural]) + 1.0)
            e = float(np.nanmax(err)) if np.isfinite(err).all() else float('inf')
            worst = max(worst, e)
            if e > 5e-5 or not np.array_equal(dev[:, structural].view(np.uint32), ref[:, structural].view(np.uint32)):
                nbad += 1
                i = np.unravel_index(np.nanargmax(err), err.shape)
                print('FAIL seed %d tc %.3f fw %.1f: %s err %.3g (dev %.6g ref %.6g)' % (seed, tc, fw, np.array(names)[~structural][i[1]], e,
                      dev[:, ~structural][i], ref[:, ~structural][i]), flush=True)
print('worst relative error %.3g, %d failures' % (worst, nbad))
