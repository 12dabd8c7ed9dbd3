#!/usr/bin/env python3
"""One-off soak: slot counts other than the shipped 1024 / 1536 (any multiple of 256 is accepted) and all
three workgroup sizes: binned accumulate == direct atomics == oracle on the transcendental-free flame
(td > 0: every slot has its own palette row and parameter block)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cuburn_amd import render
import test_gpu_parity as P

for nw, nslots in ((4, 1280), (4, 1792), (4, 2304), (4, 4096), (8, 1280), (8, 2048), (16, 1280), (16, 2048)):
    os.environ['FLAME_NW'] = str(nw)
    m = render.RenderManager(device=0, nslots=nslots, host_seed=11)
    gnm, prof = P.linear_flame()
    prof = dict(prof, width=1920, height=1080, frame_width=1.0)
    gnm['camera']['scale'] = 1.0
    ok = True
    try:
        res_a, ref_a, dev_a, dim, seeds = P.run_device_model(m, gnm, prof, nrounds=19, fuse=5, launches=1, mode=0, tc=0.3)
        res_b, ref_b, dev_b, dim, _ = P.run_device_model(m, gnm, prof, nrounds=19, fuse=5, launches=1, mode=1, seeds_in=seeds, tc=0.3)
        a, b = res_a[0], res_b[0]
        ok = (np.array_equal(a['atom_dev'], a['atom_ref']) and np.array_equal(b['atom_dev'], b['atom_ref'])
              and np.array_equal(b['front_dev'][:, 3], b['front_ref'][:, 3]) and np.array_equal(dev_a[0], ref_a[0]) and np.array_equal(dev_b[0], ref_b[0])
              and np.allclose(b['front_dev'][:, :3], b['front_ref'][:, :3], rtol=1e-5, atol=1e-4))
        extra = 'accepted %d' % int(a['ctr_dev'][0])
    except Exception as e:
        ok, extra = False, repr(e)[:200]
    print('%s nw %2d nslots %5d: %s' % ('ok  ' if ok else 'FAIL', nw, nslots, extra), flush=True)
    m.fb.free()
