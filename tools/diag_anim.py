"""Diagnostic: animated linear flame at production slots, binned vs oracle: where do cells differ?"""
import sys, os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
from cuburn_amd import render
import test_gpu_parity as T

m = render.RenderManager(device=0, host_seed=42)
gnm, prof = T.animated_linear_flame()
for mode in (0, 1):
    res, ref_state, dev_state, dim, _ = T.run_device_model(m, gnm, prof, nrounds=19, fuse=5, launches=1, mode=mode)
    r = res[0]
    a, b = r['atom_dev'], r['atom_ref']
    ca, cb = (a >> np.uint64(54)), (b >> np.uint64(54))
    print('mode', mode, 'ctr', r['ctr_dev'], r['ctr_ref'], 'cells differing', int((a != b).sum()), 'count differing', int((ca != cb).sum()),
          'sum counts', int(ca.sum()), int(cb.sum()))
    bad = np.where(a != b)[0][:5]
    for i in bad:
        print('  gi', i, 'y', i // dim.astride, 'x', i % dim.astride, hex(int(a[i])), hex(int(b[i])))
