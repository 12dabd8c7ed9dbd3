import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_parity as T
from cuburn_amd import render
mgr = render.RenderManager(device=0, nslots=T.NSLOTS, host_seed=42)
gnm, prof = T.linear_flame()
prof = dict(prof, width=1920, height=1080)
gnm['camera']['scale'] = 1.0
ra, _, da, dim, seeds = T.run_device_model_gpu_only(mgr, gnm, prof, nrounds=13, fuse=5, mode=0)
rb, _, db, dim, _ = T.run_device_model_gpu_only(mgr, gnm, prof, nrounds=13, fuse=5, mode=1, seeds_in=seeds)
a, b = ra['atom'], rb['atom']
ca, cb = (a >> np.uint64(54)).astype(np.int64), (b >> np.uint64(54)).astype(np.int64)
print('ctr', ra['ctr'], rb['ctr'])
print('count sums', ca.sum(), cb.sum(), 'cells differing', (a != b).sum(), 'count diffs', (ca != cb).sum())
d = cb - ca
idx = np.nonzero(a != b)[0]
print('first diffs', [(int(i) % dim.astride, int(i) // dim.astride, int(ca[i]), int(cb[i])) for i in idx[:12]])
print('sum of positive diffs', d[d > 0].sum(), 'negative', d[d < 0].sum())
front = mgr.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
print('debug words', front.view(np.uint32).reshape(-1)[:8], [hex(int(x)) for x in front.view(np.uint32).reshape(-1)[:8]])
