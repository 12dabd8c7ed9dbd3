import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import test_gpu_parity as T
from cuburn_amd import render
mgr = render.RenderManager(device=0, nslots=1024, host_seed=42)
gnm, prof = T.linear_flame()
prof = dict(prof, width=int(sys.argv[1]), height=int(sys.argv[2]))
gnm['camera']['scale'] = float(sys.argv[3])
nr = int(sys.argv[4])
ra, _, _, dim, seeds = T.run_device_model(mgr, gnm, prof, nrounds=nr, fuse=5, launches=1, mode=0)
rb, _, _, dim, _ = T.run_device_model(mgr, gnm, prof, nrounds=nr, fuse=5, launches=1, mode=1, seeds_in=seeds)
a, b = ra[0]['atom_dev'], rb[0]['atom_dev']
print('ctr a', ra[0]['ctr_dev'], 'ctr b', rb[0]['ctr_dev'], 'ref', ra[0]['ctr_ref'])
ca, cb = (a >> np.uint64(54)).astype(np.int64), (b >> np.uint64(54)).astype(np.int64)
print('sum counts a', ca.sum(), 'b', cb.sum(), 'max a', ca.max(), 'max b', cb.max())
diff = np.nonzero(a != b)[0]
print('differing cells', len(diff))
if len(diff):
    y, x = diff // dim.astride, diff % dim.astride
    print('x range', x.min(), x.max(), 'y range', y.min(), y.max())
    print('tiles', sorted(set(zip((y // 128).tolist(), (x // 128).tolist())))[:20])
    for k in diff[:10]:
        print(int(k), 'a', int(ca[k]), 'b', int(cb[k]), hex(int(a[k])), hex(int(b[k])))
    print('front density equal:', np.array_equal(ra[0]['front_dev'][:, 3], rb[0]['front_dev'][:, 3]),
          'a total', ra[0]['front_dev'][:, 3].sum(), 'b total', rb[0]['front_dev'][:, 3].sum())
