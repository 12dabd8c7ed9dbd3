import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from cuburn_amd import configs, profile, render, _lib
from common import frame_times
gnm, prof = configs.cfg2(samples=2 ** 25)
prof = dict(prof, width=480, height=270)
gprof = profile.wrap(prof, gnm)
lib = _lib.load()
tc = 0.5
ts, td = frame_times(gprof, tc)
m = render.RenderManager(device=0, nslots=1024, host_seed=42)
rd = render.Renderer(gnm, gprof)
dim = m.fb.calc_dim(gprof.width, gprof.height)
nbins = dim.ah * dim.astride
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def it(n):
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m._copy(rd, gnm)
    g = rd._handle(m.fb)
    _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(n), m.fuse, mode, C.byref(run)))
    return m.fb.read('front', (nbins, 4), np.float32).astype(np.float64), run.value
def filt():
    outs = []
    for f in rd.filts:
        f.apply(m.fb, gprof, getattr(gprof.filters, f.name), dim, tc)
        outs.append((f.name, m.fb.read('front', (nbins, 4), np.float32).astype(np.float64)))
    rd.out.convert(m.fb, gprof, dim)
    h = rd.out.copy(m.fb, dim)
    _lib.check(lib.fl_ctx_sync(m.fb.ctx))
    return np.array(h).astype(np.float64), outs
for k in range(6):
    a, r = it(2 ** 21)
    h, outs = filt()
    print('acc: dens %.0f Y %.0f U %.0f V %.0f maxd %.0f | out rgb %.2f a %.2f |' % (a[:, 3].sum(), a[:, 0].sum(), a[:, 1].sum(), a[:, 2].sum(), a[:, 3].max(), h[..., :3].mean(), h[..., 3].mean()),
          ' '.join('%s %.4g/%.4g' % (n, o[:, 0].sum(), o[:, 3].sum()) for n, o in outs))
