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
def it(n, mode):
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m._copy(rd, gnm)
    g = rd._handle(m.fb)
    _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, ts, td))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, float(n), m.fuse, mode, C.byref(run)))
    return m.fb.read('front', (nbins, 4), np.float32).astype(np.float64), run.value
for mode in (1, 0):
    a, ra = it(2 ** 24, mode); b, rb = it(2 ** 24, mode); w, rw = it(2 ** 25, mode)
    s = a + b
    print('mode', mode, 'runs', ra, rb, rw)
    print(' density sums  halves %.1f  whole %.1f  ratio %.5f' % (s[:, 3].sum(), w[:, 3].sum(), s[:, 3].sum() / w[:, 3].sum()))
    print(' Y sums ratio %.5f' % (s[:, 0].sum() / w[:, 0].sum()))
    for lo, hi in ((0, 50), (50, 500), (500, 5000), (5000, 1e9)):
        msk = (w[:, 3] >= lo) & (w[:, 3] < hi)
        print('  px with whole density in [%g,%g): n=%d halves %.1f whole %.1f ratio %.5f' % (lo, hi, msk.sum(), s[msk, 3].sum(), w[msk, 3].sum(), s[msk, 3].sum() / max(w[msk, 3].sum(), 1)))
    print(' max density', s[:, 3].max(), w[:, 3].max())
print('--- filter chain on accumulators')
def chain(acc):
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m.fb.write('front', acc.astype(np.float32))
    for filt in rd.filts:
        filt.apply(m.fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    rd.out.convert(m.fb, gprof, dim)
    h = rd.out.copy(m.fb, dim)
    _lib.check(lib.fl_ctx_sync(m.fb.ctx))
    return np.array(h).astype(np.float64)
a, _ = it(2 ** 24, 1); b, _ = it(2 ** 24, 1); w, _ = it(2 ** 25, 1); w2, _ = it(2 ** 25, 1)
imgs = [chain(x) for x in (a + b, w, w2, a + b)]
print('means', [round(i[..., :3].mean(), 3) for i in imgs])
print('alpha means', [round(i[..., 3].mean(), 3) for i in imgs])
print('s vs w', np.abs(imgs[0] - imgs[1])[..., :3].mean(), 'w vs w2', np.abs(imgs[1] - imgs[2])[..., :3].mean(), 's vs s', np.abs(imgs[0] - imgs[3])[..., :3].mean())
e, h = m.queue_frame(rd, gnm, gprof, tc); e.synchronize(); print('queue_frame mean', np.array(h)[..., :3].mean())
e, h = m.queue_frame(rd, gnm, gprof, tc); e.synchronize(); print('queue_frame mean', np.array(h)[..., :3].mean())
