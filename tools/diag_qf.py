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
m = render.RenderManager(device=0, nslots=1024, host_seed=42)
rd = render.Renderer(gnm, gprof)
dim = m.fb.calc_dim(gprof.width, gprof.height)
nbins = dim.ah * dim.astride
print('spp', gprof.spp(tc), 'samples', gprof.spp(tc) * 480 * 270)
for k in range(6):
    e, h = m.queue_frame(rd, gnm, gprof, tc); e.synchronize()
    h = np.array(h)
    print('sync each: mean rgb %.3f alpha %.3f run %d' % (h[..., :3].mean(), h[..., 3].mean(), m.last_nsamples))
pend = None; outs = []
for k in range(6):
    nxt = m.queue_frame(rd, gnm, gprof, tc)
    if pend: pend[0].synchronize(); outs.append(np.array(pend[1]))
    pend = nxt
pend[0].synchronize(); outs.append(np.array(pend[1]))
print('pipelined means', [round(float(o[..., :3].mean()), 2) for o in outs])
