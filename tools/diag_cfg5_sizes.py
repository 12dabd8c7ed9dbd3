"""cfg5's genome at several image sizes: iterate / accumulate ms per 2^28 samples (how much of the 8K cost is the binning?)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cuburn_amd import configs, profile, render
gnm, prof = configs.cfg5()
for (w, h, nw) in ((1920, 1080, None), (1920, 1080, '8'), (3840, 2160, None), (7680, 4320, None)):
    if nw: os.environ['FLAME_NW'] = nw
    else: os.environ.pop('FLAME_NW', None)
    os.environ['FLAME_LANES'] = '1'
    p = dict(prof, width=w, height=h, spp=2 ** 30 / float(w * h))
    gprof = profile.wrap(p, gnm)
    m = render.RenderManager(device=0, nslots=1024 if nw else None, host_seed=3)
    rdr = render.Renderer(gnm, gprof)
    for k in range(3):
        if k == 1: m.timings_reset()
        e, _ = m.queue_frame(rdr, gnm, gprof, 0.5); e.synchronize()
    t = m.timings()
    n = m.last_nsamples * 2 / 2 ** 28
    print('%dx%d nw=%d slots=%d: iter %.3f  accum %.3f  flush %.3f  de %.3f ms per 2^28 samples' % (
        w, h, m.fb.nw, m.fb.nslots, t['iter_ms'] / n, t['accum_ms'] / n, t['flush_only_ms'] / n, t['de_ms'] / 2))
    m.fb.free()
