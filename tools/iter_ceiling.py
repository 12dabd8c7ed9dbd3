"""Measure k_iter with and without the accumulate (ceiling of the walk itself)."""
import sys, os, ctypes as C, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from cuburn_amd import configs, profile, render, _lib
lib = _lib.load()
mgr = render.RenderManager(device=0, nslots=int(os.environ.get('NSLOTS', 1024)), host_seed=42)
for cfgname in sys.argv[1:] or ['cfg2']:
    gnm, prof = configs.CONFIGS[cfgname](samples=2 ** 28)
    prof = dict(prof, width=1920, height=1080)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof); g = rdr._handle(mgr.fb); mgr._copy(rdr, gnm)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, 1920, 1080, 0.5, 0.0))
    for mode in [int(m) for m in os.environ.get('MODES', '0,2,1').split(',')]:
        mgr.timings_reset()
        run = C.c_uint64()
        _lib.check(lib.fl_iterate(mgr.fb.ctx, g, 1920, 1080, float(2 ** 28), 64, mode, C.byref(run)))
        t = mgr.timings()
        print(cfgname, 'mode', mode, 'iter %.3f ms  %.1f Gsamples/s (incl. fuse rounds: %.1f Giter/s)' % (
            t['iter_ms'], run.value / t['iter_ms'] / 1e6, (run.value + 64 * mgr.fb.nslots * 256) / t['iter_ms'] / 1e6))
