#!/usr/bin/env python3
"""How much of a BASELINE frame is empty as the DE sees it: the fraction of accumulator pixels with no hit, and of
DE tiles (32 x 8 / 32 x 16 output pixels) whose whole staged region (tile + the reach of taps and blurs) is empty,
for the input of each of the eight directions (liveness spreads 15 pixels per direction).
usage: tools/diag_de_sparsity.py [cfg2]"""
import ctypes as C
import os
import sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cuburn_amd import configs, profile, render, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
gnm, prof = configs.CONFIGS[cfg]()
gprof = profile.wrap(prof, gnm)
mgr = render.RenderManager(device=0, host_seed=42)
rdr = render.Renderer(gnm, gprof)
lib = _lib.load()
dim = mgr.fb.set_dim(gprof.width, gprof.height, nsamples=gprof.spp(0.5) * gprof.width * gprof.height)
g = rdr._handle(mgr.fb)
mgr._copy(rdr, gnm)
_lib.check(lib.fl_interp(mgr.fb.ctx, g, dim.w, dim.h, 0.5, 0.0))
run = C.c_uint64()
ns = gprof.spp(0.5) * dim.w * dim.h
_lib.check(lib.fl_iterate(mgr.fb.ctx, g, dim.w, dim.h, float(ns), mgr.fuse, mgr.resolve_accum_mode(dim), C.byref(run)))
nb = dim.ah * dim.astride
w = mgr.fb.read('front', (nb, 4), np.float32)[:, 3].reshape(dim.ah, dim.astride)
live = w > 0
print('%s: %dx%d accumulator, %d samples: %.2f %% of pixels empty' % (cfg, dim.astride, dim.ah, run.value, 100 * (1 - live.mean())))
from scipy.ndimage import maximum_filter
for d in range(8):
    # a tile is skippable when nothing within its reach (<= 24 rows / 24 columns beyond the tile) is live
    for th, tw in ((32, 8), (32, 16)):
        grown = maximum_filter(live, size=(th + 48, tw + 48), mode='constant')
        t = grown[th // 2::th, tw // 2::tw]
        print('  direction %d input: %.2f %% empty pixels, %dx%d tiles with an empty reach: %.2f %%' % (d, 100 * (1 - live.mean()), th, tw, 100 * (1 - t.mean())))
    live = maximum_filter(live, size=31, mode='constant')        # upper bound of the spread of one pass
