#!/usr/bin/env python3
"""Where a DE workgroup's time goes (library built with -DDE_X_PHASES): renders cfg2 frames and prints, per direction,
the mean microseconds a workgroup spends in each phase.   FLAME_HIP_LIB=.../libflame_hip_ph.so python tools/de_phases.py"""
import ctypes as C, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import numpy as np
from cuburn_amd import configs, profile, render, _lib
os.environ['FLAME_LANES'] = '1'
lib = _lib.load()
gnm, prof = configs.CONFIGS['cfg2']()
gprof = profile.wrap(prof, gnm)
m = render.RenderManager(device=0, host_seed=42)
rdr = render.Renderer(gnm, gprof)
for _ in range(20):
    evt, _h = m.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
out = (C.c_ulonglong * 48)()
assert lib.fl_debug_de_phases(out, 1) == 0
n = 10
for _ in range(n):
    evt, _h = m.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
assert lib.fl_debug_de_phases(out, 1) == 0
a = np.array(list(out), dtype=np.float64).reshape(8, 6)
print('dir  workgroups/frame   load->LDS   blur1   tap terms   write B   taps+store   total  (us per workgroup, 100 MHz ticks)')
for p in range(8):
    wg = a[p, 5]
    us = a[p, :5] / wg / 100.0
    print('%d   %8d          %7.2f   %6.2f   %7.2f    %6.2f    %7.2f    %6.2f' % (p, wg / n, us[0], us[1], us[2], us[3], us[4], us.sum()))
