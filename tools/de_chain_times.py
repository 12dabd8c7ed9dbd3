#!/usr/bin/env python3
"""Mean microseconds a workgroup of the persistent DE launch spends per tile in each phase (library built with
-DCH_X_TIMES: s_memrealtime stamps per workgroup, read through fl_debug_chain_times).  usage: FLAME_DE_CHAIN=1 FLAME_HIP_LIB=... tools/de_chain_times.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from cuburn_amd import configs, profile, render, _lib
gnm, prof = configs.cfg2()
gprof = profile.wrap(prof, gnm)
mgr = render.RenderManager(device=0, host_seed=42)
rdr = render.Renderer(gnm, gprof)
lib = C.CDLL(os.environ.get('FLAME_HIP_LIB') or os.path.join(os.path.dirname(_lib.__file__), '_lib', 'libflame_hip.so'))
out = (C.c_ulonglong * 5)()
for _ in range(3):
    evt, h = mgr.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
lib.fl_debug_chain_times(out, 1)
n = 8
for _ in range(n):
    evt, h = mgr.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
assert lib.fl_debug_chain_times(out, 1) == 0
tiles = out[4]
names = ['fetch item', 'wait for neighbours', 'tile', 'publish']
print('%d tiles per frame; per tile:' % (tiles // n), '  '.join('%s %.2f us' % (names[i], out[i] / 100.0 / tiles) for i in range(4)),
      ' total %.2f us' % (sum(out[i] for i in range(4)) / 100.0 / tiles))
