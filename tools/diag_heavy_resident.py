#!/usr/bin/env python3
"""A three-xform flame of heavy variations with post affines: its per-genome kernel takes 88 vector registers with the full hoist budget,
82 with budget 5, 77 with none — rtc_iter_kernel settles on the last (four-wave workgroups must stay at or below 80 for the 1536-slot
geometry).  Renders two frames through queue_frame and prints the timings (spec_launches > 0: the per-genome kernel ran)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from cuburn_amd import configs, profile, render
gnm, prof = configs.cfg2()
names = ['super_shape', 'flux', 'elliptic']
for xf, name in zip(gnm['xforms'].values(), names):
    xf['variations'] = {name: {'weight': 0.5}, 'bipolar': {'weight': 0.3}, 'julian': {'weight': 0.2}}
    xf['post_affine'] = configs._affine(10.0, 1.05, 0.02, -0.03)
gprof = profile.wrap(prof, gnm)
m = render.RenderManager(device=0, host_seed=3)
rdr = render.Renderer(gnm, gprof)
t0 = time.perf_counter()
evt, h = m.queue_frame(rdr, gnm, gprof, 0.3); evt.synchronize()
t1 = time.perf_counter()
evt, h = m.queue_frame(rdr, gnm, gprof, 0.31); evt.synchronize()
t2 = time.perf_counter()
a = np.array(h)
print('first frame %.2f s (with compiles), second %.4f s, alpha>0 fraction %.3f, stats %s' % (t1 - t0, t2 - t1, (a[..., 3] > 0).mean(), m.timings()))
