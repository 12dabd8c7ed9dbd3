#!/usr/bin/env python3
"""Round 4: what the lean sinh / cosh of variations.h buy a flame that uses the trigonometric / hyperbolic variations (cfg2 with its three
variations replaced by cosh, sech and csc): frame time through queue_frame, lean forms against the device library's
(FLAME_RTC_FLAGS=-DFL_LIBM_MATH compiles the per-genome kernel with the latter).   python tools/exp_hyp.py"""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from cuburn_amd import configs, profile, render
    gnm, prof = configs.cfg2()
    for xf, name in zip(gnm['xforms'].values(), ('cosh', 'sech', 'csc')):
        xf['variations'] = {name: {'weight': 0.6}, 'linear': {'weight': 0.4}}
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, host_seed=3)
    rdr = render.Renderer(gnm, gprof)
    prev = None
    for phase in (0, 1):
        n = 8 if phase == 0 else 60
        t0 = time.perf_counter()
        for k in range(n):
            cur = m.queue_frame(rdr, gnm, gprof, 0.3 + 0.001 * k)
            if prev is not None:
                prev[0].synchronize()
            prev = cur
        prev[0].synchronize()
        dt = time.perf_counter() - t0
    print('%-18s %.3f ms per frame' % (os.environ.get('FLAME_RTC_FLAGS', '') or 'lean', dt / n * 1e3))
else:
    for flags in ('', '-DFL_LIBM_MATH', '', '-DFL_LIBM_MATH'):
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, FLAME_RTC_FLAGS=flags), stderr=subprocess.DEVNULL)
