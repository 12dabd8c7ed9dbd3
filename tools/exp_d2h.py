#!/usr/bin/env python3
"""What the frame's D2H copy costs the frame loop: cfg2 frames queued two ahead with and without the copy to the pinned host buffer,
under the runtime's default copy path and with HSA_ENABLE_SDMA=0 / 1.   python tools/exp_d2h.py [cfg2 cfg4 cfg5]"""
import os, sys, subprocess, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
if len(sys.argv) > 1 and sys.argv[1] == '--child':
    sys.path.insert(0, ROOT)
    import torch
    from cuburn_amd import configs, profile, render, distributed as D
    gnm, prof = configs.CONFIGS[sys.argv[3]]()
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, host_seed=42)
    rdr = render.Renderer(gnm, gprof)
    host = sys.argv[2] == '1'
    nfr = {'cfg2': 300, 'cfg3': 100, 'cfg4': 120, 'cfg5': 12}[sys.argv[3]]
    q = lambda slot: m.queue_frame(rdr, gnm, gprof, 0.5, host=host)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.5:
        D.run_frame_loop(q, max(4, nfr // 8), depth=2); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t = time.perf_counter(); D.run_frame_loop(q, nfr, depth=2); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t) / nfr)
    print('%s host copy %s  HSA_ENABLE_SDMA=%s  %.4f ms per frame' % (sys.argv[3], host, os.environ.get('HSA_ENABLE_SDMA', 'default'), best * 1e3), flush=True)
    sys.exit(0)
for cfg in (sys.argv[1:] or ['cfg2']):
    for sdma in (None, '0', '1'):
        for host in ('1', '0'):
            env = dict(os.environ)
            if sdma is not None:
                env['HSA_ENABLE_SDMA'] = sdma
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', host, cfg], env=env)
