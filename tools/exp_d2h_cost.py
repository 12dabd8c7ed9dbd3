#!/usr/bin/env python3
"""Round 6: what the frame's D2H copy (a blit KERNEL on this pool: __amd_rocclr_copyBuffer, ~150 us per 1080p frame) costs the two-lane
frame loop: cfg2 frames delivered to pinned host memory (the bench's loop) against frames left in a device tensor (fl_output dev_out)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from cuburn_amd import configs, profile, render, distributed as D
gnm, prof = configs.cfg2()
gprof = profile.wrap(prof, gnm)
mgr = render.RenderManager(device=0, host_seed=42)
rdr = render.Renderer(gnm, gprof)
slot = torch.empty((1080, 1920, 4), dtype=torch.uint8, device='cuda')
def run(n, host):
    q = (lambda s: mgr.queue_frame(rdr, gnm, gprof, 0.5)) if host else (lambda s: mgr.queue_frame(rdr, gnm, gprof, 0.5, dev_out=slot.data_ptr(), host=False))
    D.run_frame_loop(q, n, depth=2)
for host in (True, False):
    run(200, host)
for rep in range(3):
    for host in (True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(300, host); torch.cuda.synchronize()
        print('host copy' if host else 'device only', '%.4f ms per frame' % ((time.perf_counter() - t0) / 300 * 1e3))
