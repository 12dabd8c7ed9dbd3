import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuburn_amd import configs, profile, render
gnm, prof = configs.cfg2(samples=2 ** 24)
free0 = None
for size in [(1920, 1080), (640, 360), (3840, 2160), (1920, 1080)] * 5:
    gprof = profile.wrap(dict(prof, width=size[0], height=size[1], spp=2 ** 24 / float(size[0] * size[1])), gnm)
    m = render.RenderManager(device=0, host_seed=3)
    rdr = render.Renderer(gnm, gprof)
    for _ in range(3):
        evt, out = m.queue_frame(rdr, gnm, gprof, 0.5)
    evt.synchronize()
    assert np.array(out)[..., 3].max() > 0
    del rdr
    m.fb.free()
    del m
    gc.collect()
    free, total = torch.cuda.mem_get_info(0)
    if free0 is None: free0 = free
    print(size, 'free MB', free >> 20, 'delta vs first MB', (free0 - free) >> 20)
