import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from cuburn_amd import configs, profile, render
gnm, prof = configs.cfg2()
gprof = profile.wrap(prof, gnm)
m = render.RenderManager(device=0, host_seed=1)
rdr = render.Renderer(gnm, gprof)
for _ in range(5):
    e, o = m.queue_frame(rdr, gnm, gprof, 0.5); e.synchronize()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    e, o = m.queue_frame(rdr, gnm, gprof, 0.5)
    t1 = time.perf_counter()
    e.synchronize()
    ts.append((t1 - t0) * 1e3)
print('queue_frame host time ms: min %.3f median %.3f max %.3f' % (min(ts), sorted(ts)[10], max(ts)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    e, o = m.queue_frame(rdr, gnm, gprof, 0.5); e.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
