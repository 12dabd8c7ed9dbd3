#!/usr/bin/env python3
"""Per-workgroup run times of k_accum_tiles (library built with -DACC_X_TIMES): which tiles' workgroups run longest, when
they start, and how the kernel's span compares with the sum of the work.   FLAME_HIP_LIB=..._at.so python tools/acc_times.py"""
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
for _ in range(300):                                   # long enough for the clocks to settle
    evt, _h = m.queue_frame(rdr, gnm, gprof, 0.5); evt.synchronize()
nparts, ntiles = 16, 288
n = ntiles * nparts
out = (C.c_ulonglong * (4 * n))()
assert lib.fl_debug_acc_times(out, n) == 0
a = np.array(list(out), dtype=np.float64).reshape(n, 4) / 100.0          # us
a = a[a[:, 0] > a[:, 0].max() - 2000.0]                                    # (entries of workgroups without work in the last launch are stale)
n = len(a)
t0 = a[:, 0].min()
start = a[:, 0] - t0
zero, recs, add = a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2]
dur = a[:, 3] - a[:, 0]
span = (a[:, 3] - t0).max()
print('workgroups %d: span %.1f us; sum of run times / 512 slots = %.1f us (%.0f workgroups in flight on average)' % (n, span, dur.sum() / 512, dur.sum() / span))
print('per workgroup (mean / median / p90 / max us): zero tile + stage palette %.1f / %.1f / %.1f / %.1f; records %.1f / %.1f / %.1f / %.1f; add tile to the global cells %.1f / %.1f / %.1f / %.1f'
      % (zero.mean(), np.median(zero), np.percentile(zero, 90), zero.max(), recs.mean(), np.median(recs), np.percentile(recs, 90), recs.max(),
         add.mean(), np.median(add), np.percentile(add, 90), add.max()))
uniform = n == ntiles * nparts
per_tile = dur.reshape(ntiles, nparts).mean(1) if uniform else dur[:ntiles]
order = np.argsort(-per_tile)
if uniform: print('slowest tiles (tile: mean run time of its 16 workgroups, first start):', ', '.join('%d: %.0f us @%.0f' % (t, per_tile[t], start.reshape(ntiles, nparts)[t].min()) for t in order[:8]))
late = np.argsort(-(a[:, 3] - t0))[:6]
print('last to finish (workgroup: tile, start, run):', ', '.join('%d: tile %d @%.0f +%.0f' % (w, w // nparts, start[w], dur[w]) for w in late))
busy = np.zeros(int(span) + 2)
for s0, d0 in zip(start, dur):
    busy[int(s0):int(s0 + d0) + 1] += 1
print('workgroups in flight over time (every 40 us):', ' '.join('%d' % busy[i] for i in range(0, min(int(span), 2000), 40)))
st = np.sort(start)
print('start times of workgroups 0, 100, 200, 255, 256, 300, 400, 511, 512, 600 (sorted by start): ' + ' '.join('%.1f' % st[i] for i in (0, 100, 200, 255, 256, 300, 400, 511, 512, 600)))
ends = np.sort(a[:, 3] - t0)
print('first ends: ' + ' '.join('%.1f' % e for e in ends[:6]))
fast = np.argsort(dur)[:8]
print('fastest workgroups (tile: zero / records / add us):', ', '.join('%d: %.1f / %.1f / %.1f' % (w // nparts, zero[w], recs[w], add[w]) for w in fast))
q = np.percentile(recs, [1, 5, 10, 25, 50, 75])
print('records phase percentiles 1/5/10/25/50/75 %%: ' + ' '.join('%.1f' % x for x in q))
# shader clocks wave 0 of each workgroup spends per step (library built after the step clocks were added)
if hasattr(lib, 'fl_debug_acc_steps'):
    o5 = (C.c_ulonglong * (5 * n))()
    if lib.fl_debug_acc_steps(o5, n) == 0:
        st5 = np.array(list(o5), dtype=np.float64).reshape(n, 5)
        ok = st5[:, 4] > 0
        tot = st5[ok].sum(0)
        print('wave 0 of every workgroup, shader clocks per step of 64 x ILP records (mean over %d steps): run lookup + requests %.0f, waiting for the records %.0f, '
              'palette + tile adds %.0f; per-group work outside the steps %.0f per step (record phase %.0f clocks per step in all)'
              % (tot[4], tot[0] / tot[4], tot[1] / tot[4], tot[2] / tot[4], (tot[3] - tot[0] - tot[1] - tot[2]) / tot[4], tot[3] / tot[4]))
        hot = np.argsort(-st5[:, 4])[:200]
        t2 = st5[hot].sum(0)
        print('  the 200 workgroups with most steps: lookup %.0f, wait %.0f, adds %.0f, outside %.0f clocks per step (%.1f steps per workgroup-wave)'
              % (t2[0] / t2[4], t2[1] / t2[4], t2[2] / t2[4], (t2[3] - t2[0] - t2[1] - t2[2]) / t2[4], t2[4] / 200))
