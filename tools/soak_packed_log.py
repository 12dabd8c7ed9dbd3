#!/usr/bin/env python3
"""Round 6 soak of the packed sample log and its accumulate (k_accum_tiles_p3): whole frames of random genomes at random image sizes, sample
counts and walker geometries through fl_iterate's binned path; one line per case with a digest of the flushed density channel (integers:
adds commute, so ANY correct accumulate gives the same bits) and the colour sums.  Run under two builds of the library and diff:
    python tools/soak_packed_log.py 80 > a.txt;  FLAME_HIP_LIB=.../libflame_hip_p0.so python tools/soak_packed_log.py 80 > b.txt;  diff a.txt b.txt
(libflame_hip_p0.so: make EXTRA=-DFL_LOG_PACK3=0 — the 32-bit log and round 5's accumulate kernel.)  Against DIRECT ATOMICS the same frames
differ wherever a cell is hot enough to wrap the packed cell's 10-bit count before its drain (multiples of 1024 lost, a different number every
run: the direct-atomic back-end's documented limit, as in the reference, cuburn/code/iter.py:361-406) — 18 of 60 random cases; the binned
densities of those cases are identical under both builds and sum to the samples in frame."""
import ctypes as C, hashlib, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from cuburn_amd import _lib, profile, render
import test_gpu_random_genomes as T

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SIZES = [(33, 17), (640, 360), (1000, 999), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (4096, 1716), (7680, 4320)]
GEOMS = [(4, 1024), (4, 1536), (8, 512), (8, 1024), (16, 256), (16, 1024)]
lib = _lib.load()
rs = np.random.RandomState(2026)
for case in range(ncases):
    seed = 1000 + case
    gnm, prof = T.random_genome(seed)
    w, h = SIZES[rs.randint(len(SIZES))]
    nw, nslots = GEOMS[rs.randint(len(GEOMS))]
    nsamp = float(2 ** rs.uniform(21.0, 28.6))             # (above 2^28: two launches)
    os.environ['FLAME_NW'] = str(nw)
    prof = dict(prof, width=w, height=h, spp=nsamp / (w * h))
    gprof = profile.wrap(prof, gnm)
    m = render.RenderManager(device=0, nslots=nslots, host_seed=77 + case)
    rdr = render.Renderer(gnm, gprof)
    dim = m.fb.set_dim(w, h)
    g = rdr._handle(m.fb)
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(m.fb.ctx, C.byref(fid)))
    m._copy(rdr, gnm)
    _lib.check(lib.fl_interp(m.fb.ctx, g, dim.w, dim.h, 0.4, 0.02))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(m.fb.ctx, g, dim.w, dim.h, nsamp, m.fuse, _lib.ACCUM_BINNED, C.byref(run)))
    a = m.fb.read('front', (dim.ah * dim.astride, 4), np.float32)
    m.fb.free()
    dens = a[:, 3].astype(np.float64)
    print('case %d: seed %d, %dx%d, %d x %d waves, %d samples run, in frame %.0f, max cell %.0f, density sha1 %s, colour sums %.9g %.9g %.9g' %
          (case, seed, w, h, nslots, nw, run.value, dens.sum(), dens.max(), hashlib.sha1(a[:, 3].tobytes()).hexdigest()[:16],
           float(a[:, 0].astype(np.float64).sum()), float(a[:, 1].astype(np.float64).sum()), float(a[:, 2].astype(np.float64).sum())), flush=True)
