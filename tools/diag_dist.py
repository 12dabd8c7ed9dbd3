"""Diagnostic: GPU histogram vs CPU oracle (flam3-style and device model) block z-scores."""
import sys, os, ctypes as C
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
from common import O, prepare
from cuburn_amd import configs, profile, render, _lib, mwc

W, H = 480, 270
def blocks(h, d):
    dd = h[:, 3].reshape(d.ah, d.astride).astype(np.float64)
    HH, WW = d.ah // 8 * 8, d.astride // 8 * 8
    return dd[:HH, :WW].reshape(HH // 8, 8, WW // 8, 8).sum((1, 3))
def zs(a, b):
    s = b.sum() / a.sum()
    return (a * s - b) / np.sqrt(b + a * s * s + 1.0)

mgr = render.RenderManager(device=0, nslots=1024, host_seed=42)
lib = _lib.load()
for cfgname in sys.argv[1:] or ['cfg2']:
  for logn in (26, 28):
    gnm, prof = configs.CONFIGS[cfgname](samples=2 ** logn); prof = dict(prof, width=W, height=H)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof); g = rdr._handle(mgr.fb); mgr._copy(rdr, gnm)
    dim = mgr.fb.calc_dim(W, H)
    _lib.check(lib.fl_interp(mgr.fb.ctx, g, W, H, 0.5, 0.0))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(mgr.fb.ctx, g, W, H, float(2 ** logn), 256, 0, C.byref(run)))
    nb = dim.ah * dim.astride
    front = mgr.fb.read('front', (nb, 4), np.float32)
    F = prepare(gnm, prof); d = F['dim']
    ref, _, _ = O.flam3_render(d, F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** logn, os.cpu_count())
    z = zs(blocks(front, d), blocks(ref, d))
    i = np.unravel_index(np.argmax(np.abs(z)), z.shape)
    print(cfgname, 'N=2^%d' % logn, 'GPU vs flam3: z std %.3f max %.2f at block %s (gpu %.0f ref %.0f)  in-frame %.5f vs %.5f' % (
        z.std(), np.abs(z).max(), i, blocks(front, d)[i], blocks(ref, d)[i], front[:, 3].sum() / run.value, ref[:, 3].sum() / 2 ** logn))
    # worst 5 blocks
    idx = np.argsort(-np.abs(z).ravel())[:5]
    print('   worst:', [(tuple(int(v) for v in np.unravel_index(k, z.shape)), round(float(z.ravel()[k]), 2)) for k in idx])

# ---- GPU vs GPU (different seeds), GPU vs CPU device model, device model vs flam3
def gpu_hist(seed, cfgname, logn):
    m = render.RenderManager(device=0, nslots=1024, host_seed=seed)
    gnm, prof = configs.CONFIGS[cfgname](samples=2 ** logn); prof = dict(prof, width=W, height=H)
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof); g = rdr._handle(m.fb); m._copy(rdr, gnm)
    dim = m.fb.calc_dim(W, H)
    _lib.check(lib.fl_interp(m.fb.ctx, g, W, H, 0.5, 0.0))
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(m.fb.ctx, g, W, H, float(2 ** logn), 256, 0, C.byref(run)))
    return m.fb.read('front', (dim.ah * dim.astride, 4), np.float32), (gnm, prof)

logn = 26
hA, (gnm, prof) = gpu_hist(101, 'cfg2', logn)
hB, _ = gpu_hist(202, 'cfg2', logn)
F = prepare(gnm, prof); d = F['dim']; nb = d.ah * d.astride
def st(name, a, b):
    z = zs(blocks(a, d), blocks(b, d))
    print('%-28s z std %.3f  p99.9 %.2f  max %.2f' % (name, z.std(), np.percentile(np.abs(z), 99.9), np.abs(z).max()))
st('GPU(101) vs GPU(202)', hA, hB)
rng = mwc.make_seeds(1024 * 256, 7); pts = np.full((1024 * 256, 4), np.nan, np.float32)
hot = np.zeros(nb // 16, np.uint32); atom = np.zeros(nb, np.uint64); dm = np.zeros((nb, 4), np.float32)
O.iter_launch(O.GEOM_4x64, d, F['packer'].prog, F['params'], F['palette'], rng, pts, 1024, hot, atom, dm, 0, 256 + 256, 256)
O.flush(d, atom, dm, hot)
f3a, _, _ = O.flam3_render(d, F['packer'].prog, F['params'], F['palette'], F['seeds'], 2 ** logn, 64)
f3b, _, _ = O.flam3_render(d, F['packer'].prog, F['params'], F['palette'], F['seeds'][70000:], 2 ** logn, 64)
st('GPU(101) vs devmodel', hA, dm)
st('GPU(101) vs flam3', hA, f3a)
st('devmodel vs flam3', dm, f3a)
st('flam3 vs flam3', f3a, f3b)
