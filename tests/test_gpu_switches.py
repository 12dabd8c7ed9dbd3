"""Every run-time switch the library reads (include/flame_hip.h, "Run-time switches") renders the frame the default renders:
the switches choose launch orders, work splits and stream layouts, never results.  One child process per setting (the library
reads its environment when a context is created); the frames are compared as the 8-bit pictures a user would get — equal up to
the order of float additions where a switch regroups them (cells that spill to the float accumulator, flushes per launch)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %(repo)r)
import numpy as np
from cuburn_amd import configs, profile, render
gnm, prof = configs.cfg2(samples=2 ** 25)
prof = dict(prof, width=640, height=360)
gprof = profile.wrap(prof, gnm)
mgr = render.RenderManager(device=0, host_seed=42)
rdr = render.Renderer(gnm, gprof)
frames = []
pend = [mgr.queue_frame(rdr, gnm, gprof, 0.5) for _ in range(3)]          # three frames in flight: every lane is used
for evt, h in pend:
    evt.synchronize()
    frames.append(np.array(h, copy=True))
np.save(%(out)r, np.stack(frames))
'''

SWITCHES = [('FLAME_LANES', '1'), ('FLAME_LANES', '3'), ('FLAME_LANES', '4'), ('FLAME_NO_INTRA_OVERLAP', '1'),
            ('FLAME_DE_ORDER', '0'), ('FLAME_DE_ORDER', '1'), ('FLAME_DE_ORDER', '2'), ('FLAME_DE_ORDER', '01201201'),
            ('FLAME_BIN_GANG', '32'), ('FLAME_BIN_GANG', '4'), ('FLAME_BIN_PARTS', '5'), ('FLAME_BIN_ROUNDS', '8'),
            ('FLAME_LAUNCH_ROUNDS', '64'), ('FLAME_BIN_WIDE', '1'), ('FLAME_RTC', '0'),
            # code-generation switches of the per-genome kernel (round 6): the plot of a round inside the next round's xform block
            # or behind its own walk; the batch epilogue compiled once, for any batch length
            ('FLAME_RTC_FLAGS', '-DFL_ITER_MERGE_MAX_XF=0'), ('FLAME_RTC_FLAGS', '-DFL_SORT_FULL_COPY=0 -DFL_SORT_LOCAL_TID=0')]


def render_with(tmp_path, tag, env):
    out = str(tmp_path / ('%s.npy' % tag))
    r = subprocess.run([sys.executable, '-c', CHILD % dict(repo=REPO, out=out)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **env))
    assert r.returncode == 0, (tag, r.stderr[-3000:])
    return np.load(out).astype(np.int16)


@pytest.fixture(scope='module')
def default_frames(built, tmp_path_factory):
    clean = {k: v for k, v in os.environ.items() if not k.startswith('FLAME_')}
    out = str(tmp_path_factory.mktemp('switches') / 'default.npy')
    r = subprocess.run([sys.executable, '-c', CHILD % dict(repo=REPO, out=out)], capture_output=True, text=True, timeout=300, env=clean)
    assert r.returncode == 0, r.stderr[-3000:]
    f = np.load(out).astype(np.int16)
    assert f.shape == (3, 360, 640, 4) and f[..., :3].max() > 100       # a picture, not a blank
    # (the three frames of one run come from consecutive RNG states: different noise, the same picture)
    assert not np.array_equal(f[0], f[1]) and np.abs(f[0] - f[1]).mean() < 8.0
    return f


@pytest.mark.gpu
@pytest.mark.parametrize('name,value', SWITCHES)
def test_switch_renders_the_default_frame(default_frames, tmp_path, name, value):
    f = render_with(tmp_path, '%s_%s' % (name, value), {name: value})
    assert f.shape == default_frames.shape
    d = np.abs(f - default_frames)
    # same walkers, same RNG streams, same samples: the integer histogram is the same whatever the switch; what may differ is the
    # order of float additions (spilled cells, one flush per launch), i.e. the last bit of a float here and there, an 8-bit step rarely
    assert d.max() <= 1, (name, value, int(d.max()), int((d > 1).sum()))
    assert (d != 0).mean() < 2e-3, (name, value, float((d != 0).mean()))
