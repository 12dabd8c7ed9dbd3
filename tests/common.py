"""Shared helpers for the tests: oracle-side frame preparation (CPU only)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import oracle as O                      # noqa: E402
from cuburn_amd import profile, mwc                  # noqa: E402
from cuburn_amd.packer import GenomePacker           # noqa: E402
from cuburn_amd.genome.util import palette_decode    # noqa: E402


def frame_times(gprof, tc):
    """cuburn/render.py:410-411"""
    td = gprof.frame_width(tc) / round(gprof.fps * gprof.duration)
    return tc - 0.5 * td, td


def oracle_params(gnm, packer, dim, ts, td, nts=1024):
    """Parameter blocks of all ``nts`` temporal samples computed by the oracle (by name).  The
    device evaluates one block per walker slot (block i at ts + i*td/nts)."""
    names = ['.'.join(n) for n in packer.packed]
    out = np.zeros((nts, packer.pstride), dtype=np.float32)
    if td == 0:
        out[:] = O.param_block(gnm, names, np.float32(ts), dim).astype(np.float32)
    else:
        tstep = np.float32(np.float32(td) / np.float32(nts))
        for i in range(nts):
            t = np.float32(ts) + np.float32(i) * tstep
            out[i] = O.param_block(gnm, names, float(t), dim).astype(np.float32)
    # the last cumulative density takes whatever is left (>= 1 on device)
    return out


def oracle_palette(gnm, ts, td, rng_pal):
    palsrc = dict((v[0], palette_decode(v[1:])) for v in gnm['palette'])
    ptimes, pvals = zip(*sorted(palsrc.items()))
    return O.interp_palette(np.array(pvals, np.float32), np.array(ptimes, np.float32), ts, td, rng_pal)


def prepare(gnm, prof, tc=0.5, nslots=1024, host_seed=42):
    """Everything the oracle needs to iterate a frame: dims, program, params, palette, seeds."""
    gprof = profile.wrap(prof, gnm)
    packer = GenomePacker(gnm)
    dim = O.calc_dim(gprof.width, gprof.height)
    ts, td = frame_times(gprof, tc)
    seeds = mwc.make_seeds((nslots + 64) * 256, host_seed)
    params = oracle_params(gnm, packer, dim, ts, td, nslots)
    palette, rng_pal = oracle_palette(gnm, ts, td, seeds[nslots * 256:])
    return dict(gprof=gprof, packer=packer, dim=dim, ts=ts, td=td, seeds=seeds, params=params,
                palette=palette, rng_pal_after=rng_pal, nslots=nslots)
