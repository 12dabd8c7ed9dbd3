"""
Render profiles: builtin sizes, command-line overlay, genome-adjusted view, frame times.

Same entry points as cuburn/profile.py:10-159 (``BUILTIN``, ``add_args``,
``get_from_args``, ``wrap``, ``enumerate_times``, ``enumerate_jobs``).
"""
import os
import json
import argparse
import numpy as np

from .genome.specs import toplevels
from .genome.use import RefWrapper, SplineWrapper

BUILTIN = {
    '1080p': dict(width=1920, height=1080),
    '720p': dict(width=1280, height=720),
    '540p': dict(width=960, height=540),
    'preview': dict(width=640, height=360, spp=1200, skip=1),
}

_OVERRIDES = 'duration fps frame_width start end skip shard spp width height'.split()


def add_args(parser=None):
    """Add the profile option groups to ``parser`` (a new one if None)."""
    parser = argparse.ArgumentParser() if parser is None else parser
    prof = parser.add_argument_group('Profile options')
    prof.add_argument('-P', '--builtin-profile', choices=list(BUILTIN.keys()), default='720p',
                      help='Set parameters below from a builtin profile. (default: 720p)')
    prof.add_argument('-p', '--profile', type=argparse.FileType(), metavar='PROFILE',
                      help='Set profile from a JSON file.')

    tmp = parser.add_argument_group('Temporal options')
    tmp.add_argument('--duration', type=float, metavar='TIME', help='Override base duration in seconds')
    tmp.add_argument('--fps', type=float, dest='fps', help='Override frames per second')
    tmp.add_argument('--start', metavar='FRAME_NO', type=int, help='First frame to render (1-indexed, inclusive)')
    tmp.add_argument('--end', metavar='FRAME_NO', type=int, help='Last frame to render (1-indexed, exclusive)')
    tmp.add_argument('--skip', dest='skip', metavar='N', type=int, help='Skip N frames between rendered frames')
    tmp.add_argument('--shard', dest='shard', metavar='SECS', type=float,
                     help='Write SECS of output into each file (start/end/skip ignored)')
    tmp.add_argument('--frame_width', metavar='SCALE', type=float, help='Adjustment factor for temporal frame width.')
    tmp.add_argument('--still', action='store_true',
                     help='Render one frame without motion blur (overrides start, end, frame width).')

    spa = parser.add_argument_group('Spatial options')
    spa.add_argument('--spp', type=int, metavar='SPP', help='Set base samples per pixel')
    spa.add_argument('--width', type=int, metavar='PX')
    spa.add_argument('--height', type=int, metavar='PX')

    out = parser.add_argument_group('Output options')
    out.add_argument('--codec', choices=['jpeg', 'png', 'tiff', 'x264', 'vp8', 'vp9', 'prores', 'raw'])
    out.add_argument('-n', metavar='NAME', type=str, dest='name', help='Prefix to use when saving files')
    out.add_argument('--suffix', metavar='NAME', type=str, dest='suffix', default='', help='Suffix for saved files')
    out.add_argument('-o', metavar='DIR', type=str, dest='dir', default='.', help='Output directory')
    out.add_argument('--resume', action='store_true', dest='resume', help="Don't overwrite existing output files")
    out.add_argument('--subdir', action='store_true', help='Use basename as subdirectory of out dir')
    return parser


def get_from_args(args):
    """Profile dict from parsed arguments; returns ``(name, prof)`` (cuburn/profile.py:76-95)."""
    if args.profile:
        name = os.path.basename(args.profile.name).rsplit('.', 1)[0]
        base = json.load(args.profile)
    else:
        name = args.builtin_profile
        base = dict(BUILTIN[args.builtin_profile])
    if args.still:
        base.update(frame_width=0, start=1, end=2)
    for arg in _OVERRIDES:
        if getattr(args, arg, None) is not None:
            base[arg] = getattr(args, arg)
    if args.codec is not None:
        base.setdefault('output', {})['type'] = args.codec
    return name, base


def wrap(prof, gnm):
    """Genome-adjusted profile view: RefScalars scale the genome's splines (cuburn/profile.py:97-105)."""
    scale = gnm.get('time', {}).get('duration', 1)
    return RefWrapper(prof, toplevels['profile'], other=SplineWrapper(gnm, scale=scale))


def enumerate_times(gprof):
    """``[(frame_no, [center_times])]`` before/after start, end, skip (cuburn/profile.py:107-127)."""
    nframes = int(round(gprof.fps * gprof.duration))
    times = np.linspace(0, 1, nframes + 1)
    times = times[:-1] + 0.5 * (times[1] - times[0])
    if gprof.shard:
        s = max(1, int(round(gprof.fps * gprof.shard)))
        return [(i, times[t:t + s]) for i, t in enumerate(range(0, len(times), s), 1)]
    times = list(enumerate([[t] for t in times], 1))
    if gprof.end is not None:
        times = times[:gprof.end]
    if gprof.start is not None:
        times = times[gprof.start:]
    return times[::gprof.skip + 1]


def enumerate_jobs(gprof, basename, args, resume=None):
    """``[(output_basepath, center_times)]`` with optional resume filtering (cuburn/profile.py:129-159)."""
    from . import output
    if args.name is not None:
        basename = args.name
    prefix = os.path.join(args.dir, basename)
    if args.subdir:
        if not os.path.isdir(prefix):
            os.mkdir(prefix)
        prefix_plus = prefix + '/'
    else:
        prefix_plus = prefix + '_'
    frames = [('%s%05d%s' % (prefix_plus, i, args.suffix), t) for i, t in enumerate_times(gprof)]
    resume = args.resume if resume is None else resume
    if resume:
        out_suffix = output.get_suffix_for_profile(gprof)
        frames = [(n, t) for (n, t) in frames if not os.path.isfile(n + out_suffix)]
    return frames
