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
from .genome.use import genome_view, profile_view

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
    """
    ``(name, profile dict)`` from parsed arguments (cuburn/profile.py:76-95): a JSON profile file or
    a builtin as the base, ``--still`` pins one unblurred frame (start 1, end 2, frame_width 0 — which,
    with enumerate_times below, is frame 2), then every explicitly given option overrides the base.
    """
    ns = vars(args)
    if ns.get('profile'):
        name = os.path.splitext(os.path.basename(ns['profile'].name))[0]
        prof = json.load(ns['profile'])
    else:
        name = ns['builtin_profile']
        prof = dict(BUILTIN[name])
    if ns.get('still'):
        prof['frame_width'], prof['start'], prof['end'] = 0, 1, 2
    prof.update((k, ns[k]) for k in _OVERRIDES if ns.get(k) is not None)
    if ns.get('codec') is not None:
        prof['output'] = dict(prof.get('output') or {}, type=ns['codec'])
    return name, prof


def wrap(prof, gnm):
    """Genome-adjusted profile view: RefScalars scale the genome's splines (cuburn/profile.py:97-105)."""
    scale = gnm.get('time', {}).get('duration', 1)
    return profile_view(prof, toplevels['profile'], genome_view(gnm, scale))


def enumerate_times(gprof):
    """
    ``[(frame_no, centre times)]`` of the frames to render (cuburn/profile.py:107-127).  The
    animation's unit interval is cut into ``round(fps * duration)`` frames; frame k (1-based) is
    centred on ``(k - 0.5) / nframes``.  With ``shard`` every output holds ``round(fps * shard)``
    consecutive frames and start / end / skip are ignored.  Otherwise ``end`` keeps frames
    ``1..end``, ``start`` then drops the first ``start`` of them (so start=1, end=2 — what --still
    sets — selects frame 2), and every ``skip + 1``-th frame of the rest is rendered.
    """
    nframes = int(round(gprof.fps * gprof.duration))
    step = 1.0 / nframes if nframes > 0 else 0.0
    centres = np.arange(nframes) * step + 0.5 * step
    if gprof.shard:
        per_file = max(1, int(round(gprof.fps * gprof.shard)))
        return [(n + 1, centres[lo:lo + per_file]) for n, lo in enumerate(range(0, nframes, per_file))]
    first = 0 if gprof.start is None else gprof.start
    last = nframes if gprof.end is None else gprof.end
    chosen = list(range(nframes))[:last][first:][::gprof.skip + 1]
    return [(k + 1, [centres[k]]) for k in chosen]


def enumerate_jobs(gprof, basename, args, resume=None):
    """
    ``[(output path without extension, centre times)]`` for every output file of the run
    (cuburn/profile.py:129-159).  Files are numbered ``<dir>/<name>_00001<suffix>`` — or
    ``<dir>/<name>/00001<suffix>`` with ``--subdir``, creating the directory — where ``-n`` replaces
    the genome's own name.  With ``resume`` (argument, or ``--resume`` when the argument is None)
    outputs whose file already exists under the output module's extension are left out.
    """
    from . import output
    stem = os.path.join(args.dir, basename if args.name is None else args.name)
    if args.subdir:
        os.makedirs(stem, exist_ok=True)             # every per-GPU process gets here: no race on the directory
    pattern = stem + ('/' if args.subdir else '_') + '%05d' + args.suffix
    jobs = [(pattern % number, times) for number, times in enumerate_times(gprof)]
    if args.resume if resume is None else resume:
        ext = output.get_suffix_for_profile(gprof)
        jobs = [job for job in jobs if not os.path.isfile(job[0] + ext)]
    return jobs
