"""
Command line renderer: ``python -m cuburn_amd ID [-d GENOMEDB] [profile options]``.

Same arguments and frame loop as the reference's main.py:29-137 (double-buffered: frame k+1
is queued before frame k is waited for and written), over the HIP path.  ``--list-devices``
lists the HIP devices; ``--print`` prints the blended animation and exits.
"""
import argparse
import os
import sys
import time
import traceback

from . import profile
from .genome import convert, store


def list_devices():
    import torch
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        print('Device %d (%s): %s, %d CUs, total mem %d' % (i, p.name, getattr(p, 'gcnArchName', '?'),
                                                            p.multi_processor_count, p.total_memory))


def _deliver(encoder, frame, basename):
    """Hand one host frame (None: end of stream) to the output module and write whatever segments it
    has finished as <basename><suffix>; encoder logs go to stderr."""
    media, logs = encoder.encode(frame)
    for suffix in media:
        seg = media[suffix]
        with open(basename + suffix, 'wb') as fp:
            fp.write(seg.read())
        close = getattr(seg, 'close', None)
        if close:
            close()
    for title, text in logs:
        print('\n=== %s ===\n%s' % (title, text), file=sys.stderr)


def _preview(path, frame):
    """--raw: the newest frame, replaced atomically so that a viewer never sees half a file."""
    try:
        frame.tofile(path + '.tmp')
        os.rename(path + '.tmp', path)
    except Exception:
        print('Failed to write %s: %s' % (path, traceback.format_exc()), file=sys.stderr)


def _one_ahead(queue, times):
    """Yield (index, finished frame) with the next frame already queued behind it (main.py:64-70)."""
    inflight = None
    for k, t in enumerate(times):
        nxt = queue(t)
        if inflight is not None:
            yield k, inflight
        inflight = nxt
    if inflight is not None:
        yield len(times), inflight


def render(args, prof):
    gnm, basename = store.connect(args.genomedb).animation(args.flame, args.half)
    if getattr(args, 'print'):
        print(convert.to_json(gnm))
        return
    gprof = profile.wrap(prof, gnm)
    jobs = profile.enumerate_jobs(gprof, basename, args)
    if not jobs:
        return
    from . import render as R
    rmgr = R.RenderManager(device=args.device or 0)
    rdr = R.Renderer(gnm, gprof, keep=args.keep)
    tag = ('%d: ' % args.device) if args.device is not None and args.device >= 0 else ''
    took_ms = 0

    for name, times in jobs:
        times = list(times)
        for idx, (evt, frame) in _one_ahead(lambda t: rmgr.queue_frame(rdr, gnm, gprof, t), times):
            while took_ms > 2000 and not evt.query():       # long frames: poll, keep the interpreter responsive
                time.sleep(0.2)
            evt.synchronize()
            took_ms = evt.time()
            _deliver(rdr.out, frame, name)
            if args.rawfn:
                _preview(args.rawfn, frame)
            print('%s%s (%3d/%3d), %dms' % (tag, name, idx, len(times), took_ms), file=sys.stderr)
            sys.stderr.flush()
        _deliver(rdr.out, None, name)                        # flush: video outputs return their segment here


def build_parser():
    parser = argparse.ArgumentParser(prog='python -m cuburn_amd', description='Render fractal flames.')
    parser.add_argument('flame', metavar='ID', type=str, nargs='?', help='Filename or flame ID of genome to render')
    parser.add_argument('-d', '--genomedb', metavar='PATH', type=str, default='.',
                        help="Path to genome database (file or directory, default '.')")
    parser.add_argument('--raw', metavar='PATH', type=str, dest='rawfn',
                        help='Target file for raw buffer, to enable previews.')
    parser.add_argument('--half', action='store_true', help='Use half-loops when converting nodes to animations')
    parser.add_argument('--print', action='store_true', help='Print the blended animation and exit.')
    parser.add_argument('--list-devices', action='store_true', help='List devices and exit.')
    parser.add_argument('--device', metavar='NUM', type=int, help='GPU device number to use.')
    parser.add_argument('--keep', action='store_true', help='Accepted for compatibility (kernels are precompiled).')
    profile.add_args(parser)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.list_devices:
        list_devices()
        return 0
    if not args.flame:
        build_parser().error('a flame ID or file is required')
    pname, prof = profile.get_from_args(args)
    render(args, prof)
    return 0


if __name__ == '__main__':
    sys.exit(main())
