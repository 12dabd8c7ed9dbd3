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


def render(args, prof):
    gnm, basename = store.connect(args.genomedb).animation(args.flame, args.half)
    if getattr(args, 'print'):
        print(convert.to_json(gnm))
        return
    gprof = profile.wrap(prof, gnm)
    frames = profile.enumerate_jobs(gprof, basename, args)
    if not frames:
        return
    from . import render as R
    rmgr = R.RenderManager(device=args.device or 0)
    rdr = R.Renderer(gnm, gprof, keep=args.keep)
    last_ms = 0

    for name, times in frames:
        def save(buf):
            out, log = rdr.out.encode(buf)
            for suffix, file_like in out.items():
                with open(name + suffix, 'wb') as fp:
                    fp.write(file_like.read())
                if getattr(file_like, 'close', None):
                    file_like.close()
            for key, val in log:
                print('\n=== %s ===\n%s' % (key, val), file=sys.stderr)

        pending = None
        times = list(times)
        for idx, t in enumerate(times + [None]):
            done, pending = pending, (rmgr.queue_frame(rdr, gnm, gprof, t) if t is not None else None)
            if done is None:
                continue
            evt, buf = done
            if last_ms > 2000:              # long frames: poll instead of blocking the interpreter
                while not evt.query():
                    time.sleep(0.2)
            evt.synchronize()
            last_ms = evt.time()
            save(buf)
            if args.rawfn:
                try:
                    buf.tofile(args.rawfn + '.tmp')
                    os.rename(args.rawfn + '.tmp', args.rawfn)
                except Exception:
                    print('Failed to write %s: %s' % (args.rawfn, traceback.format_exc()), file=sys.stderr)
            dev = ('%d: ' % args.device) if args.device is not None and args.device >= 0 else ''
            print('%s%s (%3d/%3d), %dms' % (dev, name, idx, len(times), last_ms), file=sys.stderr)
            sys.stderr.flush()
        save(None)


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
