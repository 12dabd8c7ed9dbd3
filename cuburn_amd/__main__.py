"""
Command line renderer: ``python -m cuburn_amd ID [-d GENOMEDB] [profile options]``.

Same arguments and frame loop as the reference's main.py:29-137 (double-buffered: frame k+1
is queued before frame k is waited for and written), over the HIP path.  ``--list-devices``
lists the HIP devices; ``--print`` prints the blended animation and exits.  Started once per GPU
(``python -m torch.distributed.run --nproc-per-node N -m cuburn_amd ...``) the processes share the
output files between them — the role of the reference's distribute.py (see jobs.py).
"""
import argparse
import os
import sys
import time
import traceback

from . import jobs, profile
from .genome import convert, store


def list_devices():
    import torch
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        print('Device %d (%s): %s, %d CUs, total mem %d' % (i, p.name, getattr(p, 'gcnArchName', '?'),
                                                            p.multi_processor_count, p.total_memory))


def _deliver(encoder, frame, basename):
    """Hand one host frame (None: end of stream) to the output module and write whatever segments it
    has finished as <basename><suffix> (complete files only: .tmp + rename); encoder logs go to stderr."""
    media, logs = encoder.encode(frame)
    jobs.write_segments(media, basename)
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
    """Render every output file of the run; under a one-process-per-GPU launcher (WORLD_SIZE > 1) this
    process takes its share of the files on its own GPU (jobs.py).  Returns the number of files lost."""
    gnm, basename = store.connect(args.genomedb).animation(args.flame, args.half)
    if getattr(args, 'print'):
        print(convert.to_json(gnm))
        return 0
    gprof = profile.wrap(prof, gnm)
    rank, world, local = jobs.world_from_env()
    # deal first, filter second: the share of a process must not depend on what the others have finished
    todo = jobs.deal(profile.enumerate_jobs(gprof, basename, args, resume=False), rank, world)
    unfinished = set(name for name, _ in profile.enumerate_jobs(gprof, basename, args, resume=True if world > 1 else None))
    todo = [job for job in todo if job[0] in unfinished]
    if not todo:
        return 0
    from . import render as R, output
    device = args.device if args.device is not None and args.device >= 0 else (local if world > 1 else 0)
    rmgr = R.RenderManager(device=device)
    rdr = R.Renderer(gnm, gprof, keep=args.keep)
    tag = ('%d: ' % device) if world > 1 or (args.device is not None and args.device >= 0) else ''
    took = [0]

    def render_job(name, times):
        try:
            render_job_(name, times)
        except Exception:
            # a failed job leaves nothing behind: frames still queued are waited for, the encoder is killed
            try:
                from . import _lib
                _lib.load().fl_ctx_sync(rmgr.fb.ctx)
            except Exception:
                pass
            rdr.out.abort()
            raise

    def render_job_(name, times):
        times = list(times)
        rdr.out = output.get_output_for_profile(gprof)       # a fresh encoder per file
        for idx, (evt, frame) in _one_ahead(lambda t: rmgr.queue_frame(rdr, gnm, gprof, t), times):
            while took[0] > 2000 and not evt.query():        # long frames: poll, keep the interpreter responsive
                time.sleep(0.2)
            evt.synchronize()
            took[0] = evt.time()
            _deliver(rdr.out, frame, name)
            if args.rawfn:
                _preview(args.rawfn, frame)
            print('%s%s (%3d/%3d), %dms' % (tag, name, idx, len(times), took[0]), file=sys.stderr)
            sys.stderr.flush()
        _deliver(rdr.out, None, name)                        # flush: video outputs return their segment here

    done, lost = jobs.run_jobs(todo, render_job)
    return len(lost)


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
    return 1 if render(args, prof) else 0


if __name__ == '__main__':
    sys.exit(main())
