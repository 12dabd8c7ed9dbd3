"""
Job farm: output files ("jobs": a still, or a shard of consecutive frames going through one
encoder) dealt to the GPUs of a node, one process per GPU, no exchange between them.

Role of the reference's distribute.py:131-248 (gevent dispatcher + ssh/pipe workers): jobs are
independent, so the launcher that starts one process per GPU (``python -m torch.distributed.run
--nproc-per-node N -m cuburn_amd ...``; RANK / WORLD_SIZE / LOCAL_RANK in the environment) IS the
dispatcher.  What is kept from the reference's protocol: finished jobs are skipped (resume is forced
on, distribute.py:160-161), a job that raises is retried up to three more times, a process whose
jobs keep failing gives up (distribute.py:216-232), and every file appears under its final name only
when complete (``.tmp`` + rename, distribute.py:210-212).
"""
import os
import sys
import traceback

MAX_RETRIES = 3            # distribute.py:218
MAX_CONSECUTIVE_FAILURES = 4   # distribute.py:229


def world_from_env(environ=None):
    """(rank, world, local_rank) as the per-GPU launcher exports them; (0, 1, 0) when run alone."""
    env = os.environ if environ is None else environ
    world = int(env.get('WORLD_SIZE', '1') or 1)
    rank = int(env.get('RANK', '0') or 0)
    if not 0 <= rank < world:
        raise ValueError('RANK %d outside WORLD_SIZE %d' % (rank, world))
    return rank, world, int(env.get('LOCAL_RANK', rank) or 0)


def deal(jobs, rank, world):
    """This rank's share: jobs rank, rank + world, ... (neighbouring shards cost about the same, so
    a static deal balances as well as the reference's queue of five)."""
    return list(jobs)[rank::world]


def write_segments(media, basename):
    """Write the segments an output module returned as <basename><suffix>, atomically."""
    written = []
    for suffix in media:
        seg = media[suffix]
        final = basename + suffix
        with open(final + '.tmp', 'wb') as fp:
            while True:
                chunk = seg.read(1 << 20)
                if not chunk:
                    break
                fp.write(chunk)
        os.rename(final + '.tmp', final)
        close = getattr(seg, 'close', None)
        if close:
            close()
        written.append(final)
    return written


def run_jobs(jobs, render_job, log=None):
    """
    Run ``render_job(name, times)`` for every job, in order; a job that raises is put back at the end
    of the list, at most MAX_RETRIES times; MAX_CONSECUTIVE_FAILURES failures in a row end the run
    (the GPU or the encoder is gone).  Returns (names done, names given up).
    """
    log = log or (lambda msg: print(msg, file=sys.stderr))
    todo = [(name, times, 0) for name, times in jobs]
    done, lost, streak = [], [], 0
    while todo:
        name, times, tries = todo.pop(0)
        try:
            render_job(name, times)
        except Exception:
            log('job %s failed (attempt %d):\n%s' % (name, tries + 1, traceback.format_exc()))
            streak += 1
            if tries < MAX_RETRIES:
                todo.append((name, times, tries + 1))
            else:
                lost.append(name)
            if streak >= MAX_CONSECUTIVE_FAILURES:
                lost.extend(n for n, _, _ in todo)
                log('giving up after %d consecutive failures; %d jobs left undone' % (streak, len(todo)))
                break
            continue
        streak = 0
        done.append(name)
    return done, lost
