"""
Job farm (cuburn_amd/jobs.py; role of distribute.py:131-248): how output files are dealt to the
per-GPU processes, retried, and written.
"""
import io
import os

import pytest

from cuburn_amd import jobs


def test_world_from_env():
    assert jobs.world_from_env({}) == (0, 1, 0)
    assert jobs.world_from_env({'RANK': '5', 'WORLD_SIZE': '8', 'LOCAL_RANK': '5'}) == (5, 8, 5)
    assert jobs.world_from_env({'RANK': '9', 'WORLD_SIZE': '16', 'LOCAL_RANK': '1'}) == (9, 16, 1)
    with pytest.raises(ValueError):
        jobs.world_from_env({'RANK': '8', 'WORLD_SIZE': '8'})


def test_deal_covers_every_job_exactly_once():
    todo = [('f%05d' % i, [i / 60.0]) for i in range(1, 61)]                  # cfg4: 60 frames
    for world in (1, 2, 4, 8, 7):
        shares = [jobs.deal(todo, r, world) for r in range(world)]
        assert sorted(j for s in shares for j in s) == todo
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
    assert jobs.deal(todo, 3, 8)[:2] == [todo[3], todo[11]]


def test_write_segments_is_atomic_and_closes(tmp_path):
    class Seg(io.BytesIO):
        closed_by_writer = False

        def close(self):
            Seg.closed_by_writer = True
            io.BytesIO.close(self)
    base = str(tmp_path / 'clip_00001')
    out = jobs.write_segments({'_color.h264': Seg(b'C' * 3000000), '_alpha.h264': io.BytesIO(b'A' * 10)}, base)
    assert sorted(out) == [base + '_alpha.h264', base + '_color.h264']
    assert os.path.getsize(base + '_color.h264') == 3000000 and open(base + '_alpha.h264', 'rb').read() == b'A' * 10
    assert Seg.closed_by_writer and not [n for n in os.listdir(str(tmp_path)) if n.endswith('.tmp')]
    assert jobs.write_segments({}, base) == []


def test_failed_jobs_are_retried_then_given_up():
    calls, msgs = [], []
    flaky = {'b': 2, 'd': 99}                    # b fails twice, d always

    def render_job(name, times):
        calls.append(name)
        if flaky.get(name, 0) > 0:
            flaky[name] -= 1
            raise IOError('encoder died on ' + name)

    done, lost = jobs.run_jobs([(n, [0.5]) for n in 'abcde'], render_job, log=msgs.append)
    assert done == ['a', 'c', 'e', 'b'] and lost == ['d']
    assert calls.count('b') == 3 and calls.count('d') == 1 + jobs.MAX_RETRIES
    assert len(msgs) == 2 + 4 and 'encoder died on d' in msgs[-1]


def test_a_dead_gpu_stops_the_run():
    seen = []

    def render_job(name, times):
        seen.append(name)
        raise RuntimeError('device lost')

    done, lost = jobs.run_jobs([(n, [0.5]) for n in 'abcdefgh'], render_job, log=lambda m: None)
    assert done == [] and len(seen) == jobs.MAX_CONSECUTIVE_FAILURES
    assert sorted(lost) == list('abcdefgh')                                    # nothing is reported done that is not
