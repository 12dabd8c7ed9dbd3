"""bench.py end to end on the GPU box: the N = 1 line, and the N = 2 launch path (bench.py starts its own
two ranks; both render on the one GPU of the box, collectives over gloo) — the 8-GPU run is the
driver's, this keeps its control flow honest."""
import json
import os
import subprocess
import sys

import pytest

from common import REPO

pytestmark = pytest.mark.gpu


def run_bench(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


SMALL = ['--steps', '4', '--warmup', '1', '--min-timed-frames', '8', '--preheat-seconds', '0']


def test_bench_single_gpu_line(built):
    out = run_bench(['--gpus', '1', '--cpu-seconds', '1'] + SMALL)
    assert out['n_gpus'] == 1 and out['steps'] == 4 and out['unit'] == 'Msamples/s' and out['value'] > 1000
    assert out['config']['fuse'] == 256 and out['config']['fuse_short']['fuse'] == 64        # the reference's schedule is the default
    assert out['config']['timed_frames'] == 8
    assert 0 < out['roofline']['frac'] < 1 and out['roofline']['bound'] == 'hbm'
    assert out['cpu_baseline']['kind'] == 'port' and out['cpu_baseline']['value'] > 0
    assert out['config']['samples_per_frame'] >= 2 ** 28
    # counter-derived notes name the config they were measured on and the profiles/ file they come from (no literals in bench.py)
    assert 'cfg2' in out['roofline']['k_iter']['bound'] and 'sq_counters_k_iter_spec.json' in out['roofline']['k_iter']['bound']
    assert 'cfg2' in out['de_filter']['bound'] and 'de_slot_budget' in out['de_filter']['bound']


def test_bench_counter_notes_belong_to_the_config(built):
    """A cfg3 line must not carry cfg2's counters: its notes name cfg3 (and a cfg3 file) or say that nothing was measured."""
    out = run_bench(['--gpus', '1', '--cpu-seconds', '0', '--config', 'cfg3', '--steps', '2', '--warmup', '1', '--min-timed-frames', '2',
                     '--preheat-seconds', '0'])
    for note in (out['roofline']['k_iter']['bound'], out['de_filter']['bound']):
        assert 'cfg3' in note and 'cfg2' not in note
    src = open(os.path.join(REPO, 'bench.py')).read()
    import re
    assert not re.search(r"\d+ M vector|\d+ M scalar|\d+ M branches", src)      # no literal counter values



@pytest.mark.parametrize('shard', ['frames', 'samples'])
def test_bench_starts_two_ranks(built, shard):
    out = run_bench(['--gpus', '2', '--cpu-seconds', '0', '--shard', shard] + SMALL,
                    env={'FLAME_BENCH_BACKEND': 'gloo', 'FLAME_BENCH_DEVICE': '0'})
    assert out['n_gpus'] == 2
    assert out['scaling'] == ('weak' if shard == 'frames' else 'strong')
    per_frame = out['config']['samples_per_frame']
    assert per_frame >= 2 ** 28
    # whole-job value: two ranks' frames (weak) or one frame's samples (strong) over the max-over-ranks time
    want = (2 if shard == 'frames' else 1) * per_frame * out['steps'] / (out['ms_per_step'] * 1e-3 * out['steps']) / 1e6
    assert abs(out['value'] - want) < 0.02 * want
