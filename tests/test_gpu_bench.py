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
    # counter-derived notes name the profiles/ file they come from (no literals in bench.py) and, when the file was measured on the
    # library that is loaded, the config; a file of another build yields no figures (next test)
    assert 'sq_counters_k_iter_spec.json' in out['roofline']['k_iter']['bound'] and 'de_slot_budget' in out['de_filter']['bound']
    for note, frac in ((out['roofline']['k_iter']['bound'], out['roofline']['k_iter']['valu_issue_frac']), (out['de_filter']['bound'], out['de_filter']['valu_frac'])):
        assert ('another build' in note and frac is None) or ('cfg2' in note and 0 < frac < 1.5), (note, frac)
    if out['roofline']['traffic'] is not None:
        meta = json.load(open(os.path.join(REPO, 'profiles', out['roofline']['traffic_source'])))['_meta']
        assert meta['lib_sha256'] == out['roofline']['library_sha256'] and out['roofline']['traffic_note'] is None
    else:
        assert out['roofline']['traffic_note']
    # the DE's denominator: the library's own non-temporal copy kernel, torch's copy_ beside it
    de = out['de_filter']
    assert de['measured_copy_gbps'] > 2000 and de['torch_copy_gbps'] > 2000
    assert abs(de['frac_of_copy'] - de['gbps'] / de['measured_copy_gbps']) < 1e-3
    assert out['config']['stream_lanes']['lanes'] == 2


def test_counter_files_are_quoted_only_for_the_library_they_were_measured_on(built, tmp_path):
    """roofline.traffic, k_iter.valu_issue_frac and de_filter.valu_frac come from counter files under profiles/ that carry the sha256
    of the library they were measured on (tools/pmc_traffic.sh, pmc_sq.sh, de_slot_budget.sh).  Another build of the library — here the
    same library with one byte appended — must print none of them, and say why."""
    import shutil
    from cuburn_amd import _lib
    lib2 = str(tmp_path / 'libflame_hip_other.so')
    shutil.copy(_lib.LIB_PATH, lib2)
    with open(lib2, 'ab') as f:
        f.write(b'\0')
    out = run_bench(['--gpus', '1', '--cpu-seconds', '0'] + SMALL, env={'FLAME_HIP_LIB': lib2})
    r = out['roofline']
    assert r['traffic'] is None and r['traffic_source'] is None and 'another build' in r['traffic_note']
    assert r['k_iter']['valu_issue_frac'] is None and 'another build' in r['k_iter']['bound']
    assert out['de_filter']['valu_frac'] is None and out['de_filter']['traffic'] is None
    assert 0 < r['frac'] < 1                      # what is measured live stays


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
