"""
CPU test of the multi-GPU path: frame sharding + gather with world_size 2 on the gloo backend
(the GPU run uses the same code on nccl = RCCL).
"""
import os
import subprocess
import sys
import textwrap

from common import REPO

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from cuburn_amd import distributed as D
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=2)
    rank = dist.get_rank()
    nframes = 5
    mine = D.shard(range(nframes))
    assert mine == list(range(nframes))[rank::2], mine
    frames = [torch.full((4, 6, 4), 10 * i + 1, dtype=torch.uint8) for i in mine]
    anim = D.gather_animation(frames, nframes, dst=0)
    if rank == 0:
        assert len(anim) == nframes
        for i, f in enumerate(anim):
            assert f.shape == (4, 6, 4) and int(f[0, 0, 0]) == 10 * i + 1, (i, f[0, 0, 0])
        print('GATHER_OK')
    else:
        assert anim is None
    one = D.gather_frame(torch.full((2, 2), rank, dtype=torch.int32))
    if rank == 0:
        assert [int(t[0, 0]) for t in one] == [0, 1]
    dist.barrier()
    dist.destroy_process_group()
''') % REPO


def test_frame_shard_and_gather_world2(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = str(29600 + os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=port, MASTER_ADDR='127.0.0.1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GATHER_OK' in outs[0]


def test_shard_without_process_group():
    from cuburn_amd import distributed as D
    assert D.shard(range(7)) == list(range(7))
    assert D.shard(range(7), rank=1, world=3) == [1, 4]


SHARD_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
    import numpy as np, torch, torch.distributed as dist
    from cuburn_amd import distributed as D, configs
    from common import prepare, O
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=2)
    rank, world = dist.get_rank(), 2
    gnm, prof = configs.cfg1()
    prof = dict(prof, width=96, height=64)
    total = 200001
    def render(r):
        F = prepare(gnm, prof, nslots=4, host_seed=D.rank_seed(42, r))
        n = D.sample_share(total, r, world)
        h, _, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, 1)
        return h, n
    mine, n = render(rank)
    acc = torch.from_numpy(mine.copy().reshape(-1))
    D.sum_accumulators(acc)                       # the one exchange of sample-sharded rendering
    h0, n0 = render(0); h1, n1 = render(1)
    assert n0 + n1 == total and abs(n0 - n1) <= 1
    assert not np.array_equal(h0, h1)             # ranks use different RNG streams
    assert np.array_equal(acc.numpy().reshape(-1, 4), h0 + h1)
    assert abs(float(acc.numpy().reshape(-1, 4)[:, 3].sum()) - float(h0[:, 3].sum() + h1[:, 3].sum())) < 1e-3
    if rank == 0:
        print('ALLREDUCE_OK')
    dist.barrier()
    dist.destroy_process_group()
''') % (REPO, REPO)


def test_sample_shard_allreduce_world2(tmp_path):
    """Single-frame sample sharding: disjoint RNG streams per rank, one all-reduce of the accumulators."""
    script = tmp_path / 'worker.py'
    script.write_text(SHARD_WORKER)
    port = str(29950 + os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=port, MASTER_ADDR='127.0.0.1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'ALLREDUCE_OK' in outs[0]


def test_sample_share_and_rank_seed():
    from cuburn_amd import distributed as D
    for n in (0, 1, 7, 2 ** 28, 2 ** 28 + 5):
        for world in (1, 2, 3, 8):
            shares = [D.sample_share(n, r, world) for r in range(world)]
            assert sum(shares) == n and max(shares) - min(shares) <= 1
    assert D.rank_seed(42, 0) == 42 and D.rank_seed(None, 0) == 42
    assert len({D.rank_seed(42, r) for r in range(8)}) == 8
