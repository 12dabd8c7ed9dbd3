"""
CPU test of the multi-GPU path: frame sharding + gather with world_size 2 on the gloo backend
(the GPU run uses the same code on nccl = RCCL).
"""
import os
import subprocess
import sys
import textwrap

import pytest

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


LOOP_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from cuburn_amd import distributed as D
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=2)
    rank = dist.get_rank()

    class Evt(object):                      # stands in for DurationEvent
        def __init__(self, log, i): self.log, self.i = log, i
        def synchronize(self): self.log.append(('done', self.i)); return self

    for nframes, block, depth in ((11, 4, 2), (3, 4, 1), (8, 4, 3), (9, 2, 2)):
        got, log, count = {}, [], [0]
        def sink(r, index, frame):
            got[(r, index)] = frame.clone()
        g = D.FrameGather((3, 5, 4), torch.uint8, torch.device('cpu'), block=block, sink=sink)
        def queue(slot):
            i = count[0]; count[0] += 1
            log.append(('queue', i))
            frame = np.full((3, 5, 4), (17 * i + 5 * rank + 1) %% 251, np.uint8)     # what the device would render
            return (Evt(log, i), frame)
        def stage(slot, h_out):
            slot.copy_(torch.from_numpy(h_out))
        assert D.run_frame_loop(queue, nframes, depth=depth, gather=g, stage=stage) == nframes
        # the loop keeps `depth` frames queued ahead of the one it waits for
        for k in range(nframes - depth):
            assert log.index(('queue', k + depth)) < log.index(('done', k)) < log.index(('queue', k + depth + 1)) if k + depth + 1 < nframes else True
        if rank == 0:
            assert sorted(got) == [(r, i) for r in range(2) for i in range(nframes)], sorted(got)
            for (r, i), f in got.items():
                assert f.shape == (3, 5, 4) and int(f[0, 0, 0]) == (17 * i + 5 * r + 1) %% 251, (r, i)
        else:
            assert not got
        dist.barrier()
    if rank == 0:
        print('LOOP_OK')
    dist.destroy_process_group()
''') % REPO


def test_bench_frame_loop_and_block_gather_world2(tmp_path):
    """bench.py's own loop (distributed.run_frame_loop + FrameGather: frames queued ahead, one
    asynchronous gather per block of frames, partial last block, two alternating blocks) with two
    ranks on gloo: rank 0 receives every frame of every rank, in order, exactly once."""
    script = tmp_path / 'worker.py'
    script.write_text(LOOP_WORKER)
    port = str(29300 + os.getpid() % 300)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=port, MASTER_ADDR='127.0.0.1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'LOOP_OK' in outs[0]


def test_frame_gather_single_process():
    """Without a process group the gather is the identity: the sink sees this rank's frames."""
    import torch
    from cuburn_amd import distributed as D
    got = []
    g = D.FrameGather((2, 2, 4), torch.uint8, torch.device('cpu'), block=3, sink=lambda r, i, f: got.append((r, i, int(f[0, 0, 0]))))
    for i in range(7):
        g.slot().fill_(i + 1)
        g.submit()
    g.flush()
    assert got == [(0, i, i + 1) for i in range(7)]


EMPTY_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from cuburn_amd import distributed as D
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=2)
    rank = dist.get_rank()
    mine = D.shard(range(1))                      # one frame, two ranks: rank 1 has nothing
    frames = [torch.full((2, 3, 4), 7, dtype=torch.uint8) for _ in mine]
    anim = D.gather_animation(frames, 1, dst=0)
    if rank == 0:
        assert len(anim) == 1 and int(anim[0][0, 0, 0]) == 7
        print('EMPTY_OK')
    dist.barrier()
    dist.destroy_process_group()
''') % REPO


def test_gather_animation_with_an_empty_shard(tmp_path):
    """Fewer frames than ranks: the rank without a frame still takes part in the collective
    (it used to raise on zeros_like(None) while the others blocked inside the gather)."""
    script = tmp_path / 'worker.py'
    script.write_text(EMPTY_WORKER)
    port = str(29100 + os.getpid() % 150)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=port, MASTER_ADDR='127.0.0.1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'EMPTY_OK' in outs[0]


CFG4_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from cuburn_amd import distributed as D, configs, profile
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=2)
    rank, world = dist.get_rank(), 2
    gnm, prof = configs.cfg4()
    gprof = profile.wrap(prof, gnm)
    frames = profile.enumerate_times(gprof)                 # the 60 frames of BASELINE configs[3]
    assert len(frames) == 60 and frames[0][0] == 1 and frames[-1][0] == 60
    mine = D.shard(frames)                                   # rank r renders frames r+1, r+1+world, ...
    assert [f[0] for f in mine] == list(range(rank + 1, 61, world))
    td = gprof.frame_width(0.5) / round(gprof.fps * gprof.duration)
    assert abs(td - 1.0 / 60) < 1e-12
    got = {}
    g = D.FrameGather((2, 4, 4), torch.uint8, torch.device('cpu'), block=4,
                      sink=lambda r, i, f: got.__setitem__(r + i * world + 1, (int(f[0, 0, 0]), int(f[0, 0, 1]))))
    it = iter(mine)
    class Evt(object):
        def synchronize(self): return self
    def queue(slot):                                         # stands in for queue_frame: a frame that encodes its number and time
        no, (tc,) = next(it)
        slot[...] = 0
        slot[0, 0, 0] = no
        slot[0, 0, 1] = int(round(float(tc) * 120))
        return (Evt(), None)
    D.run_frame_loop(queue, len(mine), depth=2, gather=g)
    if rank == 0:
        assert sorted(got) == list(range(1, 61))
        for no, (a, b) in got.items():
            assert a == no and b == 2 * no - 1, (no, a, b)    # centre time (no - 0.5) / 60
        print('CFG4_OK')
    dist.barrier()
    dist.destroy_process_group()
''') % REPO


def test_cfg4_sixty_frames_shard_and_gather_world2(tmp_path):
    """BASELINE configs[3] bookkeeping with two ranks: the profile yields 60 frames, rank r takes
    frames r+1, r+3, ...; every frame passes through the block-wise gather and arrives on rank 0
    under its own frame number with its own centre time."""
    script = tmp_path / 'worker.py'
    script.write_text(CFG4_WORKER)
    port = str(29450 + os.getpid() % 100)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=port, MASTER_ADDR='127.0.0.1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'CFG4_OK' in outs[0]


def test_frame_gather_refuses_to_reuse_a_block_with_outstanding_frames():
    """A queue deeper than the gather block would hand out a tensor whose previous frame is still
    being rendered (or not yet gathered): slot() raises instead, and run_frame_loop checks up front."""
    import pytest
    import torch
    from cuburn_amd import distributed as D
    g = D.FrameGather((1, 1, 4), torch.uint8, torch.device('cpu'), block=2)
    for _ in range(4):                               # both blocks handed out, nothing submitted
        g.slot()
    with pytest.raises(RuntimeError):
        g.slot()                                     # would re-enter block 0: a and b never submitted
    g2 = D.FrameGather((1, 1, 4), torch.uint8, torch.device('cpu'), block=1)
    with pytest.raises(ValueError):
        D.run_frame_loop(lambda slot: (None, None), 4, depth=2, gather=g2)


def _bench(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py')] + args, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    return p, lines


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` outside a launcher starts a child torch.distributed.run with two ranks
    (the reference's dispatcher starts its own workers, distribute.py:131-186) and relays rank 0's line;
    --dry-run exercises exactly that control flow (process group, world-size check, barrier, reduction)
    without rendering, so it runs without a GPU.  The N = 1 path starts nothing."""
    import json
    p, lines = _bench(['--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1'])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1
    p, lines = _bench(['--gpus', '1', '--dry-run'])
    assert p.returncode == 0 and json.loads(lines[0])['n_gpus'] == 1
    # a launcher that started the wrong number of ranks is an error, not a silent one-GPU run
    p, lines = _bench(['--gpus', '2', '--dry-run'], env={'WORLD_SIZE': '1', 'RANK': '0'})
    assert p.returncode != 0 and not lines


BAND_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from cuburn_amd import distributed as D
    world = int(os.environ['WORLD'])
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=world)
    rank = dist.get_rank()
    done = 0
    for ah in (1104, 480, 2192, 4352):
        plan = D.band_plan(ah, world)
        if plan is None:
            continue
        accs = [np.random.RandomState(100 * ah + r).rand(ah, 12).astype(np.float32) for r in range(world)]
        total = accs[0].copy()
        for a in accs[1:]:
            total += a                                   # gloo sums in rank order as well
        rows_per, ranges = plan
        assert rows_per %% 16 == 0 and ranges[0][0] == 0 and ranges[-1][1] == ah
        band, top = D.exchange_bands(torch.from_numpy(accs[rank].copy()), plan, rank, world)
        r0, r1 = ranges[rank]
        assert top == (D.BAND_HALO if r0 > 0 else 0)
        want = total[r0 - top:min(r1 + D.BAND_HALO, ah)]
        assert band.shape == want.shape and band.shape[0] %% 16 == 0, (band.shape, want.shape)
        assert np.allclose(band.numpy(), want, rtol=1e-6, atol=0), ah
        done += 1
    assert done >= (2 if world <= 4 else 1), done
    # the collectives object queue_frame_sharded drives its generator with (round 5): same exchange, rows gathered in rank order
    comm = D.DistComm()
    assert (comm.rank, comm.world) == (rank, world)
    plan = D.band_plan(1104, world) or D.band_plan(4352, world)
    ahc = plan[1][-1][1]
    acc = torch.from_numpy(np.random.RandomState(7 + rank).rand(ahc, 12).astype(np.float32))
    b1, t1 = comm.exchange(acc.clone(), plan)
    b2, t2 = D.exchange_bands(acc.clone(), plan, rank, world)
    assert t1 == t2 and torch.equal(b1, b2)
    mine = torch.full((plan[0], 5, 4), rank + 1, dtype=torch.uint8)
    allb = comm.gather_rows(mine)
    assert allb.shape == (world * plan[0], 5, 4)
    assert [int(allb[r * plan[0], 0, 0]) for r in range(world)] == list(range(1, world + 1))
    s = comm.sum(torch.ones(4) * (rank + 1))
    assert float(s[0]) == world * (world + 1) / 2
    if rank == 0:
        print('BANDS_OK')
    dist.barrier()
    dist.destroy_process_group()
''') % REPO


@pytest.mark.parametrize('world', [2, 3, 8])
def test_row_band_exchange(tmp_path, world):
    '''Sample-sharded frames are summed by row bands: every rank ends up with the sum of its own rows plus 224
    halo rows from its neighbours.  RCCL: reduce-scatter; gloo has none and all-reduces, then keeps its own band —
    the isend / irecv halo exchange that follows is the same code on both backends, and here it runs with a
    middle rank (world 3) and with 4K on 8 ranks, whose last band (176 rows) is shorter than the halo.'''
    script = tmp_path / 'worker.py'
    script.write_text(BAND_WORKER)
    port = str(29700 + (os.getpid() + 7 * world) % 200)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD=str(world), PORT=port, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'BANDS_OK' in outs[0]


def test_band_and_halo_plans_are_consistent():
    '''The band plan + halo sizes + peer lists of the RCCL path as a pure function: for every accumulator height of
    the BASELINE configs and 2..16 ranks, every send has a matching receive of the same size on the peer, the rows a
    rank ends up with are exactly [r0 - 224, r1 + 224) clipped to the image, and no rank is left without rows.'''
    from cuburn_amd import distributed as D
    H = D.BAND_HALO
    assert D.band_plan(400, 2) is None and D.band_plan(1104, 1) is None          # bands shorter than the halo; one rank
    seen_short_last = False
    for ah in (1104, 2192, 4352, 480, 8672):
        for world in range(2, 17):
            plan = D.band_plan(ah, world)
            if plan is None:
                rows_per = 16 * -(-ah // (16 * world))
                assert rows_per < H or (world - 1) * rows_per >= ah
                continue
            rows_per, bands = plan
            assert rows_per % 16 == 0 and rows_per >= H and len(bands) == world
            assert bands[0][0] == 0 and bands[-1][1] == ah
            assert all(b[1] > b[0] and b[0] % 16 == 0 for b in bands)            # no empty band
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
            assert all(b[1] - b[0] == rows_per for b in bands[:-1])
            hps = [D.halo_plan(plan, r, ah) for r in range(world)]
            for r, hp in enumerate(hps):
                r0, r1 = bands[r]
                assert hp['top'] == (H if r > 0 else 0)
                assert hp['bot'] == min(H, ah - r1)
                assert (hp['top'] + (r1 - r0) + hp['bot']) % 16 == 0
                for peer, first, rows in hp['sends']:
                    assert abs(peer - r) == 1 and 0 <= first and first + rows <= r1 - r0 and rows > 0
                    # the peer expects exactly these rows: my first rows as its bottom halo, my last rows as its top halo
                    side = 'bot' if peer < r else 'top'
                    match = [x for x in hps[peer]['recvs'] if x[0] == r and x[1] == side]
                    assert len(match) == 1 and match[0][2] == rows, (ah, world, r, peer)
                    # and they are the rows adjacent to the peer's band
                    if side == 'bot':
                        assert first == 0
                    else:
                        assert first + rows == r1 - r0
                for peer, side, rows in hp['recvs']:
                    assert [x for x in hps[peer]['sends'] if x[0] == r and x[2] == rows], (ah, world, r, peer)
            if bands[-1][1] - bands[-1][0] < H:
                seen_short_last = True
    assert seen_short_last                                                       # e.g. 4K on 8 ranks: 176 rows
    assert D.band_plan(2192, 8)[1][-1] == (2016, 2192)


def test_chain_reach_covers_every_spatial_filter():
    '''The band halo must cover the reach of the profile's chain: the DE, and the blurs of haloclip / smearclip.'''
    from cuburn_amd import distributed as D, filters
    assert set(D.FILTER_REACH) <= set(filters.Filter.filter_map)
    assert D.chain_reach(['yuv', 'bilateral', 'logscale', 'colorclip']) == 192
    assert D.chain_reach(['yuv', 'bilateral', 'haloclip', 'smearclip', 'logscale', 'colorclip']) == 207 <= D.BAND_HALO
    assert D.chain_reach(['yuv', 'bilateral', 'bilateral']) > D.BAND_HALO        # such a profile takes the all-reduce path


def test_band_path_only_for_interleaved_frames():
    """Sample-sharded frames take the row-band path only when the output is an interleaved (h, w, 4) frame: the
    encoders' planar YUV frames (3, h, w) cannot be cut into row bands of interleaved pixels and go through the
    all-reduce path like every output did before the band path existed."""
    from cuburn_amd import distributed as D, output, encoders
    from cuburn_amd.render import Framebuffers
    dim = Framebuffers.calc_dim(1920, 1080)
    chain = ['yuv', 'bilateral', 'logscale', 'colorclip']
    assert D.band_path_ok(output.Output(), dim, chain)
    assert D.band_path_ok(output.TiffOutput(), dim, chain) and D.band_path_ok(output.Raw16Output(), dim, chain)
    planar = [encoders.ProResOutput(), encoders.VPxOutput('vp9', pix_fmt='yuv420p'), encoders.VPxOutput('vp9', pix_fmt='yuv444p10')]
    for out in planar:
        assert len(out.shape(dim)) in (1, 3) and not D.band_path_ok(out, dim, chain), type(out).__name__
    assert not D.band_path_ok(output.Output(), dim, ['yuv', 'bilateral', 'bilateral'])     # reach 384 > halo
