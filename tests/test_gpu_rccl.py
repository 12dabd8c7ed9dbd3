"""The RCCL branches of cuburn_amd.distributed on the device: a process group of ONE rank on backend "nccl" (= RCCL on
ROCm).  The pool's boxes have one GPU and RCCL refuses two ranks on one device, so the multi-rank exchange itself is covered on gloo
(tests/test_cpu_dist.py, tests/test_gpu_bench.py) — what a single rank CAN pin is everything that differs between the backends: that
RCCL accepts the calls as distributed.py makes them (reduce_scatter_tensor on the padded accumulator, all_gather_into_tensor of the
rows, all_reduce on the zero-copy view of the library's own hipMalloc'd accumulator), on torch's stream, ordered against the native
lanes by events only."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np
import torch
import torch.distributed as dist
from cuburn_amd import configs, profile, render, distributed as D

dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1)
assert dist.get_backend() == 'nccl'
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)

# 1. reduce-scatter by bands: the padded form (rows not a multiple of the band) and the plain one
for ah in (270 + 30, 288):
    rows_per = 16 * -(-ah // 16)
    plan = (rows_per, [(0, ah)])
    acc = torch.randn((ah, 4 * 96), device=dev)
    want = acc.clone()
    band, top = D.exchange_bands(acc, plan, 0, 1)
    torch.cuda.synchronize()
    assert top == 0 and band.shape == want.shape and torch.equal(band, want), (ah, band.shape)

# 2. the row gather and the whole-accumulator sum
comm = D.DistComm()
assert (comm.rank, comm.world) == (0, 1)
mine = torch.arange(64 * 480 * 4, device=dev, dtype=torch.uint8).reshape(64, 480, 4)
allb = comm.gather_rows(mine)
torch.cuda.synchronize()
assert torch.equal(allb, mine)

# 3. the collectives of a sample-sharded frame on the zero-copy views of the library's own accumulator (memory torch did not
#    allocate), queued behind the native lane with events only.  The generator is driven as rank 0 of a world of TWO (a world of one
#    has nothing to exchange); the one-rank RCCL group answers for what it can — the sum over the ranks it has — and the test
#    cuts rank 0's band + halo out of the result, as the reduce-scatter of two ranks would have delivered it.
gnm, prof = configs.cfg2(samples=2 ** 27)
prof = dict(prof, width=480, height=720)                    # (bands of two ranks must be taller than the halo)
gprof = profile.wrap(prof, gnm)
mgr = render.RenderManager(device=0, host_seed=D.rank_seed(42, 0))
rdr = render.Renderer(gnm, gprof)
seen = []
for bands in (False, True):
    steps = D.sharded_frame_steps(mgr, rdr, gnm, gprof, 0.5, 0, 2, device=0, bands=bands)
    reply = None
    try:
        while True:
            req = steps.send(reply)
            seen.append(req[0])
            if req[0] == 'sum':
                assert float(req[1].sum()) > 0.0
                dist.all_reduce(req[1], op=dist.ReduceOp.SUM)          # DistComm.sum's call
                reply = req[1]
            elif req[0] == 'exchange':
                acc, (rows_per, ranges) = req[1], req[2]
                full = torch.empty_like(acc)
                dist.reduce_scatter_tensor(full, acc, op=dist.ReduceOp.SUM)      # exchange_bands' call (one rank: the whole accumulator)
                hp = D.halo_plan(req[2], 0, acc.shape[0])
                r0, r1 = ranges[0]
                reply = (full[r0 - hp['top']:r1 + hp['bot']].clone(), hp['top'])
            elif req[0] == 'gather':
                own = torch.empty_like(req[1])
                dist.all_gather_into_tensor(own, req[1])                # DistComm.gather_rows' call
                assert torch.equal(own, req[1])
                reply = torch.cat([own, torch.zeros_like(own)])         # (rank 1's rows: not rendered here)
            else:
                raise AssertionError(req[0])
    except StopIteration as fin:
        evt, h_out = fin.value
    evt.synchronize()
    img = np.asarray(h_out)
    assert img.shape == (720, 480, 4), img.shape
    top_half = img[:300, :, :3]
    assert top_half.max() > 0, bands                                     # rank 0's rows carry the picture
assert seen == ['sum', 'exchange', 'gather'], seen
# a world of one through queue_frame_sharded itself (its own DistComm on the nccl group): no collective, the plain frame
evt2, h2 = D.queue_frame_sharded(mgr, rdr, gnm, gprof, 0.5, device=0)
evt2.synchronize()
assert np.asarray(h2).shape == (720, 480, 4) and np.asarray(h2)[..., :3].max() > 0

# 4. frames gathered block-wise: the single-rank path on device tensors
got = []
fg = D.FrameGather((270, 480, 4), torch.uint8, dev, block=2, sink=lambda r, idx, fr: got.append((idx, int(fr.sum()))))
for k in range(3):
    fg.slot().fill_(k + 1)
    fg.submit()
fg.flush()
assert [g[0] for g in got] == [0, 1, 2] and [g[1] for g in got] == [(k + 1) * 270 * 480 * 4 for k in range(3)], got
# 5. the two calls of the multi-rank paths that a world of one never reaches through the library: FrameGather's asynchronous gather,
#    and the halo exchange's batched isend / irecv — here with this rank as its own peer (RCCL may refuse a self-peer: then the test
#    says so and the call stays covered on gloo only, tests/test_cpu_dist.py)
blk = torch.arange(2 * 270 * 480 * 4, device=dev, dtype=torch.int32).to(torch.uint8).reshape(2, 270, 480, 4)
recv = [torch.empty_like(blk)]
h = dist.gather(blk, recv, dst=0, async_op=True)                  # FrameGather._launch's call
h.wait()
torch.cuda.synchronize()
assert torch.equal(recv[0], blk)
a = torch.randn((224, 4 * 96), device=dev)
b = torch.empty_like(a)
try:
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0)]):      # exchange_bands' call
        w.wait()
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    print('RCCL_SELF_P2P_OK')
except Exception as exc:                                          # noqa: BLE001 - any refusal is the answer this step records
    print('RCCL_SELF_P2P_REFUSED: %%r' %% (exc,))
# 6. exchange_bands on the accumulator's own reserve: rows the (pretended) ranks do not divide, handed over padded, no copy
ah, rows_per = 300, 304
acc_all = torch.randn((rows_per, 4 * 96), device=dev)
band, top = D.exchange_bands(acc_all[:ah], (rows_per, [(0, ah)]), 0, 1, padded=acc_all)
torch.cuda.synchronize()
assert top == 0 and torch.equal(band, acc_all[:ah])
dist.barrier()
dist.destroy_process_group()
print('RCCL_SINGLE_RANK_OK')
'''


@pytest.mark.gpu
def test_rccl_branches_with_one_rank(built):
    port = 29500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-c', CHILD % dict(repo=REPO, port=port)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'RCCL_SINGLE_RANK_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
