#!/usr/bin/env python3
"""
bench.py — headline benchmark of the flame hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full frame of BASELINE.json configs[1] ("cfg2": 1920x1080 still, 3 xforms
linear + spherical + swirl, 2^28 samples) through the drop-in entry point
RenderManager.queue_frame: parameter interpolation, chaos-game iteration + flush, the filter
chain (yuv -> bilateral DE -> logscale -> colorclip) and output conversion.  Frames are
independent, so with N GPUs every rank renders its own frames (weak scaling, no collective
in the data path) and the finished 8-bit frames are gathered to rank 0 over RCCL.

Prints ONE JSON line on rank 0:
  value     = write-enabled chaos-game samples per second, whole job (Msamples/s)
  roofline  = dominant kernel (k_iter): algorithmic 16 B/sample (the 8-byte
              read-modify-write of the packed cell) x samples per launch / HIP-event launch time,
              vs the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from the TCC counters
              (profiles/r01_pmc_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
              passes, FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  `pipeline` is the same
              figure over the whole iterate+accumulate chain (k_iter + k_accum_tiles + k_flush).
  cpu_baseline = the oracle's flam3-style chaos game on the host cores (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(gnm, prof, seconds):
    """flam3-style CPU chaos game (oracle/flame_ref.c ref_flam3_render) on a bounded sample."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from common import O, prepare
    F = prepare(gnm, prof)
    cores = usable_cores()
    # grow the sample until one run takes at least ~80 % of the budget (thread start-up makes
    # short probes underestimate the rate), capped at 2^33 samples
    n = (1 << 21) * cores
    while True:
        _, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, cores)
        if secs >= 0.8 * seconds or n >= 2 ** 33:
            break
        n = int(min(2 ** 33, max(2 * n, n * seconds / max(secs, 1e-3))))
    return {'value': round(n / secs / 1e6, 3), 'unit': 'Msamples/s', 'cores': cores, 'kind': 'port',
            'sample': '%d samples of the cfg2 flame (1920x1080 histogram, per-thread private float4 '
                      'accumulators merged at the end), %.1f s wall' % (n, secs)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU baseline budget (0 = skip)')
    ap.add_argument('--preheat-seconds', type=float, default=3.0,
                    help='untimed frames rendered before the W warm-up steps so that the GPU has left its idle '
                         'power state (a fresh box needs seconds of load before sclk ramps up; the iterate kernel '
                         'is latency/ALU-bound and runs ~1.6x slower until then)')
    ap.add_argument('--shard', default='frames', choices=['frames', 'samples'],
                    help="multi-GPU split: whole frames per rank (default, weak scaling, the reference's "
                         "distribute.py model) or the samples of each single frame with one RCCL "
                         "all-reduce of the accumulators (strong scaling; for frames like cfg5)")
    ap.add_argument('--accum', default=os.environ.get('FLAME_ACCUM', 'binned'), choices=['binned', 'atomic'])
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # FLAME_BENCH_BACKEND=gloo + FLAME_BENCH_DEVICE=k: dry run of the multi-rank control flow on
        # a box with fewer GPUs than ranks (every rank renders on device k, collectives on CPU tensors)
        backend = os.environ.get('FLAME_BENCH_BACKEND', 'nccl')
        if 'FLAME_BENCH_DEVICE' in os.environ:
            local = int(os.environ['FLAME_BENCH_DEVICE'])
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    coll_dev = 'cuda' if world == 1 or dist.get_backend() == 'nccl' else 'cpu'
    torch.cuda.set_device(local)

    from cuburn_amd import configs, profile, render, _lib, distributed as D
    gnm, prof = configs.CONFIGS[args.config]()
    gprof = profile.wrap(prof, gnm)
    nslots = int(os.environ['FLAME_NSLOTS']) if 'FLAME_NSLOTS' in os.environ else None     # None: chosen by image size
    mgr = render.RenderManager(device=local, nslots=nslots, host_seed=42 + rank)
    mgr.accum_mode = _lib.ACCUM_BINNED if args.accum == 'binned' else _lib.ACCUM_ATOMIC
    if 'FLAME_FUSE' in os.environ:
        mgr.fuse = int(os.environ['FLAME_FUSE'])
    rdr = render.Renderer(gnm, gprof)
    w, h = gprof.width, gprof.height
    frame = torch.empty((h, w, 4), dtype=torch.uint8, device=coll_dev)
    gathered = [torch.empty_like(frame) for _ in range(world)] if (world > 1 and rank == 0) else None
    tc = 0.5

    def finish(evt, h_out):
        """Wait for a queued frame; multi-GPU: hand it to the RCCL gather (frames are the only exchange)."""
        evt.synchronize()
        if world > 1:
            frame.copy_(torch.from_numpy(h_out), non_blocking=False)
            dist.gather(frame, gathered, dst=0)

    def run(nframes):
        """The double-buffered frame loop of the reference (main.py:64-76): queue frame k+1, then wait for frame k."""
        if args.shard == 'samples':
            for _ in range(nframes):
                evt, _h = D.queue_frame_sharded(mgr, rdr, gnm, gprof, tc, device=local)
                evt.synchronize()
            return
        pending = None
        for _ in range(nframes):
            nxt = mgr.queue_frame(rdr, gnm, gprof, tc)
            if pending is not None:
                finish(*pending)
            pending = nxt
        if pending is not None:
            finish(*pending)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Preheat is timed per rank, so it must not contain collectives (ranks would run different
    # numbers of them and deadlock): plain local frames, no gather / all-reduce.
    t_heat = time.perf_counter()
    while time.perf_counter() - t_heat < args.preheat_seconds:
        pend = None
        for _ in range(4):
            nxt = mgr.queue_frame(rdr, gnm, gprof, tc)
            if pend is not None:
                pend[0].synchronize()
            pend = nxt
        pend[0].synchronize()
    fence()
    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0

    # Kernel-level numbers (roofline of k_iter, DE-filter GB/s) are taken un-overlapped: a second
    # context with ONE stream lane renders a few frames, its HIP-event kernel times are what
    # rocprofv3 --kernel-trace reports for the same command line with FLAME_LANES=1
    # (profiles/).  `value` above comes from the real two-lane pipeline.
    ksteps = max(4, min(args.steps, 16))
    os.environ['FLAME_LANES'] = '1'
    kmgr = render.RenderManager(device=local, nslots=nslots, host_seed=1042 + rank)
    del os.environ['FLAME_LANES']
    kmgr.accum_mode, kmgr.fuse = mgr.accum_mode, mgr.fuse
    krdr = render.Renderer(gnm, gprof)
    # frames are queued back to back (one lane: nothing overlaps, but the GPU never idles between
    # kernels, as in the rocprofv3 runs): a frame-by-frame loop lets the clocks sag between frames
    prev = None
    for k in range(ksteps + 2):
        if k == 2:
            if prev is not None:
                prev.synchronize()
                prev = None
            kmgr.timings_reset()
        e, _ = kmgr.queue_frame(krdr, gnm, gprof, tc)
        if prev is not None:
            prev.synchronize()
        prev = e
    prev.synchronize()
    acc = kmgr.timings()
    acc['samples'] = kmgr.last_nsamples * ksteps
    acc['steps'] = ksteps
    fence()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    job_samples_per_step = mgr.last_nsamples * world            # frames: every rank runs the same workload
    if world > 1 and args.shard == 'samples':
        ns = torch.tensor([mgr.last_nsamples], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ns)
        job_samples_per_step = int(ns.item())

    if rank == 0:
        dim = render.Framebuffers.calc_dim(w, h)
        nbins = dim.ah * dim.astride
        samples_total = job_samples_per_step * args.steps
        iter_s = acc['iter_ms'] * 1e-3
        achieved = 16.0 * acc['samples'] / iter_s / 1e9 if iter_s > 0 else 0.0
        pipe_s = (acc['iter_ms'] + acc['flush_ms']) * 1e-3
        pipe = 16.0 * acc['samples'] / pipe_s / 1e9 if pipe_s > 0 else 0.0
        traffic = None
        try:        # measured offline with tools/pmc_traffic.sh on this workload (KB units, reads x2)
            pmc = json.load(open(os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')))
            key = [k for k in pmc if 'k_iter' in k and ((', 1>' in k) == (args.accum == 'binned'))]
            if key and args.config == 'cfg2':
                c = pmc[key[0]]
                traffic = int((2 * c['FETCH_SIZE']['median_per_launch'] + c['WRITE_SIZE']['median_per_launch']) * 1024)
        except Exception:
            traffic = None
        de_bytes = 512.0 * nbins * acc['steps']            # 64 B/px/direction x 8
        out = {
            'metric': 'Msamples/s into 1920x1080 histogram + DE-filter GB/s vs HBM roofline',
            'value': round(samples_total / elapsed / 1e6, 2),
            'unit': 'Msamples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak' if args.shard == 'frames' else 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: 1920x1080 still, 3 xforms (linear+spherical+swirl), '
                                   '2^28 samples/frame, filters yuv+bilateral+logscale+colorclip, rgba8 out',
                       'samples_per_frame': mgr.last_nsamples if args.shard == 'frames' else job_samples_per_step, 'stream_lanes': 2,
                       'accum': args.accum, 'preheat_s': args.preheat_seconds, 'fuse': mgr.fuse, 'nslots': mgr.fb.nslots, 'frames_per_gpu': args.steps,
                       'parallelism': ('frame-sharded x%d, RCCL gather' if args.shard == 'frames' else 'sample-sharded x%d, RCCL all-reduce of accumulators') % world},
            'roofline': {'bound': 'hbm', 'kernel': 'k_iter', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                         'pipeline': {'kernels': 'k_iter+k_accum_tiles+k_flush', 'achieved': round(pipe, 2), 'frac': round(pipe / HBM_PEAK_GBS, 5)},
                         'avg_launch_ms': round(acc['iter_ms'] / max(acc['launches'], 1), 4),
                         'iter_msamples_per_s': round(acc['samples'] / iter_s / 1e6, 1) if iter_s > 0 else 0.0},
            'de_filter': {'gbps': round(de_bytes / (acc['filter_ms'] * 1e-3) / 1e9, 2) if acc['filter_ms'] > 0 else 0.0,
                          'frac_of_peak': round(de_bytes / (acc['filter_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if acc['filter_ms'] > 0 else 0.0,
                          'note': 'algorithmic 512 B/px over the whole filter chain time (yuv+bilateral+logscale+colorclip)',
                          'filter_ms_per_frame': round(acc['filter_ms'] / acc['steps'], 4)},
            'kernel_ms_per_frame': {'iter': round(acc['iter_ms'] / acc['steps'], 4), 'accum_flush': round(acc['flush_ms'] / acc['steps'], 4),
                                    'filters': round(acc['filter_ms'] / acc['steps'], 4), 'note': 'un-overlapped (single stream lane)'},
        }
        if world == 1 and args.cpu_seconds > 0:
            out['cpu_baseline'] = cpu_baseline(gnm, prof, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
