#!/usr/bin/env python3
"""
bench.py — headline benchmark of the flame hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full frame of BASELINE.json configs[1] ("cfg2": 1920x1080 still, 3 xforms
linear + spherical + swirl, 2^28 samples) through the drop-in entry point
RenderManager.queue_frame: parameter interpolation, chaos-game iteration + tile accumulate +
flush, the filter chain (yuv -> bilateral DE -> logscale -> colorclip) and output conversion.
Frames are independent, so with N GPUs every rank renders its own frames (weak scaling, no
collective in the data path); the finished 8-bit frames go straight from fl_output into device
tensors that RCCL gathers to rank 0, four frames per collective, asynchronously.

Order of work: the CPU baseline FIRST (so that GPU activity is contiguous afterwards), then
preheat, W warm-up steps, K timed steps (barrier + synchronize on both sides, max over ranks),
then the kernel-level section on a one-lane context.

Prints ONE JSON line on rank 0:
  value     = write-enabled chaos-game samples per second, whole job (Msamples/s), at the default fuse —
              the reference's 256 write-disabled rounds per walker and frame (cuburn/render.py:215)
  config.fuse_short = the same loop with a fuse of 64 (enough for the BASELINE flames; not the default)
  roofline  = the iterate CHAIN (k_iter + k_accum_tiles + k_flush — the kernels that together perform
              the 8-byte read-modify-write per sample that SURVEY.md §8d's 16 B/sample stand for):
              16 B x samples / HIP-event time of those kernels, vs the 8 TB/s HBM peak.  `traffic` =
              HBM bytes of the chain per frame from the TCC counters (profiles/r0N_pmc_traffic.json, newest round:
              rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per
              MI355X_MICROARCH.md §HBM).  `k_iter` carries that kernel's own launch time and its
              MEASURED bytes (it writes 4-byte log records, not the packed cells).
  de_filter = the DE proper (normalise + 8 direction kernels + un-normalise): 512 B/px algorithmic
              over its own HIP-event time, vs 8 TB/s and vs the copy bandwidth measured here.
  cpu_baseline = the oracle's flam3-style chaos game built -O3 -march=native on this host,
              T = 1 and T = all usable cores (rank 0, N = 1 only)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0    # MI355X_MICROARCH.md, HBM: ~6.3 TB/s achievable


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


def native_oracle():
    """the CPU baseline is built -O3 -march=native for the host it runs on.  The
    portable oracle/libflame_ref.so (x86-64-v3, what the tests use) is the fallback."""
    src = [os.path.join(REPO, 'oracle', f) for f in ('flame_ref.c', 'filters_ref.c')]
    out = os.path.join(tempfile.gettempdir(), 'libflame_ref_native_%d.so' % os.getuid())
    cmd = ['gcc', '-O3', '-march=native', '-fPIC', '-std=gnu11', '-ffp-contract=off', '-fno-fast-math', '-pthread',
           '-shared', '-o', out] + src + ['-lm', '-lpthread']
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=120)
        ctypes.CDLL(out)
        return out, '-O3 -march=native'
    except Exception:
        return None, '-O2 -march=x86-64-v3 (prebuilt; native build failed)'


def cpu_baseline(gnm, prof, seconds):
    """flam3-style CPU chaos game (oracle/flame_ref.c ref_flam3_render) on a bounded sample."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from oracle import oracle as OM
    path, flags = native_oracle()
    if path:
        OM.LIB_PATH = path
    from common import O, prepare
    F = prepare(gnm, prof)
    cores = usable_cores()

    def timed(nthreads, budget):
        # grow the sample until one run takes most of the budget (thread start-up makes short
        # probes underestimate the rate), capped at 2^33 samples
        n = (1 << 21) * nthreads
        while True:
            _, secs, acc = O.flam3_render(F['dim'], F['packer'].prog, F['params'], F['palette'], F['seeds'], n, nthreads)
            if secs >= 0.7 * budget or n >= 2 ** 33:
                return n, secs
            n = int(min(2 ** 33, max(2 * n, n * budget / max(secs, 1e-3))))

    n1, s1 = timed(1, 0.25 * seconds)
    nc, sc = timed(cores, 0.75 * seconds)
    return {'value': round(nc / sc / 1e6, 3), 'unit': 'Msamples/s', 'cores': cores, 'kind': 'port',
            'single_thread': {'value': round(n1 / s1 / 1e6, 3), 'unit': 'Msamples/s', 'cores': 1,
                              'sample': '%d samples, %.1f s' % (n1, s1)},
            'build': 'gcc ' + flags,
            'sample': '%d samples of the cfg2 flame (1920x1080 histogram, per-thread private float4 '
                      'accumulators merged at the end), %.1f s wall' % (nc, sc)}


def copy_bandwidth(torch, device):
    """float4 device copy of 1 GiB: read + write bytes per second (the practical HBM ceiling of this box)."""
    n = 1 << 26
    a = torch.ones((n, 4), dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    del a, b
    return 2 * n * 16 / (ms * 1e-3) / 1e9


def hip_copy_bandwidth(device):
    """The same 1 GiB float4 copy through the library's own kernel (fl_measure_copy: non-temporal loads and stores, HIP events)."""
    import ctypes as C
    from cuburn_amd import _lib
    ms = C.c_float()
    n = 1 << 30
    _lib.check(_lib.load().fl_measure_copy(int(device), n, 10, C.byref(ms)))
    return 2 * n / (ms.value * 1e-3) / 1e9


def lib_sha256():
    """sha256 of the library that is loaded: counter files under profiles/ name the library they were measured on."""
    import hashlib
    from cuburn_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()
    except Exception:
        return None


def newest_profile(regex):
    """Newest file under profiles/ whose NAME matches ``regex`` (group 1 = the round number), or None."""
    import re
    best = None
    for f in os.listdir(os.path.join(REPO, 'profiles')):
        m = re.match(regex + '$', f)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(REPO, 'profiles', f))
    return best[1] if best else None


def launch_ranks(ngpus, argv):
    """
    `python bench.py --gpus N` with N > 1 outside a launcher: start N ranks ourselves (one process
    per GPU, as the reference's dispatcher starts its own workers, distribute.py:131-186) as a CHILD
    `python -m torch.distributed.run`, relay its output and return its exit code.  This process has
    not touched the GPU (torch is not even imported) and never execs.
    """
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ngpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                       # rank 0's JSON line (and anything else a rank prints)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def dry_run(args, rank, world):
    """--dry-run: the multi-rank control flow without any rendering (CPU test of the launcher): process
    group, world-size check, barrier, max-over-ranks reduction, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, 'launched with %d ranks for --gpus %d' % (dist.get_world_size(), args.gpus)
        tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        assert int(tt.item()) == world
        n = dist.get_world_size()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': n, 'steps': args.steps, 'warmup': args.warmup, 'value': None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--dry-run', action='store_true', help='control flow only: no rendering, no GPU (launcher test)')
    ap.add_argument('--min-timed-frames', type=int, default=300,
                    help='the timed region repeats the K steps until it holds at least this many frames '
                         '(K frames of 1.7 ms are a 30 ms window); per-step numbers are reported')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg2')
    ap.add_argument('--cpu-seconds', type=float, default=16.0, help='CPU baseline budget (0 = skip)')
    ap.add_argument('--preheat-seconds', type=float, default=3.0,
                    help='untimed frames rendered before the W warm-up steps so that the GPU has left its idle '
                         'power state (a fresh box needs seconds of load before sclk ramps up)')
    ap.add_argument('--depth', type=int, default=2,
                    help='frames queued ahead of the one being waited for (the reference keeps 1, main.py:64-76; '
                         'a frame takes ~2 ms here, about what the host needs to queue the next one)')
    ap.add_argument('--shard', default='frames', choices=['frames', 'samples'],
                    help="multi-GPU split: whole frames per rank (default, weak scaling, the reference's "
                         "distribute.py model) or the samples of each single frame with one RCCL "
                         "all-reduce of the accumulators (strong scaling; for frames like cfg5)")
    ap.add_argument('--accum', default=os.environ.get('FLAME_ACCUM', 'binned'), choices=['binned', 'atomic'])
    ap.add_argument('--gather-block', type=int, default=4, help='frames per gather collective')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:          # not under a launcher: start the ranks
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)

    from cuburn_amd import configs
    # FLAME_BENCH_SAMPLES: another sample count for the chosen config (geometry experiments; the headline runs without it)
    _ns = os.environ.get('FLAME_BENCH_SAMPLES')
    gnm, prof = configs.CONFIGS[args.config](samples=float(_ns)) if _ns and args.config != 'cfg1' else configs.CONFIGS[args.config]()
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:          # before anything touches the GPU
        cpu = cpu_baseline(gnm, prof, args.cpu_seconds)

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
        assert dist.get_world_size() == args.gpus, 'process group of %d ranks for --gpus %d' % (dist.get_world_size(), args.gpus)
    coll_dev = 'cuda' if world == 1 or dist.get_backend() == 'nccl' else 'cpu'
    torch.cuda.set_device(local)

    from cuburn_amd import profile, render, _lib, distributed as D
    gprof = profile.wrap(prof, gnm)
    nslots = int(os.environ['FLAME_NSLOTS']) if 'FLAME_NSLOTS' in os.environ else None     # None: chosen by image size
    mgr = render.RenderManager(device=local, nslots=nslots, host_seed=42 + rank)
    mgr.accum_mode = _lib.ACCUM_BINNED if args.accum == 'binned' else _lib.ACCUM_ATOMIC
    if 'FLAME_FUSE' in os.environ:
        mgr.fuse = int(os.environ['FLAME_FUSE'])
    rdr = render.Renderer(gnm, gprof)
    w, h = gprof.width, gprof.height
    tc = 0.5
    dev = torch.device('cuda', local) if coll_dev == 'cuda' else torch.device('cpu')
    gather = D.FrameGather((h, w, 4), torch.uint8, dev, block=args.gather_block) if (world > 1 and args.shard == 'frames') else None

    phase_ms = {}                                       # --shard samples: per-phase times of the band path, summed over frames
    phase_evts = []

    def queue(slot):
        if args.shard == 'samples':
            evt, h_out = D.queue_frame_sharded(mgr, rdr, gnm, gprof, tc, device=local)
            if hasattr(evt, 'phases'):
                phase_evts.append(evt)
                if len(phase_evts) > 64:                # resolved long ago: fold into the sums
                    for k, v in phase_evts.pop(0).phases().items():
                        phase_ms[k] = phase_ms.get(k, 0.0) + v
                        phase_ms['_n_' + k] = phase_ms.get('_n_' + k, 0) + 1
            return evt, h_out
        if slot is not None and slot.is_cuda:          # the frame goes straight into the tensor RCCL gathers
            return mgr.queue_frame(rdr, gnm, gprof, tc, dev_out=slot.data_ptr(), host=False)
        return mgr.queue_frame(rdr, gnm, gprof, tc)

    def stage(slot, h_out):                             # CPU collectives (gloo dry run): host frame -> slot
        if slot is not None and not slot.is_cuda:
            slot.copy_(torch.from_numpy(np.asarray(h_out)))

    def run(nframes):
        # (sample-sharded frames are queued ahead like whole frames since round 5: nothing in queue_frame_sharded waits on the host,
        # so frame k+1 iterates under frame k's exchange and filters)
        D.run_frame_loop(queue, nframes, depth=args.depth, gather=gather, stage=stage)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Preheat is timed per rank, so it must not contain collectives (ranks would run different
    # numbers of them and deadlock): plain local frames, no gather / all-reduce.
    # It runs for at least --preheat-seconds and then until two consecutive blocks of frames take the same
    # time within 2 % (the clocks have settled), at most three times as long.
    t_heat = time.perf_counter()
    last = None
    while args.preheat_seconds > 0:
        tb = time.perf_counter()
        D.run_frame_loop(lambda slot: mgr.queue_frame(rdr, gnm, gprof, tc), 16, depth=args.depth)
        torch.cuda.synchronize()
        now = time.perf_counter()
        block = now - tb
        settled = last is not None and abs(block - last) <= 0.02 * last
        last = block
        if (now - t_heat >= args.preheat_seconds and settled) or now - t_heat >= 3.0 * args.preheat_seconds:
            break
    # The timed region holds `reps` back-to-back repetitions of the K steps (K frames of 1.7 ms are a
    # 30 ms window): barrier + synchronize on both sides, per-step numbers reported.
    reps = max(1, -(-args.min_timed_frames // max(args.steps, 1)))
    fence()
    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        run(args.steps)
    fence()
    elapsed = (time.perf_counter() - t0) / reps
    samples_per_frame = mgr.last_nsamples

    # the same loop at the other fuse length: the default is the reference's 256 write-disabled rounds
    # at the start of every frame (cuburn/render.py:215); 64 is what the BASELINE flames need
    fuse_main = mgr.fuse
    fuse_other = 64 if fuse_main != 64 else 256
    mgr.fuse = fuse_other
    run(min(args.warmup, 2))
    fence()
    t1 = time.perf_counter()
    for _ in range(reps):
        run(args.steps)
    fence()
    elapsed_ref = (time.perf_counter() - t1) / reps
    mgr.fuse = fuse_main

    # Kernel-level numbers are taken un-overlapped: a second context with ONE stream lane renders
    # a few frames back to back (nothing overlaps, but the GPU never idles between kernels, as in
    # the rocprofv3 runs of profiles/, which use FLAME_LANES=1 too).
    ksteps = max(4, min(args.steps, 16))
    os.environ['FLAME_LANES'] = '1'
    os.environ['FLAME_NO_INTRA_OVERLAP'] = '1'          # multi-launch frames: drains in series with the iterate kernels
    kmgr = render.RenderManager(device=local, nslots=nslots, host_seed=1042 + rank)
    kmgr.accum_mode, kmgr.fuse = mgr.accum_mode, mgr.fuse
    krdr = render.Renderer(gnm, gprof)

    def kernel_times(fuse):
        kmgr.fuse = fuse
        D.run_frame_loop(lambda slot: kmgr.queue_frame(krdr, gnm, gprof, tc), 2, depth=1)
        kmgr.timings_reset()
        D.run_frame_loop(lambda slot: kmgr.queue_frame(krdr, gnm, gprof, tc), ksteps, depth=1)
        t = kmgr.timings()
        t['samples'] = kmgr.last_nsamples * ksteps
        return t
    acc = kernel_times(fuse_main)
    acc_ref = kernel_times(fuse_other)
    # the native context reads the switches when it is created, which happens lazily on the first frame
    del os.environ['FLAME_LANES']
    del os.environ['FLAME_NO_INTRA_OVERLAP']
    fence()
    if world > 1:
        tt = torch.tensor([elapsed, elapsed_ref], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_ref = float(tt[0].item()), float(tt[1].item())
    job_samples_per_step = samples_per_frame * world            # frames: every rank runs the same workload
    if world > 1 and args.shard == 'samples':
        ns = torch.tensor([samples_per_frame], dtype=torch.float64, device=dev)
        dist.all_reduce(ns)
        job_samples_per_step = int(ns.item())

    sample_phases = None
    if args.shard == 'samples' and phase_evts:
        for e in phase_evts:
            for k, v in e.phases().items():
                phase_ms[k] = phase_ms.get(k, 0.0) + v
                phase_ms['_n_' + k] = phase_ms.get('_n_' + k, 0) + 1
        # milliseconds per frame on torch's stream of rank 0: up to the end of this rank's iterate + flush kernels, the
        # reduce-scatter + halo exchange, the band's filters + conversion, the all-gather + the copy to the host
        sample_phases = dict((k, round(v / phase_ms['_n_' + k], 4)) for k, v in phase_ms.items() if not k.startswith('_n_'))
    if rank == 0:
        dim = render.Framebuffers.calc_dim(w, h)
        nbins = dim.ah * dim.astride
        # the box's streaming ceiling, two ways: the library's own copy kernel (float4, non-temporal, 1 GiB: the denominator of "DE >= 60 % of
        # measured HBM bandwidth") and torch.Tensor.copy_ of the same size beside it (what rounds 1-5 divided by)
        torch_copy_gbs = copy_bandwidth(torch, torch.device('cuda', local)) if world == 1 else None
        copy_gbs = hip_copy_bandwidth(local) if world == 1 else None
        chain_s = (acc['iter_ms'] + acc['flush_ms']) * 1e-3
        chain = 16.0 * acc['samples'] / chain_s / 1e9 if chain_s > 0 else 0.0
        chain_ref_s = (acc_ref['iter_ms'] + acc_ref['flush_ms']) * 1e-3
        # Counter-derived figures come from files committed under profiles/ (tools/pmc_traffic.sh, tools/pmc_sq.sh,
        # tools/de_slot_budget.sh), never from literals here, and only from a file that names THIS library (sha256 of the .so it
        # was measured on): a kernel change that was not re-profiled prints null and the reason instead of last round's bytes.
        sha = lib_sha256()
        tag = '' if args.config == 'cfg2' else args.config + '_'
        traffic, iter_bytes, de_traffic, traffic_note = None, None, None, None
        pmc_file = newest_profile(r'r(\d+)_%spmc_traffic\.json' % tag)
        if pmc_file is None:
            traffic_note = 'no TCC counter file for %s under profiles/ (tools/pmc_traffic.sh)' % args.config
        else:
            try:        # KB units, reads x 2 (MI355X_MICROARCH.md)
                pmc = json.load(open(pmc_file))
                if pmc.get('_meta', {}).get('lib_sha256') != sha:
                    traffic_note = '%s was measured on another build of the library (sha256 %s..., loaded %s...): re-run tools/pmc_traffic.sh' % (
                        os.path.basename(pmc_file), str(pmc.get('_meta', {}).get('lib_sha256'))[:12], str(sha)[:12])
                else:
                    def kb(sub):
                        tot = 0.0
                        for k in pmc:
                            if sub in k and 'FETCH_SIZE' in pmc[k]:
                                c = pmc[k]
                                # (launches of one frame differ in length when it has more than 1024 rounds: launches x MEAN is the frame's sum)
                                per = 'mean_per_launch' if 'mean_per_launch' in c['FETCH_SIZE'] else 'median_per_launch'
                                tot += (2 * c['FETCH_SIZE'][per] + c['WRITE_SIZE'][per]) * 1024
                        return tot
                    if args.accum == 'binned':
                        # per LAUNCH in the file; a frame of more than 2^28 samples has several iterate / accumulate / flush launches
                        nl = max(acc['launches'], 1) / float(ksteps)
                        iter_bytes = int(kb('k_iter'))
                        traffic = int((iter_bytes + kb('k_accum_tiles') + kb('k_flush')) * nl)
                    de_traffic = int(kb('k_de_')) or None
            except Exception as exc:
                traffic, iter_bytes, de_traffic, traffic_note = None, None, None, 'unreadable %s: %r' % (os.path.basename(pmc_file), exc)
        launches = max(acc['launches'], 1)
        iter_launch_s = acc['iter_ms'] * 1e-3 / launches
        de_s = (acc['de_ms'] + acc['de_finish_ms']) * 1e-3 / ksteps
        VALU_NS = 1.21              # one vector-ALU wave instruction per SIMD, eight waves resident (tools/valu_bench.hip, profiles/r03_valu_lds_microbench.txt)
        sq_file = newest_profile(r'r(\d+)_%ssq_counters_k_iter_spec\.json' % tag)
        iter_valu_frac = None
        if args.accum != 'binned':
            # (the counter files are the binned path's: the direct-atomic kernel is another kernel)
            iter_bound = ('direct 64-bit packed atomics, one per sample: the scattered-atomic ceiling of the memory side '
                          '(23.7 G atomics/s whatever the footprint, profiles/r01_atomic_microbench.txt)')
        elif sq_file:
            sq = json.load(open(sq_file))
            if sq.get('_lib_sha256') == sha:
                iter_bound = ('instruction issue across the vector, scalar and branch units: %.0f M vector + %.0f M scalar instructions + %.0f M branches '
                              'per launch, SQ counters of k_iter_spec measured on %s with this library (%s)'
                              % (sq.get('SQ_INSTS_VALU', 0) / 1e6, sq.get('SQ_INSTS_SALU', 0) / 1e6, sq.get('SQ_INSTS_BRANCH', 0) / 1e6,
                                 args.config, os.path.basename(sq_file)))
                if iter_launch_s > 0:
                    # flame-independent yardstick: the share of the launch that the vector ALUs of 1024 SIMDs need for the kernel's own instructions
                    iter_valu_frac = round(sq.get('SQ_INSTS_VALU', 0) * VALU_NS * 1e-9 / 1024.0 / iter_launch_s, 4)
            else:
                iter_bound = '%s was measured on another build of the library: no counter figures (tools/pmc_sq.sh)' % os.path.basename(sq_file)
        else:
            iter_bound = 'no SQ counter file for %s under profiles/ (tools/pmc_sq.sh)' % args.config
        de_file = newest_profile(r'r(\d+)_%sde_slot_budget\.txt' % tag)
        de_valu_frac, de_bound = None, None
        if de_file:
            txt = open(de_file).read()
            if ('lib_sha256: %s' % sha) in txt:
                # slot-equivalents per launched lane, summed over the eight directions' shipped kernels ("full" rows, column 8)
                slots = sum(float(l.split()[7]) for l in txt.splitlines() if len(l.split()) >= 10 and l.split()[1] == 'full' and l.split()[0].isdigit())
                if de_s > 0 and slots > 0:
                    de_valu_frac = round(slots * (nbins / 64.0) / 1024.0 * VALU_NS * 1e-9 / de_s, 4)
                de_bound = ('vector-ALU issue slots: %.0f slot-equivalents per pixel over the eight directions (%s, measured on %s with this library); '
                            'valu_frac = that budget at %.2f ns per slot and SIMD / the measured time; 512 B/px is the algorithmic byte count of the '
                            'reference pass structure' % (slots, os.path.basename(de_file), args.config, VALU_NS))
            else:
                de_bound = '%s was measured on another build of the library: no slot budget (tools/de_slot_budget.sh); 512 B/px is the algorithmic byte count of the reference pass structure' % os.path.basename(de_file)
        else:
            de_bound = 'no slot budget measured on %s under profiles/ (tools/de_slot_budget.sh); 512 B/px is the algorithmic byte count of the reference pass structure' % args.config
        big_ms = (acc['iter_ms'] + acc['flush_ms'] + acc['filter_ms']) / ksteps
        try:        # the library's own rule for FLAME_LANES (flame_abi.hip: 1..4, anything else 2)
            lanes = int(os.environ.get('FLAME_LANES', '2'))
            lanes = lanes if 1 <= lanes <= 4 else 2
        except ValueError:
            lanes = 2
        de_gbs = 512.0 * nbins / de_s / 1e9 if de_s > 0 else 0.0            # 64 B/px/direction x 8
        out = {
            'metric': 'Msamples/s into 1920x1080 histogram + DE-filter GB/s vs HBM roofline',
            'value': round(job_samples_per_step * args.steps / elapsed / 1e6, 2),
            'unit': 'Msamples/s',
            'n_gpus': dist.get_world_size() if dist is not None else 1, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak' if args.shard == 'frames' else 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]: 1920x1080 still, 3 xforms (linear+spherical+swirl), '
                                    '2^28 samples/frame, filters yuv+bilateral+logscale+colorclip, rgba8 out') if args.config == 'cfg2'
                       else 'BASELINE %s (diagnostic run, not the headline workload): %dx%d, %d xforms, %d samples/frame'
                            % (args.config, gprof.width, gprof.height, len(gnm['xforms']), samples_per_frame if args.shard == 'frames' else job_samples_per_step),
                       'walker_waves': mgr.fb.nw, 'walker_slots': mgr.fb.nslots, 'temporal_samples': mgr.fb.ntemporal,
                       'samples_per_frame': samples_per_frame if args.shard == 'frames' else job_samples_per_step,
                       'stream_lanes': {'lanes': lanes, 'sum_of_big_kernels_ms': round(big_ms, 4),
                                        'overlap_ms_per_frame': round(big_ms - elapsed / args.steps * 1e3, 4),
                                        'note': 'sum_of_big_kernels_ms = iterate + accumulate + flush + filter kernels of a frame timed alone on one lane; '
                                                'overlap_ms_per_frame = that sum minus ms_per_step: what the second lane hides (kernels of frame k+1 '
                                                'running in the issue slots and on the CUs that frame k leaves idle, plus copies, clears and launch gaps)'},
                       'accum': args.accum, 'preheat_s': args.preheat_seconds, 'fuse': fuse_main, 'nslots': mgr.fb.nslots,
                       'frames_queued_ahead': args.depth, 'frames_per_gpu': args.steps,
                       'per_genome_kernel': {'iterate_launches': int(acc['spec_launches']), 'interpreter_launches': int(acc['interp_launches']),
                                             'note': 'counted by the library (fl_launch_stats) over the kernel-timing frames'},
                       'fuse_short': {'fuse': fuse_other, 'value': round(job_samples_per_step * args.steps / elapsed_ref / 1e6, 2),
                                      'ms_per_step': round(elapsed_ref / args.steps * 1e3, 3),
                                      'iter_chain_ms_per_frame': round(chain_ref_s / ksteps * 1e3, 4),
                                      'note': 'the default fuse is the reference\'s: one 256-iteration round block per walker and frame '
                                              'of un-plotted iterations (render.py:215); value counts write-enabled samples only'},
                       'timed_frames': reps * args.steps,
                       'parallelism': ('frame-sharded x%d, RCCL gather of device frames, %d frames per collective' % (world, args.gather_block)
                                       if args.shard == 'frames' else 'sample-sharded x%d: reduce-scatter by row bands + halo exchange, band-wise filters, all-gather of 8-bit rows '
                                                                       '(all-reduce when the bands would be shorter than their halo)' % world),
                       'sample_shard_phases_ms': sample_phases},
            'roofline': {'bound': 'hbm', 'kernel': 'k_iter + k_accum_tiles + k_flush (iterate chain: the 8-byte packed-cell RMW per sample)',
                         'achieved': round(chain, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(chain / HBM_PEAK_GBS, 5),
                         'traffic': traffic, 'traffic_source': os.path.basename(pmc_file) if pmc_file and traffic is not None else None,
                         'traffic_measured_in_this_run': False, 'traffic_note': traffic_note, 'library_sha256': sha,
                         'algorithmic_bytes_per_frame': int(16 * acc['samples'] / ksteps),
                         'chain_ms_per_frame': round(chain_s / ksteps * 1e3, 4),
                         'k_iter': {'avg_launch_ms': round(iter_launch_s * 1e3, 4), 'bound': iter_bound, 'valu_issue_frac': iter_valu_frac,
                                    'measured_bytes_per_launch': iter_bytes,
                                    'measured_gbps': round(iter_bytes / iter_launch_s / 1e9, 1) if iter_bytes else None,
                                    'msamples_per_s': round(acc['samples'] / (acc['iter_ms'] * 1e-3) / 1e6, 1) if acc['iter_ms'] > 0 else 0.0},
                         'k_accum_tiles_ms_per_frame': round(acc['accum_ms'] / ksteps, 4),
                         'k_flush_ms_per_frame': round(acc['flush_only_ms'] / ksteps, 4),
                         'note': ('kernel times: one stream lane, the frame loop\'s walker geometry (%d slots).  For frames of up to 2^28 samples the loop '
                                  'runs 1024 slots, up to 2^30 samples 1280, above 1536: fewer un-plotted fuse iterations and LDS left for the other lane; the PIPELINE '
                                  '3-6 %% faster, this chain ALONE ~5 %% slower than at 1536 slots (FLAME_NSLOTS=1536; profiles/r06_experiments.txt section 17)' % mgr.fb.nslots)},
            'de_filter': {'kernels': '8 x k_de_dir (the first normalises the accumulator, the last un-normalises and tone-maps)', 'ms_per_frame': round(de_s * 1e3, 4),
                          'gbps': round(de_gbs, 2), 'frac_of_peak': round(de_gbs / HBM_PEAK_GBS, 5),
                          'measured_copy_gbps': round(copy_gbs, 1) if copy_gbs else None,
                          'frac_of_copy': round(de_gbs / copy_gbs, 5) if copy_gbs else None,
                          'torch_copy_gbps': round(torch_copy_gbs, 1) if torch_copy_gbs else None,
                          'frac_of_torch_copy': round(de_gbs / torch_copy_gbs, 5) if torch_copy_gbs else None,
                          'valu_frac': de_valu_frac,
                          'frac_of_achievable_6300': round(de_gbs / HBM_ACHIEVABLE_GBS, 5),
                          'traffic': de_traffic,
                          'bound': de_bound},
            'kernel_ms_per_frame': {'iter': round(acc['iter_ms'] / ksteps, 4), 'accum_flush': round(acc['flush_ms'] / ksteps, 4),
                                    'filters': round(acc['filter_ms'] / ksteps, 4), 'note': 'un-overlapped (single stream lane)'},
        }
        if cpu is not None:
            out['cpu_baseline'] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
