"""
Frame sharding across the GPUs of one node (one process per GPU) and the final gather.

Replaces the ssh/pipe job farm of the reference (distribute.py:131-248): frames of an
animation are independent, so rank r renders frames r, r+world, r+2*world, ... with no
exchange in the data path; finished 8/16-bit frames are gathered to rank 0 with one
collective per round of frames (RCCL over xGMI when the backend is "nccl"; the same code
runs on "gloo" for the CPU tests).

A single large frame can instead be sharded by SAMPLES: every rank iterates
``nsamples/world`` samples with its own RNG streams into its own accumulator.  The float4
accumulators are then summed BY ROW BANDS: a reduce-scatter leaves every rank with the sum of its
own band of rows (16 B x nbins x (N-1)/N per rank over xGMI: half of what an all-reduce moves), the
neighbours exchange ``BAND_HALO`` summed rows, every rank runs the filter chain on its band only
(a band plus halos is an image of its own: the filters are local, and a halo wider than the chain's
reach reproduces the interior exactly), and the finished 8-bit bands are all-gathered.  Without
bands (``bands=False``, images too small for the halo, one rank) the accumulators are all-reduced
and every rank filters the whole frame.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def shard(items, rank=None, world=None):
    """Round-robin share of ``items`` for this rank (all of them when not distributed)."""
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
    return list(items)[rank::world]


def gather_frame(frame, dst=0):
    """
    Gather one same-shaped frame tensor from every rank to ``dst``.
    Returns the list of per-rank tensors on ``dst`` and None elsewhere.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [frame]
    rank, world = dist.get_rank(), dist.get_world_size()
    out = [torch.empty_like(frame) for _ in range(world)] if rank == dst else None
    dist.gather(frame, out, dst=dst)
    return out


def gather_animation(local_frames, nframes, dst=0, shape=None, dtype=None, device=None):
    """
    ``local_frames``: this rank's frames in shard order (tensors of one shape).  Returns the
    full animation in frame order on ``dst`` (list of tensors), None elsewhere.  Ranks whose
    shard is one frame short contribute a dummy in the last round; a rank whose shard is EMPTY
    (nframes < world) has no frame to take the shape from, so the shape / dtype / device travel
    explicitly or are broadcast from the first rank, which always owns frame 0.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if nframes <= 0:
        return [] if rank == dst else None
    rounds = (nframes + world - 1) // world
    result = [None] * nframes if rank == dst else None
    proto = local_frames[0] if local_frames else None
    if world > 1 and shape is None:
        meta = [None]
        if rank == 0:
            meta = [(tuple(proto.shape), proto.dtype, str(proto.device))]
        dist.broadcast_object_list(meta, src=0)
        shape, dtype, dev = meta[0]
        if device is None:
            device = proto.device if proto is not None else (torch.device('cpu') if dev == 'cpu' else torch.device('cuda', torch.cuda.current_device()))
    for k in range(rounds):
        if k < len(local_frames):
            f = local_frames[k]
        elif proto is not None:
            f = torch.zeros_like(proto)
        else:
            f = torch.zeros(shape, dtype=dtype, device=device)
        got = gather_frame(f, dst)
        if rank == dst:
            for r, t in enumerate(got):
                idx = k * world + r
                if idx < nframes:
                    result[idx] = t
    return result


class FrameGather(object):
    """
    The final gather of frame-sharded rendering, off the critical path: every rank renders its own
    frames straight into ``slot()`` (a device tensor when the backend is RCCL: fl_output's
    ``dev_out``, no host bounce), ``submit()`` marks the oldest outstanding slot complete, and every
    ``block`` frames ONE asynchronous gather moves the block to rank ``dst`` (xGMI is point-to-point:
    a few large messages per peer instead of seven small receives per frame on rank 0).  Two
    blocks alternate, so a block is only waited for when it is about to be re-used, 2 x block frames
    later.  All ranks must submit the same number of frames.  ``sink(rank, index, frame)`` is
    called on ``dst`` for every received frame (index = the rank's own frame counter).

    ``slot()`` re-enters a block only when every frame it handed out from it last time has been
    submitted (and therefore its gather launched), i.e. with at most ``block`` frames handed out and not
    yet submitted; otherwise the tensor would still be rendered into, or not yet gathered, and ``slot()``
    raises.  ``run_frame_loop`` has ``depth`` frames outstanding when it asks for a slot, so it needs
    ``depth <= block`` (checked there).
    """

    def __init__(self, shape, dtype, device, block=4, dst=0, sink=None):
        self.rank, self.world = _world()
        self.block, self.dst, self.sink = int(block), dst, sink
        self.blocks = [torch.zeros((self.block,) + tuple(shape), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [[torch.empty_like(b) for _ in range(self.world)] if (self.world > 1 and self.rank == dst) else None
                     for b in self.blocks]
        self.work = [None, None]             # (handle, first frame index, count) per block
        self.done_ev = [None, None]          # device blocks: an event behind the block's gather on torch's stream
        self.n_alloc = self.n_done = 0

    def _harvest(self, b):
        if self.work[b] is None:
            return
        handle, first, count = self.work[b]
        if handle is not None:
            handle.wait()
            if self.blocks[b].is_cuda:
                # Work.wait() on RCCL only makes torch's current stream wait; the frames are written by the render
                # context's own streams, which must not re-use the block before the gather has read it.  The host polls
                # an event recorded behind the gather on torch's stream: when the block comes round again — two blocks of
                # frames later — it has long completed, and the loop does not block (the reference's loop never blocks
                # inside a frame either, distribute.py:107-122); only a gather that is really still running is waited for.
                if self.done_ev[b] is None:
                    self.done_ev[b] = torch.cuda.Event()
                self.done_ev[b].record(torch.cuda.current_stream(self.blocks[b].device))
                if not self.done_ev[b].query():
                    self.done_ev[b].synchronize()
        if self.sink is not None and self.rank == self.dst:
            srcs = self.recv[b] if self.world > 1 else [self.blocks[b]]
            for r, t in enumerate(srcs):
                for k in range(count):
                    self.sink(r, first + k, t[k])
        self.work[b] = None

    def slot(self):
        """Tensor the next frame must be written to (valid until that frame is submitted)."""
        b = (self.n_alloc // self.block) % 2
        if self.n_alloc % self.block == 0:
            # the block was last handed out for frames [n_alloc - 2 * block, n_alloc - block): all of them
            # must have been submitted, or one is still being rendered into the tensor about to be re-used
            if self.n_alloc - self.n_done > self.block:
                raise RuntimeError('FrameGather: %d frames handed out and not submitted; block %d of %d frames would be '
                                   're-used while one of its frames is still outstanding (queue depth must stay below '
                                   'the block size)' % (self.n_alloc - self.n_done, b, self.block))
            self._harvest(b)                 # the gather that last used this block has to be done
        t = self.blocks[b][self.n_alloc % self.block]
        self.n_alloc += 1
        return t

    def _launch(self, b, first, count):
        handle = None
        if self.world > 1:
            handle = dist.gather(self.blocks[b], self.recv[b], dst=self.dst, async_op=True)
        self.work[b] = (handle, first, count)

    def submit(self):
        """The oldest slot handed out by slot() now holds a finished frame."""
        self.n_done += 1
        assert self.n_done <= self.n_alloc
        if self.n_done % self.block == 0:
            self._launch(((self.n_done - 1) // self.block) % 2, self.n_done - self.block, self.block)

    def flush(self):
        """Gather a partly filled last block and wait for everything outstanding."""
        assert self.n_done == self.n_alloc, 'every slot must be submitted before flush()'
        rem = self.n_done % self.block
        cur = (self.n_done // self.block) % 2
        if rem:
            self._launch(cur, self.n_done - rem, rem)
        self._harvest(cur ^ 1)
        self._harvest(cur)
        self.n_alloc = self.n_done = 0


def run_frame_loop(queue_frame, nframes, depth=2, gather=None, stage=None):
    """
    The render loop of main.py:64-76 — queue frame k+1, then wait for frame k — with ``depth``
    frames queued ahead of the one being waited for, and the finished frames handed to a
    FrameGather.  ``queue_frame(slot)`` queues one frame and returns ``(evt, h_out)`` (``slot``:
    the tensor the frame should be rendered into, or None); ``stage(slot, h_out)`` copies a host
    frame into its slot when frames are not rendered into it directly (CPU collectives).
    Returns the number of frames completed.
    """
    if gather is not None and depth > gather.block:
        raise ValueError('run_frame_loop: %d frames queued ahead need gather blocks of at least %d frames (got %d)'
                         % (depth, depth, gather.block))
    pending = []

    def finish(item):
        (evt, h_out), slot = item
        evt.synchronize()
        if gather is not None:
            if stage is not None:
                stage(slot, h_out)
            gather.submit()

    for _ in range(nframes):
        slot = gather.slot() if gather is not None else None
        pending.append((queue_frame(slot), slot))
        if len(pending) > depth:
            finish(pending.pop(0))
    while pending:
        finish(pending.pop(0))
    if gather is not None:
        gather.flush()
    return nframes


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def rank_seed(host_seed, rank=None):
    """
    Host seed of a rank's RNG table for sample-sharded rendering.  mwc.make_seeds(n, seed) draws
    the (multiplier, state, carry) rows from RandomState(seed) (mwc.py:30-47): different seeds
    give every rank different streams.  Rank 0 keeps ``host_seed`` so that a world of 1 is the
    unsharded render.
    """
    if rank is None:
        rank = _world()[0]
    base = 42 if host_seed is None else int(host_seed)
    return base + 7919 * rank


def sample_share(nsamples, rank=None, world=None):
    """This rank's share of a frame's samples; shares differ by at most 1 and sum to nsamples."""
    if world is None:
        rank, world = _world()
    nsamples = int(nsamples)
    return nsamples // world + (1 if rank < nsamples % world else 0)


def sum_accumulators(acc):
    """All-reduce (sum) one accumulator tensor in place across ranks; no-op when not distributed."""
    if _world()[1] > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


class _DeviceArray(object):
    """Zero-copy view of a device buffer of the native context for torch (CUDA array interface)."""
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = dict(shape=(int(nfloats),), typestr='<f4',
                                             data=(int(ptr), False), version=2)


def accumulator_tensor(fb, device, dim=None, wait=True):
    """
    The float4 accumulator of the current frame as a flat float32 torch tensor.  The native
    buffers only ever grow, so their capacity depends on a rank's allocation history: with ``dim``
    the view covers exactly the frame's ah x astride cells, which is what every rank must hand to
    the all-reduce.  ``wait=True`` blocks the host until the context's queued work is done (tests);
    ``wait=False`` returns at once — order torch's stream behind the context's with ``order_streams``.
    """
    from . import _lib
    p, n = C.c_void_p(), C.c_size_t()
    fn = _lib.load().fl_buffer_ptr if wait else _lib.load().fl_buffer_ptr_async
    _lib.check(fn(fb.ctx, None, _lib.BUF['front'], C.byref(p), C.byref(n)))
    nfloats = n.value // 4
    if dim is not None:
        want = int(dim.ah) * int(dim.astride) * 4
        assert want <= nfloats
        nfloats = want
    return torch.as_tensor(_DeviceArray(p.value, nfloats), device=torch.device('cuda', device))


def order_streams(fb, device, ctx_waits):
    """
    Stream dependency between the native context's current lane and torch's current stream, no host wait
    (fl_stream_dependency; the reference orders its two streams the same way, cuburn/render.py:358-364,419-430).
    ``ctx_waits=False``: what torch queues next sees everything the context has queued; ``True``: the other way round.
    """
    from . import _lib
    st = torch.cuda.current_stream(device).cuda_stream
    _lib.check(_lib.load().fl_stream_dependency(fb.ctx, C.c_void_p(st), 1 if ctx_waits else 0))


# Rows of its input that one output row of a filter depends on, on either side (the reach of the chain is the sum
# over the profile's filter order).  bilateral: 8 directions x (15 taps + the 9 rows of the two nested density
# blurs at the outermost tap, cuburn/filters.py:62-95) = 192; haloclip: den_blur_1c on patterns 2 and 3 = 2 x 3
# rows; smearclip: full_blur on patterns 2, 3, 0, 1 = 3 x 3 rows (pattern 0 is horizontal)
# (cuburn/filters.py:118-170); everything else is per pixel.
FILTER_REACH = {'bilateral': 8 * 24, 'haloclip': 6, 'smearclip': 9}
# Rows of summed accumulator a band carries beyond its own rows on either side: a multiple of 16 (a band with its
# halos must be a valid accumulator height), at least the reach of the profile's chain (checked per frame).
BAND_HALO = 224


def chain_reach(filter_names):
    """Rows either side that the output of the filter chain ``filter_names`` depends on."""
    return sum(FILTER_REACH.get(n, 0) for n in filter_names)


def band_plan(ah, world, halo=BAND_HALO):
    """
    Row bands of an accumulator of ``ah`` rows for ``world`` ranks: ``(rows_per, [(r0, r1)] per rank)`` with
    every r0 / r1 a multiple of 16 and equal ``rows_per`` (what reduce-scatter needs; the last band may be
    short), or None when bands make no sense: one rank, a band shorter than its halo, or a rank that would be
    left without rows (the callers then all-reduce and filter the whole frame everywhere).
    """
    if world < 2:
        return None
    rows_per = 16 * -(-ah // (16 * world))
    if rows_per < halo or (world - 1) * rows_per >= ah:
        return None
    return rows_per, [(r * rows_per, min((r + 1) * rows_per, ah)) for r in range(world)]


def band_path_ok(out, dim, filter_names, halo=BAND_HALO):
    """
    Row bands need an interleaved 8 / 16-bit frame (a band's rows are a slice of the frame: the encoders' planar
    formats, shape (3, h, w), are not) and a halo that covers the reach of THIS profile's filter chain; anything
    else takes the all-reduce path, where every rank filters and converts the whole frame.
    """
    interleaved = out.dtype in ('u1', 'u2') and tuple(out.shape(dim)) == (dim.h, dim.w, 4)
    return bool(interleaved and chain_reach(filter_names) <= halo)


def halo_plan(plan, rank, ah, halo=BAND_HALO):
    """
    What ``rank`` sends and receives in the halo exchange that follows the reduce-scatter — a pure function of
    the band plan, so that every (sender, receiver) pair can be checked without a process group.  Returns
    ``dict(top, bot, sends, recvs)``: ``top`` / ``bot`` = halo rows received above / below the band (0 at the
    image's own edges; the band below may be the short last one, which then sends what it has);
    ``sends`` = [(peer, first row within this rank's band, rows)], ``recvs`` = [(peer, 'top' | 'bot', rows)],
    both in the order up, down.
    """
    rows_per, bands = plan
    r0, r1 = bands[rank]
    n = r1 - r0
    top = halo if r0 > 0 else 0
    bot = min(halo, ah - r1)
    sends, recvs = [], []
    if top:                                     # the band above always has rows_per >= halo rows
        sends.append((rank - 1, 0, min(halo, n)))
        recvs.append((rank - 1, 'top', top))
    if bot:                                     # only a full band has a band below it
        sends.append((rank + 1, n - halo, halo))
        recvs.append((rank + 1, 'bot', bot))
    return dict(top=top, bot=bot, sends=sends, recvs=recvs)


class ShardBuffers(object):
    """
    The device tensors a sample-sharded frame needs besides the accumulator — the band with its halos, the 8-bit band, the
    gathered frame — allocated once per (frame geometry, world, rank) and re-used by every frame: all of a frame's torch
    operations are queued on ONE stream in frame order, and the native lanes are ordered against that stream by events
    (order_streams), so a frame's tensors are free again by the time the next frame's operations on the same stream reach
    them.  The reference's loop allocates nothing per frame (distribute.py:107-122, cuburn/render.py:107-113).
    """
    _cache = {}

    @classmethod
    def clear(cls):
        cls._cache.clear()

    @classmethod
    def get(cls, device, key, make):
        k = (str(device),) + tuple(key)
        if k not in cls._cache:
            if len(cls._cache) > 16:
                cls._cache.clear()
            cls._cache[k] = make()
        return cls._cache[k]


def exchange_bands(acc2d, plan, rank, world, halo=BAND_HALO, padded=None):
    """
    ``acc2d``: this rank's accumulator as a (ah, row_floats) tensor.  Sums it over the ranks by bands and
    returns ``(band, top)``: the summed rows ``[r0 - top, r1 + bottom)`` of this rank's band with its
    halos (``top`` / ``bottom`` = halo, or 0 at the image's own edges) — a persistent tensor per geometry
    (ShardBuffers), valid until the next call with the same geometry on this stream.
    Collective: every rank calls it.  One reduce-scatter (gloo, the CPU tests' backend, has none: all-reduce and
    keep the own band) followed by the neighbour exchange of ``halo_plan`` — the same isend / irecv code on
    both backends.  ``padded``: the accumulator with ``rows_per * world`` rows where it lies (the rows behind
    ``ah`` hold anything: their sums land behind the last band's own rows and are dropped; see fl_reserve) — without
    it an accumulator whose rows the ranks do not divide is copied into a persistent zero-padded tensor.
    """
    rows_per, bands = plan
    ah, rowf = acc2d.shape
    r0, r1 = bands[rank]
    hp = halo_plan(plan, rank, ah, halo)
    top, bot = hp['top'], hp['bot']
    key = ('band', ah, rowf, world, rank, halo, acc2d.dtype)
    # one tensor for [top halo | core (a whole band: the reduce-scatter's output) | room for the bottom halo]
    buf = ShardBuffers.get(acc2d.device, key, lambda: torch.empty((top + rows_per + bot, rowf), dtype=acc2d.dtype, device=acc2d.device))
    core = buf[top:top + rows_per]
    n = r1 - r0
    if dist.get_backend() == 'nccl':
        src = acc2d
        if rows_per * world != ah:                       # reduce-scatter wants equal chunks
            if padded is not None:
                assert tuple(padded.shape) == (rows_per * world, rowf)
                src = padded
            else:
                src = ShardBuffers.get(acc2d.device, ('pad', ah, rowf, world, acc2d.dtype),
                                       lambda: torch.zeros((rows_per * world, rowf), dtype=acc2d.dtype, device=acc2d.device))
                src[:ah].copy_(acc2d)
        dist.reduce_scatter_tensor(core, src, op=dist.ReduceOp.SUM)
    else:
        dist.all_reduce(acc2d, op=dist.ReduceOp.SUM)
        core[:n].copy_(acc2d[r0:r1])
    # the bottom halo lands directly behind the band's own rows (the last band is short and has none)
    sides = dict(top=buf[:top], bot=buf[top + n:top + n + bot])
    ops = [dist.P2POp(dist.isend, core[first:first + rows], peer) for peer, first, rows in hp['sends']]
    ops += [dist.P2POp(dist.irecv, sides[side], peer) for peer, side, rows in hp['recvs']]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return buf[:top + n + bot], top


def filter_band(mgr, rdr, gprof, dim, band, tc, device, convert=True):
    """
    Run the profile's filter chain and the output conversion on ``band`` — summed accumulator rows, (rows,
    astride * 4) floats, rows a multiple of 16 — as an image of its own in ``mgr``'s context.  The filters'
    host-derived scalars are those of the FULL frame (``dim``: the spatial deviation scales with the width, the
    log scale with the frame's area).  Returns the band's finished 8-bit pixels as a device tensor of
    (rows - 24, w, 4): row j is accumulator row j + 12 of the band, as in a whole frame.  ``convert=False``
    stops after the filters (tests: the float result is in the context's front buffer) and returns (None, bdim).
    """
    from . import _lib
    from .render import Dimensions
    rows = int(band.shape[0])
    assert rows % 16 == 0 and rows >= 32 and band.shape[1] == dim.astride * 4
    bdim = Dimensions(dim.w, rows - 2 * mgr.fb.gutter, dim.aw, rows, dim.astride)
    assert mgr.fb.calc_dim(bdim.w, bdim.h) == bdim
    # torch's copy into the front buffer after whatever the lane still does with it, the filters after the copy: stream
    # dependencies, no host wait (round 4 synchronised the host twice here)
    order_streams(mgr.fb, device, ctx_waits=False)
    front = accumulator_tensor(mgr.fb, device, dim, wait=False)
    front[:band.numel()].copy_(band.reshape(-1))
    order_streams(mgr.fb, device, ctx_waits=True)
    for filt in rdr.filts:
        params = getattr(gprof.filters, filt.name)
        filt._run(mgr.fb, bdim, filt.scalars(gprof, params, dim, tc))
    if not convert:
        return None, bdim
    odt = torch.uint8 if rdr.out.dtype == 'u1' else torch.int16
    out = ShardBuffers.get(front.device, ('out', id(mgr), bdim.h, bdim.w, odt), lambda: torch.empty((bdim.h, bdim.w, 4), dtype=odt, device=front.device))
    rdr.out.convert(mgr.fb, gprof, bdim)
    rdr.out.copy(mgr.fb, bdim, dev_out=out.data_ptr(), host=False)
    order_streams(mgr.fb, device, ctx_waits=False)                  # torch reads `out` behind the conversion
    return out, bdim


class TorchFrameEvent(object):
    """
    Completion handle of a frame whose last step ran on torch's stream (the band path's all-gather and D2H copy):
    the interface of render.DurationEvent (cuburn/render.py:26-38) on torch events.  The frame's host buffer is
    valid only after ``synchronize()`` (or a true ``query()``): the copy into it is asynchronous.  ``keep`` holds
    the tensors that copy still reads; ``marks`` = [(name, event)] recorded along the way on torch's stream —
    ``phases()`` gives the milliseconds between them (iterate / exchange / filter / gather).
    """

    def __init__(self, marks, keep=None):
        self._marks, self._keep, self._ms = marks, keep, None

    def synchronize(self):
        if self._ms is None:
            self._marks[-1][1].synchronize()
            self._ms = self._marks[0][1].elapsed_time(self._marks[-1][1])
            self._keep = None
        return self

    _finalise = synchronize          # Framebuffers._drop_ctx resolves outstanding handles before it destroys a context

    def query(self):
        return self._ms is not None or self._marks[-1][1].query()

    def time(self):
        return self.synchronize()._ms

    def phases(self):
        self.synchronize()
        return dict((b[0], a[1].elapsed_time(b[1])) for a, b in zip(self._marks[:-1], self._marks[1:]))


class DistComm(object):
    """The collectives of a sample-sharded frame on torch.distributed (RCCL when the backend is nccl)."""

    def __init__(self):
        self.rank, self.world = _world()

    def exchange(self, acc2d, plan, padded=None):
        return exchange_bands(acc2d, plan, self.rank, self.world, padded=padded)

    def gather_rows(self, mine):
        allb = ShardBuffers.get(mine.device, ('allb', id(self), self.world, tuple(mine.shape), mine.dtype),
                                lambda: torch.empty((self.world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device))
        if dist.get_backend() == 'nccl':
            dist.all_gather_into_tensor(allb, mine)
        else:
            dist.all_gather(list(allb.view((self.world,) + tuple(mine.shape)).unbind(0)), mine)
        return allb

    def sum(self, acc):
        return sum_accumulators(acc)


def sharded_frame_steps(mgr, rdr, gnm, gprof, tc, rank, world, device=None, copy=True, bands=True):
    """
    One sample-sharded frame as a generator: it yields at the frame's collectives — ``('exchange', acc2d, plan, padded)``
    (``padded``: the same accumulator with ``rows_per * world`` rows, or None), to be answered (``send``) with ``(band, top)``; ``('gather', mine)``, answered with the gathered rows of all ranks;
    ``('sum', acc)``, answered with anything once the accumulator has been summed in place — and returns
    ``(evt, h_out)`` (StopIteration.value).  queue_frame_sharded drives it with torch.distributed; a test can drive the
    generators of several virtual ranks of one process.  Nothing in here waits on the host: the native context's
    lane and torch's stream are ordered by events (order_streams), so that with two frames queued ahead the next
    frame's iterate kernels run under this frame's exchange and filters.
    """
    from . import _lib
    from .render import DurationEvent
    lib = _lib.load()
    if device is None:
        device = torch.cuda.current_device()
    fb = mgr.fb
    total = gprof.spp(tc) * gprof.width * gprof.height
    # the walker geometry follows the NOMINAL share, the same on every rank (a rank's own share differs by one sample)
    dim = fb.set_dim(gprof.width, gprof.height, nsamples=int(total) // world)
    td = gprof.frame_width(tc) / round(gprof.fps * gprof.duration)
    ts = tc - 0.5 * td
    g = rdr._handle(fb)
    fid = C.c_uint32()
    _lib.check(lib.fl_frame_begin(fb.ctx, C.byref(fid)))
    plan = band_plan(dim.ah, world) if (bands and band_path_ok(rdr.out, dim, [f.name for f in rdr.filts])) else None
    if world > 1 and plan is not None and plan[0] * world != dim.ah:
        # rows the ranks do not divide: this lane's accumulator gets room for rows_per * world rows (once: the buffers only
        # grow), so that the reduce-scatter can run on it where it lies
        _lib.check(lib.fl_reserve(fb.ctx, dim.w, plan[0] * world - 2 * fb.gutter))
    if copy:
        mgr._copy(rdr, gnm)
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(device))
        marks.append((name, e))
    mark('start')
    _lib.check(lib.fl_interp(fb.ctx, g, dim.w, dim.h, ts, td))
    nsamps = sample_share(total, rank, world)
    run = C.c_uint64()
    _lib.check(lib.fl_iterate(fb.ctx, g, dim.w, dim.h, float(nsamps), mgr.fuse,
                              mgr.resolve_accum_mode(dim), C.byref(run)))
    mgr.last_nsamples = run.value
    if world > 1 and plan is not None:
        # row bands: reduce-scatter + halo exchange, filter and convert the band, all-gather the 8-bit rows
        order_streams(fb, device, ctx_waits=False)                  # torch's stream behind the iterate + flush kernels
        mark('iterate')
        rows_per, ranges = plan
        flat = accumulator_tensor(fb, device, None, wait=False)     # the whole buffer: the frame's rows, and the reserve behind them
        rowf = dim.astride * 4
        acc = flat[:dim.ah * rowf].view(dim.ah, rowf)
        padded = flat[:rows_per * world * rowf].view(rows_per * world, rowf) if flat.numel() >= rows_per * world * rowf else None
        band, top = yield ('exchange', acc, plan, padded)
        mark('exchange')
        r0, r1 = ranges[rank]
        gut = fb.gutter
        odt = torch.uint8 if rdr.out.dtype == 'u1' else torch.int16
        # (zeroed once: the rows a frame does not write — the frame's own gutter rows, in the first and the last band — are never written)
        mine = ShardBuffers.get(acc.device, ('mine', id(mgr), rows_per, dim.w, odt), lambda: torch.zeros((rows_per, dim.w, 4), dtype=odt, device=acc.device))
        out, bdim = filter_band(mgr, rdr, gprof, dim, band, tc, device)
        # image rows of this band: accumulator rows [r0, r1) less the frame's own gutter rows
        y0, y1 = max(r0 - gut, 0), min(r1 - gut, dim.h)
        if y1 > y0:
            j0 = y0 + gut - (r0 - top) - gut             # band output row of image row y0
            mine[y0 + gut - r0:y1 + gut - r0] = out[j0:j0 + (y1 - y0)]
        mark('filter')
        allb = yield ('gather', mine)
        frame = allb[gut:gut + dim.h]
        # asynchronous copy into the pinned frame buffer; the handle completes when the copy has (render.py:26-38)
        h_out = fb.host_buffer(rdr.out.shape(dim), rdr.out.dtype)
        torch.from_numpy(h_out.view(np.uint8 if rdr.out.dtype == 'u1' else np.int16)).copy_(frame, non_blocking=True)
        mark('gather')
        evt = TorchFrameEvent(marks, keep=(allb, frame, out, mine, band))
        fb._torch_stream_used = device                           # (Framebuffers.free waits for torch's stream too: the copy above is on it)
        fb._track(evt)
        return evt, h_out
    if world > 1:
        order_streams(fb, device, ctx_waits=False)
        acc = accumulator_tensor(fb, device, dim, wait=False)
        yield ('sum', acc)
        order_streams(fb, device, ctx_waits=True)                   # the filters behind the all-reduce
    for filt in rdr.filts:
        filt.apply(fb, gprof, getattr(gprof.filters, filt.name), dim, tc)
    rdr.out.convert(fb, gprof, dim)
    h_out = rdr.out.copy(fb, dim)
    return DurationEvent(fb, fid.value), h_out


def queue_frame_sharded(mgr, rdr, gnm, gprof, tc, device=None, copy=True, bands=True, comm=None):
    """
    RenderManager.queue_frame for ONE frame split by samples over all ranks.  Every rank must
    call it (it contains the collectives) with a RenderManager built with
    ``host_seed=rank_seed(seed)``; every rank ends up with the finished frame.
    Returns ``(evt, h_out)`` like queue_frame (cuburn/render.py:374-434); ``h_out`` is valid once
    ``evt.synchronize()`` has returned (or ``evt.query()`` is true): nothing in here waits on the host.
    """
    comm = comm or DistComm()
    steps = sharded_frame_steps(mgr, rdr, gnm, gprof, tc, comm.rank, comm.world, device=device, copy=copy, bands=bands)
    reply = None
    try:
        while True:
            req = steps.send(reply)
            if req[0] == 'exchange':
                reply = comm.exchange(req[1], req[2], padded=req[3] if len(req) > 3 else None)
            elif req[0] == 'gather':
                reply = comm.gather_rows(req[1])
            else:
                reply = comm.sum(req[1])
    except StopIteration as done:
        return done.value
