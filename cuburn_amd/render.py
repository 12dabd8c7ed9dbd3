"""
Resources and tools to perform rendering — the drop-in for cuburn.render.

Same entry points as cuburn/render.py:23-434: ``Dimensions``, ``Framebuffers.calc_dim``,
``Renderer(gnm, gprof)``, ``RenderManager().queue_frame(rdr, gnm, gprof, tc)`` returning
``(evt, h_out)`` with ``evt.synchronize() / .query() / .time()``.  All device work happens
in libflame_hip.so through the C ABI of include/flame_hip.h.
"""
import ctypes as C
import os
import weakref
from collections import namedtuple

import numpy as np

from . import _lib, filters, output, mwc
from .packer import GenomePacker
from .genome.util import palette_decode

RenderedImage = namedtuple('RenderedImage', 'buf idx gpu_time')
Dimensions = namedtuple('Dimensions', 'w h aw ah astride')


class DurationEvent(object):
    """
    Completion handle of one queued frame (cuburn/render.py:26-38).

    The handle refers to a frame of the native context that ``fb`` owned when the frame was
    queued.  Framebuffers re-creates that context when the image size asks for the other walker
    geometry: it first resolves every outstanding handle (``_finalise``), so a handle never
    touches a context that has been destroyed.
    """

    def __init__(self, fb, frame_id):
        self._fb, self._id = fb, frame_id
        self._gen = fb.generation
        self._ms = None
        fb._track(self)

    def _call(self, fn, *args):
        if self._fb is None or self._gen != self._fb.generation or self._fb._ctx is None:
            raise _lib.FlameError('frame %d belongs to a render context that no longer exists' % self._id)
        return fn(self._fb._ctx, self._id, *args)

    def _finalise(self):
        """Resolve the frame time while the context still exists (called before it is dropped)."""
        if self._ms is None:
            try:
                self.synchronize()
            except Exception:
                self._ms = float('nan')             # e.g. more than 4 frames old: no longer tracked
        self._fb = None

    def synchronize(self):
        if self._ms is None:
            ms = C.c_float()
            _lib.check(self._call(_lib.load().fl_frame_ms, C.byref(ms)))
            self._ms = ms.value
        return self

    def query(self):
        if self._ms is not None:
            return True
        rc = self._call(_lib.load().fl_frame_query)
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def time(self):
        """Milliseconds from the start of the frame to the end of its D2H copy."""
        return self.synchronize()._ms


class Framebuffers(object):
    """
    The accumulation / filter buffers and the stream that serialises their use
    (cuburn/render.py:40-170).  Device memory is owned by the native context.
    """
    gutter = 12

    @classmethod
    def calc_dim(cls, width, height):
        """Padded dimensions: gutter 12, (awidth % 32) == 0 after striding, (aheight % 16) == 0."""
        awidth = width + 2 * cls.gutter
        aheight = 16 * int(np.ceil((height + 2 * cls.gutter) / 16.))
        astride = 32 * int(np.ceil(awidth / 32.))
        return Dimensions(width, height, awidth, aheight, astride)

    # Walker geometry.  Small images: 1536 slots of 4 waves (six 26 KB workgroups per CU).  From
    # about 1440p up (more than 1024 tiles of 128x64) 1024 slots of 8 waves: batches of 8192
    # samples keep the runs per tile long enough for the accumulate (cfg4 4K: 5.61 -> 5.49 ms,
    # cfg5 8K: 97 -> 79 ms per frame; at 1080p the 4-wave geometry is 3 % faster).  Explicit
    # nslots / FLAME_NW pin the geometry.
    NARROW, WIDE, HUGE = (4, 1536), (8, 1024), (16, 1024)
    # Small images with few samples per frame: 1024 slots (the reference's 1024 ring entries, util.py:343).  Every
    # walker spends the reference's 256 un-plotted rounds per frame (render.py:214), so fewer walkers are less work:
    # with up to 2^28 samples (one launch of <= 1024 rounds) the pipelined frame loop is 2.4-3.6 % faster (cfg2 1.44 ->
    # 1.39 ms) although the iterate kernel ALONE is 12 % slower at four instead of six waves per SIMD — the other
    # stream lane's kernels run beside it.  With more samples the second launch costs more than the fuse saves
    # (profiles/r03_slots_by_samples.txt).  Decided per frame from its sample count (set_dim).
    NARROW_FEW = (4, 1024)
    # ... and 1280 slots for frames of 2^28 .. 2^30 samples (round 6): six iterate workgroups per CU hold 155 KB of its 160 KB of
    # LDS and leave the other stream lane's kernels no room beside them, five leave 31 KB (profiles/r06_experiments.txt section 17:
    # cfg3 at 2^29 / 2^30 samples 2.66 -> 2.51 / 4.56 -> 4.39 ms per frame, cfg2's flame 2.35 -> 2.26 / 4.02 -> 3.89; from 2^31
    # samples on the walk itself decides and 1536 slots win by 0.4-1.7 %)
    NARROW_MID = (4, 1280)
    MID_SAMPLES = 2 ** 30
    # The same for the 8-wave geometry (round 5; since round 6 only through FLAME_NW=8 FLAME_NSLOTS=512): 512 slots whose two halves of four waves walk two temporal samples — the
    # reference's 1024 samples x 256 threads exactly, bit for bit the walkers of NARROW_FEW, sharing 8192-record sort batches
    # (csrc/iter.hip "Sub-blocks of four waves").  1024 slots of 8 waves walk 512 threads per sample: twice the un-plotted rounds.
    WIDE_FEW = (8, 512)
    HUGE_FEW = (16, 256)        # the same above 4K: four temporal samples per 16-wave workgroup (8K frames of 2^28 samples: iterate -19 %,
                                # frame -4.4 %; with 2^32 samples the frame gains 1.8 % but the iterate kernel alone loses 10 %: not taken there)
    FEW_SAMPLES = 2 ** 28
    WIDE_FROM_TILES = 1024
    HUGE_FROM_TILES = 2047      # above 4K (where the accumulate switches to 256x64 tiles): 16-wave workgroups,
                                # batches of 16384 samples (cfg5 8K: 59.1 -> 52.2 ms per frame)

    def __init__(self, device=0, nslots=None, host_seed=None, stream=None):
        self.device, self.host_seed, self.stream = device, host_seed, stream
        env_nw = os.environ.get('FLAME_NW')
        self._auto = nslots is None and env_nw is None
        self._cfg = (int(env_nw) if env_nw in ('8', '16') else 4, nslots if nslots is not None else self.NARROW[1])
        self._ctx = None
        self.generation = 0                 # bumped whenever the native context is re-created
        self.nout = 65536                   # RNG states of the output dither kernel
        self._host = {}
        self._pinned_ptrs = []
        self._events = []                   # weak references to unresolved DurationEvents of this context
        self.ctx                            # create now: no GPU / no library must fail here, loudly

    nw = property(lambda self: self._cfg[0])             # waves per iterate workgroup
    nslots = property(lambda self: self._cfg[1])
    nthreads = property(lambda self: self._cfg[0] * 64)
    ntemporal = property(lambda self: max(self.nslots, 1024))     # temporal samples per frame (512 x 8 / 256 x 16 waves: a sample per four waves)
    nwalkers = property(lambda self: self.nslots * self.nthreads + 64 * 256 + self.nout)

    @property
    def ctx(self):
        """The native context (re-created by set_dim when the image size asks for the other geometry)."""
        if self._ctx is None:
            lib = _lib.load()
            # The reference keeps its seed table for the manager's lifetime (cuburn/render.py:95-104).  Here a geometry switch
            # re-creates the context and with it the RNG states: from the second context on the generation count is mixed into
            # the host seed, so that an animation whose samples per frame cross 2^28 does not replay the same streams after
            # every switch.  (As in the reference, a frame's noise depends on what this manager rendered before it — the RNG
            # states persist across frames — and, after a switch, on how many switches there were; only the first context's
            # streams are a function of the host seed alone.)
            seed = self.host_seed
            if seed and self.generation:
                seed = (seed + 0x9E3779B1 * self.generation) & 0x7fffffff or 1
            seeds = np.ascontiguousarray(mwc.make_seeds(self.nwalkers, seed))
            ctx = C.c_void_p()
            _lib.check(lib.fl_ctx_create(self.device, self.stream, seeds.ctypes.data, self.nwalkers,
                                         self.nslots, C.byref(ctx)))
            self._ctx = ctx
        return self._ctx

    def _pinned(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = _lib.load().fl_host_alloc(n)
        if not p:
            raise MemoryError('pinned host allocation of %d bytes failed' % n)
        self._pinned_ptrs.append(p)
        buf = (C.c_ubyte * n).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def host_buffer(self, shape, dtype):
        """Page-locked output buffers, rotated so that up to five frames may be in flight."""
        key = (tuple(shape), dtype)
        if key not in self._host:
            self._host[key] = [self._pinned(shape, dtype) for _ in range(5)]
        ring = self._host[key]
        ring.append(ring.pop(0))
        return ring[-1]

    def set_dim(self, width, height, stream=None, nsamples=None):
        dim = self.calc_dim(width, height)
        if self._auto:
            ntiles = ((dim.astride + 127) // 128) * ((dim.ah + 63) // 64)
            want = self.HUGE if ntiles > self.HUGE_FROM_TILES else self.WIDE if ntiles > self.WIDE_FROM_TILES else self.NARROW
            if want == self.NARROW:
                # decided per FRAME from its sample count, so that the GEOMETRY a frame is rendered with does not depend on
                # what was rendered before it, on whichever rank (a context that last held the other geometry is re-created:
                # 1024 <-> 1280 <-> 1536 slots only changes when a job's samples per frame cross 2^28 / 2^30; the re-created context's RNG
                # streams do depend on the number of switches, see `ctx`); without a sample count the small-image geometry
                # in use is kept
                if nsamples is not None:
                    want = self.NARROW_FEW if nsamples <= self.FEW_SAMPLES else self.NARROW_MID if nsamples <= self.MID_SAMPLES else self.NARROW
                elif self._cfg in (self.NARROW, self.NARROW_MID, self.NARROW_FEW):
                    want = self._cfg
            elif want == self.WIDE:
                if nsamples is not None:
                    # (round 6: frames of few samples take the 16-wave quarters from ~1440p up already — batches of 16384 records
                    # instead of 8192 give the 4K accumulate runs of 15 records: cfg4 2.94-2.99 -> 2.77-2.83 ms per frame,
                    # profiles/r06_experiments.txt section 10; the paired 8-wave geometry stays available to an explicit FLAME_NW=8)
                    want = self.HUGE_FEW if nsamples <= self.FEW_SAMPLES else self.WIDE
                elif self._cfg in (self.WIDE, self.WIDE_FEW, self.HUGE_FEW):
                    want = self._cfg
            elif want == self.HUGE:
                if nsamples is not None:
                    want = self.HUGE_FEW if nsamples <= self.FEW_SAMPLES else self.HUGE
                elif self._cfg in (self.HUGE, self.HUGE_FEW):
                    want = self._cfg
            if want != self._cfg:
                self._drop_ctx()
                self._cfg = want
        return dim

    def _track(self, evt):
        self._events = [r for r in self._events if r() is not None and r()._ms is None][-8:]
        self._events.append(weakref.ref(evt))

    def _sync_torch(self):
        """The sample-sharded path queues its last steps (all-gather, copy into the pinned frame buffer) on torch's stream:
        before pinned memory or the context go away, that stream has to be through with them as well."""
        dev = getattr(self, '_torch_stream_used', None)
        if dev is not None:
            import torch
            torch.cuda.current_stream(dev).synchronize()

    def _drop_ctx(self):
        if self._ctx is not None:
            _lib.load().fl_ctx_sync(self._ctx)
            self._sync_torch()
            for r in self._events:                      # resolve handles that still point into this context
                evt = r()
                if evt is not None:
                    evt._finalise()
            self._events = []
            _lib.load().fl_ctx_destroy(self._ctx)
            self._ctx = None
            self.generation += 1

    def read(self, which, shape, dtype, genome=None):
        """Debug tap: copy a device buffer to the host."""
        arr = np.empty(shape, dtype)
        _lib.check(_lib.load().fl_read_buffer(self.ctx, genome, _lib.BUF[which], arr.ctypes.data, arr.nbytes))
        return arr

    def write(self, which, arr, genome=None):
        arr = np.ascontiguousarray(arr)
        _lib.check(_lib.load().fl_write_buffer(self.ctx, genome, _lib.BUF[which], arr.ctypes.data, arr.nbytes))

    def free(self):
        if self._ctx is not None:
            _lib.load().fl_ctx_sync(self._ctx)
        self._sync_torch()
        self._host.clear()
        for p in self._pinned_ptrs:
            _lib.load().fl_host_free(p)
        self._pinned_ptrs = []
        self._drop_ctx()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Renderer(object):
    """
    A genome prepared for rendering: packer layout + device program + filters + output
    (cuburn/render.py:225-251).  ``compile`` here builds the xform program instead of CUDA.
    """

    @classmethod
    def compile(cls, gnm, arch=None, keep=False):
        packer = GenomePacker(gnm)
        return packer, packer.prog, packer.ops_array

    def __init__(self, gnm, gprof, keep=False, arch=None):
        self.packer, self.lib, self.cubin = self.compile(gnm, keep=keep, arch=arch)
        self.mod = None          # device handle, created on first use by a RenderManager
        self._mod_key = None
        self.filts = filters.create(gprof)
        self.out = output.get_output_for_profile(gprof)

    def _handle(self, fb):
        key = (id(fb), fb.generation)
        if self.mod is not None and self._mod_key != key:      # the context it belonged to is gone
            _lib.load().fl_genome_destroy(self.mod)
            self.mod = None
        if self.mod is None:
            self._mod_key = key
            g = C.c_void_p()
            prog = np.ascontiguousarray(self.packer.prog)
            ops = np.ascontiguousarray(self.packer.ops_array)
            _lib.check(_lib.load().fl_genome_create(fb.ctx, prog.ctypes.data, len(prog), ops.ctypes.data,
                                                    len(ops), self.packer.nrows, C.byref(g)))
            self.mod = g
        return self.mod

    def __del__(self):
        try:
            if self.mod:
                _lib.load().fl_genome_destroy(self.mod)
        except Exception:
            pass


class RenderManager(object):
    """Frame queue (cuburn/render.py:253-434)."""

    # 'auto': binned accumulate (sample log + LDS tiles: 128x64-pixel tiles up to 4K, 256x64 above)
    # unless the image has more than 8191 tiles, else direct packed global atomics.  Both give the
    # same histogram.
    accum_mode = 'auto'
    # Write-disabled iterations per walker at the start of a frame: the reference's 256 (render.py:215,
    # iter.py:209-216,298-300: every point set is re-seeded per frame and spends one whole 256-round block
    # un-plotted).  The BASELINE flames are converged after 16 (DESIGN.md §5 'Fuse'), but a genome of ONE
    # slowly contracting xform still shows its start-up transient at 64 — so the reference's schedule is
    # the default, and a shorter fuse (attribute or FLAME_FUSE) is the caller's decision.
    fuse = int(os.environ.get('FLAME_FUSE', 256))

    def __init__(self, device=None, nslots=None, host_seed=None, stream=None):
        if device is None:
            device = int(os.environ.get('LOCAL_RANK', 0)) if 'LOCAL_RANK' in os.environ else 0
        self.fb = Framebuffers(device, nslots, host_seed, stream)
        self.last_nsamples = 0

    def _copy(self, rdr, gnm):
        """Upload packed splines and palettes (cuburn/render.py:264-285).  Skipped when the device
        copy of this Renderer's genome handle already holds exactly these values."""
        g = rdr._handle(self.fb)
        sig = (rdr.packer.signature(gnm), tuple(tuple(v) for v in gnm['palette']))
        if getattr(rdr, '_uploaded', None) == (rdr._mod_key, sig):
            return
        times, knots = rdr.packer.pack(gnm, sig=sig[0])
        palsrc = dict([(v[0], palette_decode(v[1:])) for v in gnm['palette']])
        ptimes, pvals = zip(*sorted(palsrc.items()))
        palettes = np.ascontiguousarray(np.array(pvals, dtype=np.float32))
        palette_times = np.full(32, 1e9, dtype=np.float32)
        palette_times[:len(ptimes)] = ptimes
        _lib.check(_lib.load().fl_genome_upload(self.fb.ctx, g, times.ctypes.data,
                                                knots.ctypes.data, palettes.ctypes.data,
                                                palette_times.ctypes.data, len(ptimes)))
        rdr._uploaded = (rdr._mod_key, sig)

    def resolve_accum_mode(self, dim):
        mode = self.accum_mode
        if mode == 'auto':
            # the library bins up to 8191 tiles of 256x64 pixels (128x64 up to 2047 tiles, i.e. 4K)
            ntiles = ((dim.astride + 255) // 256) * ((dim.ah + 63) // 64)
            mode = _lib.ACCUM_BINNED if ntiles <= 8191 else _lib.ACCUM_ATOMIC
        return mode

    def queue_frame(self, rdr, gnm, gprof, tc, copy=True, dev_out=0, host=True):
        """
        Queue one frame at centre time ``tc``; returns ``(evt, h_out)`` (render.py:374-434).
        ``dev_out`` / ``host``: see Output.copy (frame straight into a device buffer of the caller).
        """
        lib = _lib.load()
        dim = self.fb.set_dim(gprof.width, gprof.height, nsamples=gprof.spp(tc) * gprof.width * gprof.height)
        td = gprof.frame_width(tc) / round(gprof.fps * gprof.duration)
        ts = tc - 0.5 * td
        g = rdr._handle(self.fb)
        fid = C.c_uint32()
        _lib.check(lib.fl_frame_begin(self.fb.ctx, C.byref(fid)))
        if copy:
            self._copy(rdr, gnm)
        _lib.check(lib.fl_interp(self.fb.ctx, g, dim.w, dim.h, ts, td))
        nsamps = gprof.spp(tc) * dim.w * dim.h
        run = C.c_uint64()
        _lib.check(lib.fl_iterate(self.fb.ctx, g, dim.w, dim.h, float(nsamps), self.fuse,
                                  self.resolve_accum_mode(dim), C.byref(run)))
        self.last_nsamples = run.value
        for filt in rdr.filts:
            params = getattr(gprof.filters, filt.name)
            filt.apply(self.fb, gprof, params, dim, tc)
        rdr.out.convert(self.fb, gprof, dim)
        h_out = rdr.out.copy(self.fb, dim, dev_out=dev_out, host=host)
        return DurationEvent(self.fb, fid.value), h_out

    def timings_reset(self):
        _lib.check(_lib.load().fl_timings_reset(self.fb.ctx))

    def timings(self):
        """HIP-event times (ms) of the iterate, drain and filter kernels since timings_reset()."""
        it, fl, ft, n = C.c_float(), C.c_float(), C.c_float(), C.c_uint32()
        _lib.check(_lib.load().fl_timings(self.fb.ctx, C.byref(it), C.byref(fl), C.byref(ft), C.byref(n)))
        d = (C.c_float * 6)()
        _lib.check(_lib.load().fl_timings_detail(self.fb.ctx, C.byref(d)))
        st = (C.c_uint32 * 4)()
        _lib.check(_lib.load().fl_launch_stats(self.fb.ctx, C.byref(st)))
        return dict(iter_ms=it.value, flush_ms=fl.value, filter_ms=ft.value, launches=n.value,
                    accum_ms=d[1], flush_only_ms=d[2], de_ms=d[4], de_finish_ms=d[5],
                    spec_launches=st[0], interp_launches=st[1])
