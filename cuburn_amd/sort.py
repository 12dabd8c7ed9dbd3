"""
Device radix sort of 32-bit keys (role of cuburn/code/sort.py ``Sorter``, :384-504).

Same calling convention: ``Sorter(max_size).sort(dst, src, size, lo_bit)`` orders ``src`` by
``radix_bits`` bits starting at ``lo_bit`` into ``dst``; ``multisort`` chains passes from the low digit
up and returns the buffer that holds the result.  ``dst`` / ``src`` are device addresses (an int, or
anything with ``data_ptr()`` such as a torch tensor).  Unlike the reference's pass this one is
stable, so the multi-pass sort is exact (the reference warns that its own is not, sort.py:437-441).
The render path does not use it (nor does the reference's: render.py:19 imports it, nothing calls it).
"""
import ctypes as C

from . import _lib


def _addr(buf):
    return int(buf.data_ptr()) if hasattr(buf, 'data_ptr') else int(buf)


class Sorter(object):
    radix_bits = 8
    group_size = 4096           # keys per workgroup (the reference: 8192 or 4096, sort.py:586-588)

    def __init__(self, max_size, fb=None, device=0):
        """``fb``: a render.Framebuffers whose native context (device, stream) the sort runs on; without
        one a private context is created on ``device``."""
        from . import render
        self.max_size = int(max_size)
        self._own = fb is None
        self.fb = fb if fb is not None else render.Framebuffers(device=device, nslots=1024, host_seed=1)
        self.nvalid = None

    @property
    def radix_size(self):
        return 1 << self.radix_bits

    def sort(self, dst, src, size, lo_bit=0, ignore_max=False, stream=None, count=False):
        """One pass.  ``ignore_max``: keys equal to 0xffffffff are dropped; with ``count`` the number of
        keys written is read back into ``self.nvalid`` (synchronises).  Passes run on the first stream of
        the native context (pass a stream to Framebuffers to choose it): a per-call ``stream`` is refused."""
        if stream is not None:
            raise ValueError('Sorter runs on its context\'s stream; create the Framebuffers with stream=... instead')
        if not 0 < size <= self.max_size:
            raise ValueError('size %d outside (0, %d]' % (size, self.max_size))
        n = C.c_uint32()
        _lib.check(_lib.load().fl_sort_u32(self.fb.ctx, _addr(dst), _addr(src), int(size), int(lo_bit), self.radix_bits,
                                           1 if ignore_max else 0, C.byref(n) if count else None))
        self.nvalid = n.value if count else None
        return dst

    def multisort(self, scratch_a, scratch_b, src, size, lo_bit=0, rounds=1, stream=None):
        """``rounds`` passes of ``radix_bits`` bits from ``lo_bit`` up, ping-ponging between the scratch
        buffers; returns the one holding the result (``src`` may be ``scratch_b``; otherwise it is
        left untouched).  The last pass is clipped at bit 32."""
        cur, out, other = src, scratch_a, scratch_b
        for i in range(rounds):
            lo = lo_bit + i * self.radix_bits
            bits = min(self.radix_bits, 32 - lo)
            if bits <= 0:
                break
            keep, self.radix_bits = self.radix_bits, bits
            try:
                self.sort(out, cur, size, lo, stream=stream)
            finally:
                self.radix_bits = keep
            cur, out, other = out, other, out
        return cur

    def free(self):
        if self._own and self.fb is not None:
            self.fb.free()
            self.fb = None
