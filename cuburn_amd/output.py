"""
Output modules: pixel-format conversion on device, copy to host, encode.

Role of cuburn/output.py:21-136,411-434 for the formats the hot path needs: 8-bit RGBA
(png / raw) and 16-bit RGBA (tiff-class, written raw).  Video encoders (x264, VPx, ProRes
pipes) are external programs and out of scope.
"""
import io
import struct
import zlib
import numpy as np

from . import _lib


class Output(object):
    fmt = 0
    dtype = 'u1'
    suffix = '.raw'

    def convert(self, fb, gprof, dim, stream=None):
        """Conversion and copy are one call here; kept for interface parity (output.py:29-37)."""

    def copy(self, fb, dim, pool=None, stream=None):
        """Queue dither+convert and the async D2H; returns the host array (output.py:85-88)."""
        h_out = fb.host_buffer((dim.h, dim.w, 4), self.dtype)
        _lib.check(_lib.load().fl_output(fb.ctx, dim.w, dim.h, self.fmt, h_out.ctypes.data, 0))
        return h_out

    def encode(self, buf):
        if buf is None:
            return {}, []
        return {self.suffix: io.BytesIO(np.ascontiguousarray(buf).tobytes())}, []


def _png_bytes(buf):
    """Minimal PNG writer (8-bit RGB / RGBA), no external imaging library needed."""
    h, w, ch = buf.shape
    ctype = {3: 2, 4: 6}[ch]
    raw = np.empty((h, 1 + w * ch), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = buf.reshape(h, w * ch)
    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0))
            + chunk(b'IDAT', zlib.compress(raw.tobytes(), 6)) + chunk(b'IEND', b''))


class PNGOutput(Output):
    suffix = '.png'

    def __init__(self, alpha=False):
        self.alpha = alpha

    def encode(self, buf):
        if buf is None:
            return {}, []
        img = buf if self.alpha else buf[:, :, :3]
        return {'.png': io.BytesIO(_png_bytes(np.ascontiguousarray(img)))}, []


class RawOutput(Output):
    suffix = '.rgba8'


class Raw16Output(Output):
    fmt = 1
    dtype = 'u2'
    suffix = '.rgba16'


_TYPES = {'png': PNGOutput, 'jpeg': PNGOutput, 'raw': RawOutput, 'tiff': Raw16Output, 'raw16': Raw16Output}


def get_output_for_profile(gprof):
    opts = dict(gprof.output._val)
    handler = _TYPES.get(opts.pop('type', 'png'), PNGOutput)
    if handler is PNGOutput:
        return PNGOutput(alpha=bool(opts.get('alpha', False)))
    return handler()


def get_suffix_for_profile(gprof):
    return get_output_for_profile(gprof).suffix
