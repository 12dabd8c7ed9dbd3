"""
Output modules: pixel-format conversion on device, copy to host, encode.

Role of cuburn/output.py:21-136,411-434: 8-bit RGBA (jpeg / png / raw) and 16-bit RGBA (tiff /
raw16) stills here; the outputs that pipe frames into a video encoder (x264, vpxenc, ffmpeg) are
in encoders.py.
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

    def shape(self, dim):
        """Shape of the host frame ``copy`` returns."""
        return (dim.h, dim.w, 4)

    def copy(self, fb, dim, pool=None, stream=None, dev_out=0, host=True):
        """Queue dither+convert and the async D2H; returns the host array (output.py:85-88).
        ``dev_out``: device address that receives the converted frame instead of the context's own
        pixel buffer (e.g. a tensor that RCCL gathers: no host round trip); ``host=False`` skips the
        D2H copy altogether (returns None)."""
        h_out = fb.host_buffer(self.shape(dim), self.dtype) if host else None
        _lib.check(_lib.load().fl_output(fb.ctx, dim.w, dim.h, self.fmt, h_out.ctypes.data if host else None, int(dev_out)))
        return h_out

    def encode(self, buf):
        if buf is None:
            return {}, []
        return {self.suffix: io.BytesIO(np.ascontiguousarray(buf).tobytes())}, []

    def abort(self):
        """Drop whatever a half-written output holds (encoder processes, temporary files)."""


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


def _tiff_bytes(buf):
    """Baseline little-endian TIFF, uncompressed, one strip; 8- or 16-bit, 1/3/4 samples per pixel."""
    buf = np.ascontiguousarray(buf)
    if buf.ndim == 2:
        buf = buf[:, :, None]
    h, w, ch = buf.shape
    bits = buf.dtype.itemsize * 8
    data = buf.astype('<u%d' % buf.dtype.itemsize).tobytes()
    entries = []        # (tag, type, count, value or bytes)
    extra = b''
    def add(tag, typ, values):
        nonlocal extra
        fmt = {3: 'H', 4: 'I'}[typ]
        raw = struct.pack('<%d%s' % (len(values), fmt), *values)
        if len(raw) <= 4:
            entries.append((tag, typ, len(values), raw.ljust(4, b'\0')))
        else:
            entries.append((tag, typ, len(values), None))
            extra += raw
            entries[-1] = (tag, typ, len(values), ('off', len(extra) - len(raw)))
    n_ifd = 10 + (1 if ch in (2, 4) else 0)
    ifd_off = 8
    extra_off = ifd_off + 2 + 12 * n_ifd + 4
    add(256, 4, [w]); add(257, 4, [h]); add(258, 3, [bits] * ch); add(259, 3, [1])
    add(262, 3, [2 if ch >= 3 else 1])
    add(273, 4, [0])                    # strip offset, patched below
    add(277, 3, [ch]); add(278, 4, [h]); add(279, 4, [len(data)]); add(284, 3, [1])
    if ch in (2, 4):
        add(338, 3, [2])                # ExtraSamples: unassociated alpha
    assert len(entries) == n_ifd
    data_off = extra_off + len(extra)
    out = [b'II*\0' + struct.pack('<I', ifd_off), struct.pack('<H', n_ifd)]
    for tag, typ, count, val in sorted(entries):
        if tag == 273:
            val = struct.pack('<I', data_off)
        elif isinstance(val, tuple):
            val = struct.pack('<I', extra_off + val[1])
        out.append(struct.pack('<HHI', tag, typ, count) + val)
    out.append(struct.pack('<I', 0))
    return b''.join(out) + extra + data


class PILOutput(Output):
    """8-bit stills: jpeg (Pillow) and png (Pillow, or the built-in writer without it);
    same file naming as cuburn/output.py:69-108."""

    def __init__(self, codec='jpeg', quality=100, alpha=False):
        self.type, self.quality, self.alpha = codec, quality, alpha
        if codec == 'jpeg':
            import PIL.Image      # noqa: F401  (fail at construction, like the reference)
        self.suffix = get_suffix(codec, alpha)

    def _save(self, arr):
        out = io.BytesIO()
        arr = np.ascontiguousarray(arr)
        try:
            import PIL.Image
            PIL.Image.fromarray(arr).save(out, self.type, quality=self.quality)
        except ImportError:
            if self.type != 'png':
                raise
            out.write(_png_bytes(arr if arr.ndim == 3 else np.repeat(arr[:, :, None], 3, 2)))
        out.seek(0)
        return out

    def encode(self, buf):
        if buf is None:
            return {}, []
        if self.type == 'jpeg':
            if self.alpha:
                return {'_color.jpg': self._save(buf[:, :, :3]), '_alpha.jpg': self._save(buf[:, :, 3])}, []
            return {'.jpg': self._save(buf[:, :, :3])}, []
        return {'.' + self.type: self._save(buf)}, []


class PNGOutput(PILOutput):
    def __init__(self, alpha=False):
        PILOutput.__init__(self, 'png', alpha=alpha)

    def encode(self, buf):
        if buf is None:
            return {}, []
        img = buf if self.alpha else buf[:, :, :3]
        return {'.png': io.BytesIO(_png_bytes(np.ascontiguousarray(img)))}, []


class TiffOutput(Output):
    """16-bit stills (cuburn/output.py:110-136), written by the built-in baseline TIFF writer."""
    fmt = 1
    dtype = 'u2'
    suffix = '.tiff'

    def __init__(self, alpha=False):
        self.alpha = alpha

    def encode(self, buf):
        if buf is None:
            return {}, []
        return {'.tiff': io.BytesIO(_tiff_bytes(buf if self.alpha else buf[:, :, :3]))}, []


class RawOutput(Output):
    suffix = '.rgba8'


class Raw16Output(Output):
    fmt = 1
    dtype = 'u2'
    suffix = '.rgba16'


_VIDEO = ('x264', 'vp8', 'vp9', 'prores')


def get_suffix(codec, alpha=False):
    ext = dict(jpeg='.jpg', png='.png', tiff='.tiff', raw='.rgba8', raw16='.rgba16', x264='.h264',
               prores='.mov', vp8='.webm', vp9='.webm')[codec]
    return ('_color' + ext) if alpha else ext


def get_output_for_profile(gprof):
    """Output module for the profile's ``output`` block (cuburn/output.py:421-436)."""
    opts = dict(gprof.output.raw())
    handler = opts.pop('type', 'jpeg')
    if handler in ('jpeg', 'png'):
        return PILOutput(codec=handler, **opts)
    if handler == 'tiff':
        return TiffOutput(**opts)
    if handler == 'raw':
        return RawOutput()
    if handler == 'raw16':
        return Raw16Output()
    if handler in _VIDEO:
        from . import encoders
        if handler == 'x264':
            return encoders.X264Output(**opts)
        if handler == 'prores':
            return encoders.ProResOutput(fps=gprof.fps, **opts)
        return encoders.VPxOutput(codec=handler, fps=gprof.fps, **opts)
    raise ValueError('Invalid output type "%s".' % handler)


def get_suffix_for_profile(gprof):
    opts = dict(gprof.output.raw())
    return get_suffix(opts.get('type', 'jpeg'), bool(opts.get('alpha')))
