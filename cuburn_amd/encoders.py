"""
Video outputs: the device converts a frame to the pixel format an external encoder takes, the host
feeds the encoder's stdin and hands back the finished segment when the stream is flushed.

Role of cuburn/output.py:139-409 (ProResOutput, X264Output, VPxOutput): same constructor options,
same encoder command lines, same ``encode(buf) -> (media, logs)`` protocol (``encode(None)``
flushes; a change of frame size flushes the x264 stream and starts a new one).  The encoders are
separate programs (x264, vpxenc, ffmpeg) found on PATH at the first frame — ``command=`` replaces
the program name, which is also how the tests drive the classes without them.
"""
import subprocess
import tempfile

import numpy as np

from . import _lib
from .output import Output


class _EncoderPipe(object):
    """One running encoder: frames go to its stdin; its stdout (or a named file it writes itself)
    is the media segment, its stderr the log."""

    def __init__(self, argv, what, named_suffix=None):
        self.what = what
        self.named = tempfile.NamedTemporaryFile(suffix=named_suffix) if named_suffix else None
        self.outf = None if named_suffix else tempfile.TemporaryFile()
        argv = [str(a) for a in argv] + ([self.named.name] if self.named else [])
        self.errf = tempfile.TemporaryFile()        # a file, not a pipe: a chatty encoder can never block on it
        try:
            self.proc = subprocess.Popen(argv, stdin=subprocess.PIPE, stdout=self.outf if self.outf else subprocess.DEVNULL,
                                         stderr=self.errf)
        except OSError as e:
            raise IOError('cannot start %s (%s): %s' % (what, argv[0], e))

    def _log(self):
        self.errf.seek(0)
        return self.errf.read().decode('utf-8', 'replace')

    def write(self, buf):
        try:
            self.proc.stdin.write(memoryview(np.ascontiguousarray(buf)).cast('B'))
        except (IOError, OSError) as e:
            raise IOError('%s stopped reading frames: %s\n%s' % (self.what, e, self._log()))

    def abort(self):
        """Kill the encoder and drop its temporary files (a failed job must leave nothing running)."""
        for step in (lambda: self.proc.stdin.close(), self.proc.kill, self.proc.wait, self.errf.close,
                     lambda: self.outf and self.outf.close(), lambda: self.named and self.named.close()):
            try:
                step()
            except Exception:
                pass

    def finish(self):
        """Close stdin, wait for the encoder; returns (media file positioned at 0, log text)."""
        try:
            self.proc.stdin.close()
        except (IOError, OSError):
            pass
        self.proc.wait()
        log = self._log()
        self.errf.close()
        if self.proc.returncode:
            raise IOError('%s exited with an error\n%s' % (self.what, log))
        if self.named:
            media = open(self.named.name, 'rb')     # keep a handle, let the name go (output.py:176-179)
            self.named.close()
        else:
            media = self.outf
            media.seek(0)
        return media, log


def _abort_pipes(obj, names):
    for n in names:
        pipe = getattr(obj, n, None)
        if pipe is not None:
            pipe.abort()
            setattr(obj, n, None)


class _PlanarOutput(Output):
    """Outputs whose device format is planar: ``copy`` shapes per cuburn/output.py:323-338."""
    pix_fmt = 'yuv444p'

    def abort(self):
        _abort_pipes(self, ('_pipe',))

    def shape(self, dim):
        if self.fmt == _lib.OUT['yuv420p10']:
            return (dim.h * dim.w * 6 // 4,)
        return (3, dim.h, dim.w)


class ProResOutput(_PlanarOutput):
    """12-bit 4:4:4 studio-swing frames into ``ffmpeg -c:v prores`` (cuburn/output.py:139-187)."""
    fmt = _lib.OUT['yuv444p12']
    dtype = 'u2'

    def __init__(self, fps=24, command='ffmpeg'):
        self.fps, self.command = fps, command
        self._pipe = None
        self._dim = None

    def convert(self, fb, gprof, dim, stream=None):
        self._dim = (dim.w, dim.h)

    def copy(self, fb, dim, *args, **kwargs):
        self._dim = (dim.w, dim.h)
        return super(ProResOutput, self).copy(fb, dim, *args, **kwargs)

    def encode(self, buf):
        if buf is None:
            if self._pipe is None:
                return {}, []
            media, _ = self._pipe.finish()
            self._pipe = None
            return {'.mov': media}, []
        if self._pipe is None:
            w, h = self._dim if self._dim else (buf.shape[2], buf.shape[1])
            argv = [self.command] + ('-loglevel panic -f rawvideo -pix_fmt yuv444p12le -s %dx%d -r %s -i - '
                                     '-c:v prores -f mov -y' % (w, h, self.fps)).split()
            self._pipe = _EncoderPipe(argv, 'ffmpeg', named_suffix='.mov')
        self._pipe.write(buf)
        return {}, []


class X264Output(Output):
    """16-bit RGB frames into x264 (high 4:4:4 by default); with ``alpha`` a second encoder receives
    the alpha plane as the luma of a 4:2:0 stream with neutral chroma (cuburn/output.py:190-293)."""
    fmt = _lib.OUT['rgba16']
    dtype = 'u2'

    profiles = {'normal': '--profile high444 --level 4.2', '': ''}
    base = '--no-progress --input-depth 16 --sync-lookahead 0 --rc-lookahead 5 --muxer raw -o - - --log-level debug'

    def __init__(self, profile='normal', csp='i444', crf=15, command='x264', x264opts='', alpha=False):
        self.args = ' '.join([command, self.base, self.profiles[profile], '--crf', str(crf), x264opts]).split()
        self.alpha, self.csp = alpha, csp
        self.framesize = None
        self._color = self._alpha = None
        self._neutral = None

    def _start(self, framesize, alpha):
        extras = ['--input-csp', 'yv12' if alpha else 'rgb', '--demuxer', 'raw', '--input-res', '%dx%d' % (framesize[1], framesize[0])]
        extras += ['--output-csp', 'i420', '--chroma-qp-offset', '24'] if alpha else ['--output-csp', self.csp]
        return _EncoderPipe(self.args + extras, 'x264')

    def abort(self):
        _abort_pipes(self, ('_color', '_alpha'))

    def _flush(self):
        if self._color is None:
            return {}, []
        media, log = self._color.finish()
        self._color = None
        if not self.alpha:
            return {'.h264': media}, [('x264_color', log)]
        amedia, alog = self._alpha.finish()
        self._alpha = None
        return {'_color.h264': media, '_alpha.h264': amedia}, [('x264_color', log), ('x264_alpha', alog)]

    def encode(self, buf):
        out = ({}, [])
        if buf is None or self.framesize != tuple(buf.shape[:2]):
            out = self._flush()
        if buf is None:
            return out
        if self._color is None:
            self.framesize = tuple(buf.shape[:2])
            self._color = self._start(self.framesize, False)
            if self.alpha:
                try:
                    self._alpha = self._start(self.framesize, True)
                except IOError:                     # do not leave the colour encoder running without its twin
                    self._color.abort(); self._color = None
                    raise
                self._neutral = np.full(self.framesize[0] * self.framesize[1] // 2, 32767, dtype='u2')   # both chroma planes
        try:
            self._color.write(buf[:, :, :3])
            if self.alpha:
                self._alpha.write(buf[:, :, 3])
                self._alpha.write(self._neutral)
        except IOError:
            self.abort()                            # neither encoder survives a failed write
            raise
        return out


class VPxOutput(_PlanarOutput):
    """Planar YUV frames into vpxenc (cuburn/output.py:295-409).  ``pix_fmt``: yuv420p (8-bit 4:4:4
    from the device, chroma decimated on the host as the reference does), and for vp9 yuv444p,
    yuv420p10, yuv444p10, yuv444p12."""

    base = 'vpxenc --end-usage=3 -p 1 -q --cpu-used=-8 --lag-in-frames=5 --min-q=2 --disable-kf --arnr-maxframes=3 -o - -'
    _formats = {                  # pix_fmt: (device format, dtype, extra arguments)
        'yuv420p': ('yuv444p', 'u1', []),
        'yuv444p': ('yuv444p', 'u1', ['--profile=1', '--i444']),
        'yuv420p10': ('yuv420p10', 'u2', ['-b', '10', '--input-bit-depth=10', '--profile=2']),
        'yuv444p10': ('yuv444p10', 'u2', ['-b', '10', '--input-bit-depth=10', '--profile=3', '--i444']),
        'yuv444p12': ('yuv444p12', 'u2', ['-b', '12', '--input-bit-depth=12', '--profile=3', '--i444']),
    }

    def __init__(self, codec='vp9', fps=24, crf=15, pix_fmt='yuv420p', command=None):
        if pix_fmt not in self._formats:
            raise ValueError('Invalid pix_fmt: ' + pix_fmt)
        if pix_fmt != 'yuv420p' and codec != 'vp9':
            raise ValueError('%s needs codec vp9' % pix_fmt)
        self.codec, self.pix_fmt = codec, pix_fmt
        dev, self.dtype, extra = self._formats[pix_fmt]
        self.fmt = _lib.OUT[dev]
        self.args = self.base.split() + extra + ['--codec=' + codec, '--cq-level=' + str(crf), '--fps=%d/1' % fps]
        if command:
            self.args[0] = command
        if codec == 'vp9':
            self.args += ['-t', '4']
        self._pipe = None
        self._dim = None

    def convert(self, fb, gprof, dim, stream=None):
        self._dim = (dim.w, dim.h)

    def copy(self, fb, dim, *args, **kwargs):
        self._dim = (dim.w, dim.h)
        return super(VPxOutput, self).copy(fb, dim, *args, **kwargs)

    def encode(self, buf):
        if buf is None:
            if self._pipe is None:
                return {}, []
            media, log = self._pipe.finish()
            self._pipe = None
            return {'.webm': media}, [('webm', log)]
        if self._pipe is None:
            w, h = self._dim if self._dim else (buf.shape[2], buf.shape[1])
            extras = ['-w', w, '-h', h]
            columns = int(max(0, min(3, np.log2(w) - 8.9)))
            if columns:
                extras.append('--tile-columns=%d' % columns)
            self._pipe = _EncoderPipe(self.args + extras, 'vpxenc')
        if self.pix_fmt == 'yuv420p':
            self._pipe.write(buf[0])
            self._pipe.write(buf[1, ::2, ::2])
            self._pipe.write(buf[2, ::2, ::2])
        else:
            self._pipe.write(buf)
        return {}, []
