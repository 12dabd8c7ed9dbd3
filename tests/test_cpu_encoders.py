"""
The video outputs (cuburn_amd/encoders.py; cuburn/output.py:139-409) driven with a stand-in encoder:
a script that copies stdin to stdout (or, like ffmpeg, to the file named by its last argument) and
reports its argument list on stderr.  What is checked is what the real x264 / vpxenc / ffmpeg would
receive — the command line and the exact byte stream — and the encode() protocol: nothing is
returned per frame, the segment comes back on flush, a change of frame size restarts x264, an
encoder that fails or is missing raises IOError.
"""
import os
import stat
import sys

import numpy as np
import pytest

from cuburn_amd import encoders

FAKE = r'''#!%s
import sys
args = sys.argv[1:]
sys.stderr.write('ARGS ' + ' '.join(args) + '\n')
if '--fail' in args:
    sys.stdin.buffer.read()
    sys.exit(3)
data = sys.stdin.buffer.read()
if '-y' in args:                        # ffmpeg style: output file is the last argument
    open(args[-1], 'wb').write(data)
else:
    sys.stdout.buffer.write(data)
''' % sys.executable


@pytest.fixture
def fake(tmp_path):
    p = tmp_path / 'fake_encoder'
    p.write_text(FAKE)
    os.chmod(str(p), os.stat(str(p)).st_mode | stat.S_IXUSR)
    return str(p)


def frames_rgba16(n, h, w, seed=0):
    rs = np.random.RandomState(seed)
    return [rs.randint(0, 65536, (h, w, 4)).astype(np.uint16) for _ in range(n)]


def test_x264_stream_and_command_line(fake):
    out = encoders.X264Output(command=fake, crf=18)
    fr = frames_rgba16(3, 6, 8)
    for f in fr:
        assert out.encode(f) == ({}, [])
    media, logs = out.encode(None)
    assert list(media) == ['.h264'] and [k for k, _ in logs] == ['x264_color']
    want = b''.join(np.ascontiguousarray(f[:, :, :3]).tobytes() for f in fr)         # alpha dropped, u16 RGB
    assert media['.h264'].read() == want
    args = logs[0][1].split('ARGS ', 1)[1].split()
    for piece in ('--input-depth 16', '--profile high444', '--level 4.2', '--crf 18', '--input-csp rgb', '--demuxer raw',
                  '--input-res 8x6', '--output-csp i444', '--muxer raw', '-o - -'):
        assert piece in ' '.join(args), piece
    assert out.encode(None) == ({}, [])                                                   # nothing left to flush


def test_x264_alpha_goes_to_a_second_encoder(fake):
    out = encoders.X264Output(command=fake, alpha=True)
    fr = frames_rgba16(2, 4, 8, seed=1)
    for f in fr:
        out.encode(f)
    media, logs = out.encode(None)
    assert sorted(media) == ['_alpha.h264', '_color.h264'] and [k for k, _ in logs] == ['x264_color', 'x264_alpha']
    neutral = np.full(4 * 8 // 2, 32767, np.uint16).tobytes()                             # 4:2:0 chroma planes
    assert media['_alpha.h264'].read() == b''.join(np.ascontiguousarray(f[:, :, 3]).tobytes() + neutral for f in fr)
    assert '--input-csp yv12' in logs[1][1] and '--output-csp i420' in logs[1][1] and '--chroma-qp-offset 24' in logs[1][1]


def test_x264_restarts_when_the_frame_size_changes(fake):
    out = encoders.X264Output(command=fake, profile='')
    a, b = frames_rgba16(1, 4, 8)[0], frames_rgba16(1, 6, 4)[0]
    assert out.encode(a) == ({}, [])
    media, logs = out.encode(b)                      # flushes the 8x4 stream, starts a 4x6 one
    assert media['.h264'].read() == np.ascontiguousarray(a[:, :, :3]).tobytes() and '--input-res 8x4' in logs[0][1]
    assert '--profile' not in logs[0][1]
    media, logs = out.encode(None)
    assert media['.h264'].read() == np.ascontiguousarray(b[:, :, :3]).tobytes() and '--input-res 4x6' in logs[0][1]


def test_vpx_420_decimates_chroma_on_the_host(fake):
    out = encoders.VPxOutput(codec='vp8', fps=30, crf=12, command=fake)
    assert (out.fmt, out.dtype) == (2, 'u1')
    rs = np.random.RandomState(2)
    fr = [rs.randint(0, 256, (3, 4, 8)).astype(np.uint8) for _ in range(2)]
    for f in fr:
        assert out.encode(f) == ({}, [])
    media, logs = out.encode(None)
    want = b''.join(f[0].tobytes() + np.ascontiguousarray(f[1, ::2, ::2]).tobytes() + np.ascontiguousarray(f[2, ::2, ::2]).tobytes() for f in fr)
    assert media['.webm'].read() == want and logs[0][0] == 'webm'
    line = logs[0][1]
    for piece in ('--end-usage=3', '--lag-in-frames=5', '--codec=vp8', '--cq-level=12', '--fps=30/1', '-w 8', '-h 4'):
        assert piece in line, piece
    assert '--tile-columns' not in line and ' -t 4' not in line


@pytest.mark.parametrize('pix_fmt,fmt,dtype,flags', [
    ('yuv444p', 2, 'u1', ['--profile=1', '--i444']),
    ('yuv420p10', 4, 'u2', ['-b 10', '--input-bit-depth=10', '--profile=2']),
    ('yuv444p10', 3, 'u2', ['-b 10', '--input-bit-depth=10', '--profile=3', '--i444']),
    ('yuv444p12', 5, 'u2', ['-b 12', '--input-bit-depth=12', '--profile=3', '--i444']),
])
def test_vp9_pixel_formats(fake, pix_fmt, fmt, dtype, flags):
    out = encoders.VPxOutput(codec='vp9', pix_fmt=pix_fmt, command=fake)
    assert (out.fmt, out.dtype) == (fmt, dtype)
    out._dim = (2048, 2)                                        # as convert() / copy() would have recorded
    n = 2048 * 2 * 6 // 4 if pix_fmt == 'yuv420p10' else 3 * 2 * 2048
    buf = (np.arange(n) % 251).astype(dtype)
    out.encode(buf if pix_fmt == 'yuv420p10' else buf.reshape(3, 2, 2048))
    media, logs = out.encode(None)
    assert media['.webm'].read() == buf.tobytes()               # planar frames pass through untouched
    line = logs[0][1]
    for piece in flags + ['--codec=vp9', '-t 4', '-w 2048', '-h 2', '--tile-columns=2']:      # log2(2048) - 8.9 = 2.1
        assert piece in line, piece


def test_vpx_rejects_unknown_formats():
    with pytest.raises(ValueError):
        encoders.VPxOutput(pix_fmt='yuv422p')
    with pytest.raises(ValueError):
        encoders.VPxOutput(codec='vp8', pix_fmt='yuv444p')


def test_prores_writes_a_named_file(fake):
    out = encoders.ProResOutput(fps=25, command=fake)
    out._dim = (8, 4)
    rs = np.random.RandomState(3)
    fr = [rs.randint(256, 3841, (3, 4, 8)).astype(np.uint16) for _ in range(2)]
    for f in fr:
        assert out.encode(f) == ({}, [])
    pipe = out._pipe
    name = pipe.named.name
    media, logs = out.encode(None)
    assert list(media) == ['.mov'] and logs == []
    assert media['.mov'].read() == b''.join(f.tobytes() for f in fr)
    assert not os.path.exists(name)                             # only the open handle keeps the segment
    assert out.encode(None) == ({}, [])


def test_encoder_failure_and_missing_program(fake, tmp_path):
    out = encoders.X264Output(command=fake, x264opts='--fail')
    out.encode(frames_rgba16(1, 4, 8)[0])
    with pytest.raises(IOError) as e:
        out.encode(None)
    assert 'exited with an error' in str(e.value) and 'ARGS' in str(e.value)          # the encoder's log travels with the error
    out = encoders.X264Output(command=str(tmp_path / 'no_such_encoder'))
    with pytest.raises(IOError):
        out.encode(frames_rgba16(1, 4, 8)[0])
