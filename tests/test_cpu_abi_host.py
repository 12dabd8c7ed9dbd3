"""
CPU tests (no GPU): the C-ABI library loads and exports every symbol include/flame_hip.h
declares, host-side entry points behave, and device entry points fail loudly without a GPU.
"""
import ctypes as C
import io
import os
import re
import zlib

import numpy as np
import pytest

from common import REPO
from cuburn_amd import _lib, render, profile, configs, output, filters, mwc


def declared_symbols():
    hdr = open(os.path.join(REPO, 'include', 'flame_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(fl_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol(built):
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(_lib.EXPORTS) == syms
    assert lib.fl_abi_version() == 1


def test_no_torch_or_cxx_types_in_the_abi():
    hdr = open(os.path.join(REPO, 'include', 'flame_hip.h')).read()
    code = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)           # declarations only, comments stripped
    for banned in ('torch', 'at::', 'std::', 'hipStream_t', 'float4', 'template', 'class '):
        assert banned not in code, banned


def test_product_never_touches_the_oracle():
    """The product path must not import / link / execute anything under oracle/."""
    for root, _, files in os.walk(os.path.join(REPO, 'cuburn_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp', 'Makefile')):
                txt = open(os.path.join(root, f), errors='replace').read()
                assert 'oracle' not in txt.replace('independent of oracle/', ''), os.path.join(root, f)
                assert 'flame_ref' not in txt, os.path.join(root, f)


def test_device_entry_points_fail_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.FlameError):
        render.RenderManager(device=0)
    assert b'HIP device' in _lib.load().fl_last_error() or b'gfx950' in _lib.load().fl_last_error()


def test_missing_library_is_an_error_not_a_fallback(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libflame_hip.so')
    with pytest.raises(_lib.FlameError):
        _lib.load()


def test_renderer_host_side_objects():
    gnm, prof = configs.cfg3()
    gprof = profile.wrap(prof, gnm)
    rdr = render.Renderer(gnm, gprof)
    assert [f.name for f in rdr.filts] == ['yuv', 'bilateral', 'logscale', 'colorclip']
    assert len(rdr.packer) == rdr.packer.pstride
    assert rdr.packer.packed[0] == ('camera', 'xx') and len(rdr.packer.prog) == 8
    assert isinstance(rdr.out, output.Output)
    assert render.Dimensions(1, 2, 3, 4, 5).astride == 5
    # default profile filter order (specs.py:107,130)
    gp = profile.wrap({}, gnm)
    assert list(gp.filter_order) == ['bilateral', 'logscale', 'smearclip']
    assert gp.width == 1280 and gp.height == 720 and gp.fps == 24


def test_refscalar_scales_genome_spline():
    gnm, prof = configs.cfg3()
    gnm['camera']['spp'] = [1.0, 0, 3.0, 0]
    gprof = profile.wrap(dict(prof, spp=100), gnm)
    assert abs(gprof.spp(0.0) - 100) < 1e-9 and abs(gprof.spp(1.0) - 300) < 1e-9
    assert abs(gprof.filters.logscale.scale(0.5) - gnm['camera']['scale']) < 1e-12


def test_png_writer_roundtrip():
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (7, 5, 4)).astype(np.uint8)
    media, logs = output.PNGOutput(alpha=True).encode(img)
    data = media['.png'].read()
    assert data[:8] == b'\x89PNG\r\n\x1a\n'
    # parse IDAT back
    pos, idat = 8, b''
    while pos < len(data):
        n = int.from_bytes(data[pos:pos + 4], 'big'); tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == int.from_bytes(data[pos + 8 + n:pos + 12 + n], 'big')
        if tag == b'IDAT':
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(7, 1 + 5 * 4)
    assert (raw[:, 0] == 0).all() and np.array_equal(raw[:, 1:].reshape(7, 5, 4), img)
    assert output.PNGOutput().encode(None) == ({}, [])


def test_make_seeds_shapes_and_cycling():
    s = mwc.make_seeds(262144 + 10, host_seed=1)
    assert s.shape == (262154, 3) and s.dtype == np.uint32
    assert s[0, 0] == 0xFFFFFF4E and s[262144, 0] == s[0, 0]
    assert (s[:, 1] >= 1).all() and (s[:, 2] < 0x7fffffff).all()


def test_unknown_variation_rejected_on_host():
    from cuburn_amd.packer import GenomePacker
    with pytest.raises(ValueError):
        GenomePacker({'type': 'animation', 'xforms': {'0': {'variations': {'nonesuch': {'weight': 1}}}}})
    with pytest.raises(ValueError):
        GenomePacker({'type': 'animation', 'xforms': {}})


def output_dim(w, h):
    from cuburn_amd.render import Framebuffers
    return Framebuffers.calc_dim(w, h)


def _wrap_output(otype, **kw):
    gnm, prof = configs.cfg1()
    return profile.wrap(dict(prof, output=dict(type=otype, **kw)), gnm)


def test_output_modules_for_profile_types():
    """Module / suffix selection follows cuburn/output.py:411-436."""
    assert output.get_suffix_for_profile(_wrap_output('jpeg')) == '.jpg'
    assert output.get_suffix_for_profile(_wrap_output('jpeg', alpha=True)) == '_color.jpg'
    assert output.get_suffix_for_profile(_wrap_output('png')) == '.png'
    assert output.get_suffix_for_profile(_wrap_output('tiff')) == '.tiff'
    o = output.get_output_for_profile(_wrap_output('tiff'))
    assert isinstance(o, output.TiffOutput) and o.fmt == 1 and o.dtype == 'u2'
    o = output.get_output_for_profile(_wrap_output('jpeg', quality=90))
    assert isinstance(o, output.PILOutput) and o.fmt == 0 and o.quality == 90
    from cuburn_amd import encoders
    o = output.get_output_for_profile(_wrap_output('x264', crf=20))
    assert isinstance(o, encoders.X264Output) and o.fmt == 1 and o.dtype == 'u2' and '20' in o.args
    o = output.get_output_for_profile(_wrap_output('vp8'))
    assert isinstance(o, encoders.VPxOutput) and o.fmt == 2 and o.dtype == 'u1' and '--codec=vp8' in o.args
    o = output.get_output_for_profile(_wrap_output('vp9', pix_fmt='yuv420p10'))
    assert o.fmt == 4 and o.dtype == 'u2' and o.shape(output_dim(64, 32)) == (64 * 32 * 6 // 4,)
    o = output.get_output_for_profile(_wrap_output('prores'))
    assert isinstance(o, encoders.ProResOutput) and o.fmt == 5 and o.shape(output_dim(64, 32)) == (3, 32, 64)
    assert output.get_suffix_for_profile(_wrap_output('vp9')) == '.webm'
    assert output.get_suffix_for_profile(_wrap_output('x264', alpha=True)) == '_color.h264'
    with pytest.raises(ValueError):
        output.get_output_for_profile(_wrap_output('vp8', pix_fmt='yuv444p10'))      # high bit depth is vp9 only
    with pytest.raises(ValueError):
        output.get_output_for_profile(_wrap_output('bogus'))


def test_jpeg_png_tiff_encode_decode():
    PIL = pytest.importorskip('PIL.Image')
    rs = np.random.RandomState(1)
    yy, xx = np.mgrid[0:24, 0:32]
    img = np.stack([xx * 8, yy * 10, (xx + yy) * 4, 255 - xx * 7], -1).astype(np.uint8)
    media, logs = output.PILOutput('png', alpha=True).encode(img)
    assert list(media) == ['.png'] and np.array_equal(np.array(PIL.open(media['.png'])), img)
    media, _ = output.PILOutput('jpeg', quality=100).encode(img)
    dec = np.array(PIL.open(media['.jpg'])).astype(int)
    assert dec.shape == (24, 32, 3) and np.abs(dec - img[..., :3]).mean() < 3
    media, _ = output.PILOutput('jpeg', alpha=True).encode(img)
    assert sorted(media) == ['_alpha.jpg', '_color.jpg']
    assert np.abs(np.array(PIL.open(media['_alpha.jpg'])).astype(int) - img[..., 3]).mean() < 3
    # the built-in PNG writer decodes to the same pixels
    media, _ = output.PNGOutput().encode(img)
    assert np.array_equal(np.array(PIL.open(media['.png'])), img[..., :3])
    # 8-bit TIFF through Pillow; 16-bit RGB(A) is beyond Pillow: parse the baseline tags by hand
    t8 = output._tiff_bytes(img[..., :3])
    assert np.array_equal(np.array(PIL.open(__import__('io').BytesIO(t8))), img[..., :3])
    img16 = rs.randint(0, 65536, (6, 5, 4)).astype(np.uint16)
    for alpha in (False, True):
        media, _ = output.TiffOutput(alpha=alpha).encode(img16)
        data = media['.tiff'].read()
        want = img16 if alpha else img16[..., :3]
        assert data[:4] == b'II*\x00'
        ifd = int.from_bytes(data[4:8], 'little')
        n = int.from_bytes(data[ifd:ifd + 2], 'little')
        tags = {}
        for i in range(n):
            e = data[ifd + 2 + 12 * i: ifd + 14 + 12 * i]
            tag, typ, cnt = int.from_bytes(e[0:2], 'little'), int.from_bytes(e[2:4], 'little'), int.from_bytes(e[4:8], 'little')
            size = {3: 2, 4: 4}[typ] * cnt
            raw = e[8:8 + size] if size <= 4 else data[int.from_bytes(e[8:12], 'little'):][:size]
            tags[tag] = np.frombuffer(raw, '<u2' if typ == 3 else '<u4').tolist()
        assert list(tags) == sorted(tags)                       # IFD entries ascending, as TIFF requires
        assert tags[256] == [5] and tags[257] == [6] and tags[258] == [16] * want.shape[2]
        assert tags[259] == [1] and tags[262] == [2] and tags[277] == [want.shape[2]]
        assert (338 in tags) == alpha
        pix = np.frombuffer(data[tags[273][0]: tags[273][0] + tags[279][0]], '<u2').reshape(want.shape)
        assert np.array_equal(pix, want)
    assert output.TiffOutput().encode(None) == ({}, [])


def test_cli_print_blended_animation(tmp_path, capsys):
    """python -m cuburn_amd ID --print: genome db lookup + node -> animation, no GPU involved."""
    import json
    from cuburn_amd import __main__ as cli
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'genome_front.json')))
    (tmp_path / 'A.json').write_text(json.dumps(gold['db']['A']))
    assert cli.main(['A', '-d', str(tmp_path), '--print']) == 0
    text = capsys.readouterr().out
    assert text == gold['json_text']['A/full'] + '\n'
    assert cli.main(['A', '-d', str(tmp_path), '--print', '--half']) == 0
    assert capsys.readouterr().out == gold['json_text']['A/half'] + '\n'


@pytest.mark.parametrize('cfg', ['cfg1', 'cfg2', 'cfg3', 'cfg5'])
def test_per_genome_kernel_compiles(built, cfg):
    """The iterate kernel specialised for a genome's structure (hipRTC, csrc/rtc.hip) compiles for
    gfx950 for every BASELINE genome, both walker geometries and all accumulate modes.  hipRTC
    needs no GPU; the same entry point runs on a genome's first launch on the device."""
    import ctypes as C
    import numpy as np
    from cuburn_amd import _lib, configs
    from cuburn_amd.packer import GenomePacker
    lib = _lib.load()
    gnm, prof = configs.CONFIGS[cfg]()
    pk = GenomePacker(gnm)
    prog = np.ascontiguousarray(pk.prog, np.int32)
    ops = np.ascontiguousarray(pk.ops_array, np.int32)
    log = C.create_string_buffer(8192)
    for nw, count, acc in ((4, 0, 1), (4, 1, 0), (8, 0, 3), (8, 1, 1)):
        rc = lib.fl_rtc_compile_check(prog.ctypes.data, len(prog), ops.ctypes.data, len(ops), nw, count, acc, log, len(log))
        if rc == _lib.FL_E_UNSUPPORTED:
            pytest.skip('libhiprtc is not installed')
        assert rc == 0, log.value.decode()[:3000]


@pytest.mark.parametrize('cfg,limit', [('cfg2', 64), ('cfg3', 64), ('cfg5', 64)])
def test_per_genome_binned_kernel_register_budget(built, tmp_path, monkeypatch, cfg, limit):
    """The four-wave binned per-genome kernel of the 1080p configs fits 64 vector registers with no scratch (round 6: the batch
    epilogue derives its addresses from a thread number it takes anew — held across the kernel as loop invariants they made it
    79; rtc.hip's limits are 128 / 80 for the 1024- / 1536-slot geometries), the twelve heavy xforms of cfg5 included."""
    import ctypes as C
    import subprocess
    import numpy as np
    from cuburn_amd import _lib, configs
    from cuburn_amd.packer import GenomePacker
    readelf = '/opt/rocm/lib/llvm/bin/llvm-readelf'
    if not os.path.exists(readelf):
        pytest.skip('no llvm-readelf')
    monkeypatch.setenv('FLAME_RTC_DUMP', str(tmp_path))
    monkeypatch.delenv('FLAME_RTC_FLAGS', raising=False)
    lib = _lib.load()
    gnm, prof = configs.CONFIGS[cfg]()
    pk = GenomePacker(gnm)
    prog = np.ascontiguousarray(pk.prog, np.int32)
    ops = np.ascontiguousarray(pk.ops_array, np.int32)
    log = C.create_string_buffer(8192)
    rc = lib.fl_rtc_compile_check(prog.ctypes.data, len(prog), ops.ctypes.data, len(ops), 4, 0, 1, log, len(log))
    if rc == _lib.FL_E_UNSUPPORTED:
        pytest.skip('libhiprtc is not installed')
    assert rc == 0, log.value.decode()[:3000]
    notes = subprocess.run([readelf, '--notes', str(tmp_path / 'k_iter_spec.co')], capture_output=True, text=True, timeout=60).stdout
    num = lambda key: int(re.search(r'\.' + key + r':\s+(\d+)', notes).group(1))
    assert num('vgpr_count') <= limit and num('vgpr_spill_count') == 0 and num('private_segment_fixed_size') == 0, notes[-1500:]


def test_asm_issued_loads_are_not_touched_in_flight(tmp_path):
    """k_accum_tiles issues its record loads and the tile add's returning atomics from inline asm and
    waits for them itself (binned.hip: ACC_PIPE, ACC_ADD_ILP), so the compiler does not know those
    registers are in flight.  The device assembly of the shipped configuration is checked: no
    instruction may read or write such a register between its load / atomic and the wait, and the
    128x64 kernel must keep the budget that lets two workgroups share a CU (64 VGPRs, 80 SGPRs, no
    scratch: profiles/r03_occupancy_probe.txt)."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(REPO, 'cuburn_amd', 'csrc', 'binned.hip')
    asm = str(tmp_path / 'binned.s')
    # the Makefile runs the same checker on the assembly of the object it ships (same flags) and fails the build on a
    # violation; here the default flags once more, for the resource numbers, and the checker's own self-test
    mk = open(os.path.join(REPO, 'cuburn_amd', 'csrc', 'Makefile')).read()
    assert 'check_asm_atomics.py $(BUILD)/binned.s' in mk
    r = subprocess.run([hipcc, '-O3', '-std=c++20', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden',
                        '--cuda-device-only', '-S', src, '-o', asm, '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    checker = os.path.join(REPO, 'tools', 'check_asm_atomics.py')
    chk = subprocess.run([sys.executable, checker, asm], capture_output=True, text=True, timeout=120)
    assert chk.returncode == 0, chk.stdout[-2000:]
    assert not chk.stdout.startswith('0 asm-issued'), chk.stdout
    # self-test: an instruction at the top of the pipelined loop that touches the record set in flight ACROSS the
    # back-edge must be reported (by the second pass over the loop body), one that touches the completed set must not
    lines = open(asm).read().split('\n')
    load = re.compile(r'\s*global_load_dword v(\d+), v(\d+), s\[(\d+):(\d+)\]$')
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loads = [(i, int(load.match(l).group(1))) for i, l in enumerate(lines) if load.match(l)]
    head = back = None
    for i in range(loads[0][0], len(lines)):
        t = lines[i].strip().split()
        if t and t[0].startswith('s_cbranch') and t[1] in labels and labels[t[1]] < loads[0][0]:
            head, back = labels[t[1]], i
            break
    assert head is not None
    flagged = 0
    for reg in sorted(set(r for i, r in loads if head < i < back)):
        inj = str(tmp_path / 'inj.s')
        open(inj, 'w').write('\n'.join(lines[:head + 1] + ['\tv_mov_b32_e32 v0, v%d' % reg] + lines[head + 1:]))
        out = subprocess.run([sys.executable, checker, inj], capture_output=True, text=True, timeout=120)
        flagged += out.returncode != 0 and 'second pass' in out.stdout
    assert 2 <= flagged <= 4, flagged                      # one of the two record sets (2-4 registers) is in flight there
    # resource usage of the two 128x64-tile kernels (the packed log's, and the 32-bit log's of a build without it), from the compiler's remarks
    for kern in ('k_accum_tiles_p3', 'k_accum_tilesILj7'):
        rem = r.stderr[r.stderr.index(kern):]
        num = lambda key: int(re.search(key + r': (\d+)', rem).group(1))
        assert num('VGPRs') <= 64 and num('TotalSGPRs') <= 80 and num(r'ScratchSize \[bytes/lane\]') == 0, (kern, rem[:1200])
    # the same self-test on the packed log's loop: three register sets of a 64-bit word each, two of them in flight across the back-edge
    load2 = re.compile(r'\s*global_load_dwordx2 v\[(\d+):(\d+)\], v(\d+), s\[(\d+):(\d+)\]$')
    p3 = [i for i, l in enumerate(lines) if 'k_accum_tiles_p3' in l and l.rstrip().endswith(':') or l.startswith('_Z16k_accum_tiles_p3')]
    lo = min(i for i, l in enumerate(lines) if l.startswith('_Z16k_accum_tiles_p3'))
    hi = min(i for i, l in enumerate(lines) if i > lo and l.startswith('_Z13k_accum_tiles'))
    loads2 = [(i, int(load2.match(l).group(1)), int(load2.match(l).group(2))) for i, l in enumerate(lines) if lo < i < hi and load2.match(l)]
    assert len(loads2) == 3, loads2                          # one request site per set
    head = back = None
    for i in range(loads2[-1][0], hi):
        t = lines[i].strip().split()
        if t and t[0].startswith('s_cbranch') and t[1] in labels and labels[t[1]] < loads2[0][0]:
            head, back = labels[t[1]], i
            break
    assert head is not None
    flagged = 0
    for _, a, b in loads2:
        for reg in range(a, b + 1):
            inj = str(tmp_path / 'inj2.s')
            open(inj, 'w').write('\n'.join(lines[:head + 1] + ['\tv_mov_b32_e32 v0, v%d' % reg] + lines[head + 1:]))
            out = subprocess.run([sys.executable, checker, inj], capture_output=True, text=True, timeout=120)
            flagged += out.returncode != 0 and 'second pass' in out.stdout
    assert flagged == 4, flagged                             # two of the three sets are in flight at the top of the loop


def test_de_kernels_hold_32_waves_per_cu_without_scratch(tmp_path):
    """Every k_de_dir kernel carries two forms of its 31-tap loop (waves whose centres are all live / waves with a dead
    centre) and must still fit 64 vector registers — eight waves per SIMD — with no scratch: a private segment costs
    every wave its set-up whether it spills on its path or not (de.hip: each form finds its pixel itself, the rare form
    is not software-pipelined)."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(REPO, 'cuburn_amd', 'csrc', 'de.hip')
    r = subprocess.run([hipcc, '-O3', '-std=c++20', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC', '-fvisibility=hidden',
                        '-fgpu-flush-denormals-to-zero', '-fno-slp-vectorize', '--cuda-device-only', '-S', src, '-o', str(tmp_path / 'de.s'),
                        '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = re.split(r'Function Name: ', r.stderr)[1:]
    de = [k for k in kernels if k.startswith('_Z8k_de_dir')]
    assert len(de) == 9                                    # 8 directions, 2 input forms of the first (raw / raw YUV accumulator)
    for k in de:
        num = lambda key: int(re.search(key + r': (\d+)', k).group(1))
        assert num('VGPRs') <= 64 and num(r'ScratchSize \[bytes/lane\]') == 0 and num('TotalSGPRs') <= 80, k[:300]
