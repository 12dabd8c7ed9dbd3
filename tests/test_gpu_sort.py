"""
Device radix sort (csrc/sort.hip, cuburn_amd/sort.py) against numpy.  The reference's own check is
``np.all(out == np.sort(keys))`` for one pass over keys below 2^radix_bits, 2^25 keys, seed 42
(cuburn/code/sort.py:524-559,583); here also: stability of a pass (which the reference's pass lacks),
the full 32-bit sort from four passes, ragged sizes, dropped 0xffffffff keys.
"""
import numpy as np
import pytest
import torch

from cuburn_amd import render, _lib
from cuburn_amd.sort import Sorter

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def fb(built):
    f = render.Framebuffers(device=0, nslots=1024, host_seed=7)
    yield f
    f.free()


def dev(a):
    return torch.from_numpy(a.view(np.int32)).to('cuda:0')


def host(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize('bits', [7, 8, 9, 10])
def test_single_pass_matches_np_sort_like_the_reference_test(fb, bits):
    """sort.py:524-559: keys below 2^radix_bits, one pass, equal to np.sort; count = 2^25, seed 42."""
    count = 1 << 25
    np.random.seed(42)
    keys = np.uint32(np.random.randint(0, 1 << bits, size=count))
    s = Sorter(count, fb=fb)
    s.radix_bits = bits
    src, dst = dev(keys), torch.empty(count, dtype=torch.int32, device='cuda:0')
    s.sort(dst, src, count)
    torch.cuda.synchronize()
    _lib.check(_lib.load().fl_ctx_sync(fb.ctx))
    assert np.array_equal(host(dst), np.sort(keys))
    assert np.array_equal(host(src), keys)                      # the source is left alone


@pytest.mark.parametrize('size', [1, 63, 4096, 4097, 1000003])
def test_a_pass_is_stable_for_any_size(fb, size):
    """Sorting by a middle digit keeps equal digits in their original order — exactly numpy's stable
    argsort — for sizes that are not multiples of the 4096-key tile."""
    rs = np.random.RandomState(size)
    keys = rs.randint(0, 2 ** 32, size=size, dtype=np.uint64).astype(np.uint32)
    s = Sorter(max(size, 4096), fb=fb)
    src, dst = dev(keys), torch.empty(size, dtype=torch.int32, device='cuda:0')
    s.sort(dst, src, size, lo_bit=11)
    _lib.check(_lib.load().fl_ctx_sync(fb.ctx))
    want = keys[np.argsort((keys >> 11) & 0xff, kind='stable')]
    assert np.array_equal(host(dst), want)


def test_four_passes_sort_all_32_bits(fb):
    n = (1 << 22) + 12345
    rs = np.random.RandomState(1)
    keys = rs.randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    keys[::7] = keys[3]                                          # many duplicates
    s = Sorter(n, fb=fb)
    src = dev(keys)
    a, b = torch.empty_like(src), torch.empty_like(src)
    out = s.multisort(a, b, src, n, rounds=4)
    _lib.check(_lib.load().fl_ctx_sync(fb.ctx))
    assert out is b or out is a
    assert np.array_equal(host(out), np.sort(keys))
    assert np.array_equal(host(src), keys)
    # 10-bit digits: 4 passes, the last one clipped to 2 bits
    s.radix_bits = 10
    out = s.multisort(a, b, src, n, rounds=4)
    _lib.check(_lib.load().fl_ctx_sync(fb.ctx))
    assert np.array_equal(host(out), np.sort(keys))


def test_ignore_max_drops_the_sentinel_keys(fb):
    """sort.py:449-452: keys 0xffffffff are discarded; the count of valid results comes back."""
    n = 300000
    rs = np.random.RandomState(2)
    keys = rs.randint(0, 256, size=n).astype(np.uint32)
    drop = rs.uniform(size=n) < 0.3
    keys[drop] = 0xffffffff
    s = Sorter(n, fb=fb)
    src, dst = dev(keys), torch.zeros(n, dtype=torch.int32, device='cuda:0')
    s.sort(dst, src, n, ignore_max=True, count=True)
    kept = keys[~drop]
    assert s.nvalid == len(kept)
    assert np.array_equal(host(dst)[:s.nvalid], np.sort(kept))
    assert (host(dst)[s.nvalid:] == 0).all()                     # nothing written past the valid keys


def test_sort_argument_validation(fb):
    lib = _lib.load()
    t = torch.zeros(4096, dtype=torch.int32, device='cuda:0')
    u = torch.zeros(4096, dtype=torch.int32, device='cuda:0')
    assert lib.fl_sort_u32(fb.ctx, t.data_ptr(), t.data_ptr(), 4096, 0, 8, 0, None) == _lib.FL_E_INVAL     # in place
    assert lib.fl_sort_u32(fb.ctx, t.data_ptr(), u.data_ptr(), 4096, 0, 11, 0, None) == _lib.FL_E_INVAL    # digit too wide
    assert lib.fl_sort_u32(fb.ctx, t.data_ptr(), u.data_ptr(), 4096, 28, 8, 0, None) == _lib.FL_E_INVAL    # past bit 32
    assert lib.fl_sort_u32(fb.ctx, t.data_ptr(), u.data_ptr(), 0, 0, 8, 0, None) == 0
    with pytest.raises(ValueError):
        Sorter(100, fb=fb).sort(t, u, 4096)
