"""
ctypes binding of libflame_hip.so (include/flame_hip.h).  The library is the product: there
is no CPU fallback — importing works without it (host-only code such as profile / packer),
but any device entry point raises if the library is missing or reports an error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_lib', 'libflame_hip.so')
# A/B measurements: an alternative build of the same library (never a different implementation)
LIB_PATH = os.environ.get('FLAME_HIP_LIB', LIB_PATH)

(FL_OK, FL_E_INVAL, FL_E_NOMEM, FL_E_HIP, FL_E_NODEV, FL_E_UNSUPPORTED) = (0, -1, -2, -3, -4, -5)

FILT = dict(yuv=0, bilateral=1, logscale=2, colorclip=3, smearclip=4, haloclip=5, plainclip=6, logencode=7)
BUF = dict(front=0, back=1, params=2, palette=3, points=4, seeds=5, atom=6, hot=7, side=8)
ACCUM_ATOMIC, ACCUM_BINNED = 0, 1
OUT = dict(rgba8=0, rgba16=1, yuv444p=2, yuv444p10=3, yuv420p10=4, yuv444p12=5)     # include/flame_hip.h FL_OUT_*


class fl_dim(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ('w', 'h', 'aw', 'ah', 'astride')]


_SIGS = {
    'fl_abi_version': (C.c_int, []),
    'fl_last_error': (C.c_char_p, []),
    'fl_calc_dim': (None, [C.c_uint32, C.c_uint32, C.POINTER(fl_dim)]),
    'fl_ctx_create': (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    'fl_ctx_destroy': (None, [C.c_void_p]),
    'fl_ctx_sync': (C.c_int, [C.c_void_p]),
    'fl_genome_create': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_void_p)]),
    'fl_genome_destroy': (None, [C.c_void_p]),
    'fl_genome_upload': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    'fl_interp': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float]),
    'fl_iterate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_int,
                             C.POINTER(C.c_uint64)]),
    'fl_filter': (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]),
    'fl_output': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64]),
    'fl_output_bytes': (C.c_size_t, [C.c_uint32, C.c_uint32, C.c_int]),
    'fl_sort_u32': (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]),
    'fl_frame_begin': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    'fl_frame_ms': (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]),
    'fl_frame_query': (C.c_int, [C.c_void_p, C.c_uint32]),
    'fl_host_alloc': (C.c_void_p, [C.c_size_t]),
    'fl_host_free': (None, [C.c_void_p]),
    'fl_timings_reset': (C.c_int, [C.c_void_p]),
    'fl_timings': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                             C.POINTER(C.c_uint32)]),
    'fl_timings_detail': (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 6)]),
    'fl_launch_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32 * 4)]),
    'fl_measure_copy': (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    'fl_read_buffer': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    'fl_write_buffer': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    'fl_buffer_ptr': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    'fl_buffer_ptr_async': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    'fl_stream_dependency': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'fl_reserve': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    'fl_debug_iter_launch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_int]),
    'fl_debug_flush': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    'fl_debug_clear': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    'fl_debug_clear_hot': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    'fl_debug_shuffle': (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    'fl_debug_counters': (C.c_int, [C.c_void_p, C.c_void_p]),
    'fl_rtc_compile_check': (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    'fl_debug_apply_xf': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]),
}
EXPORTS = sorted(_SIGS)

_lib = None


class FlameError(RuntimeError):
    pass


def load():
    """Load libflame_hip.so (once).  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise FlameError('%s not found: build it with `python __graft_entry__.py` '
                         '(make -C cuburn_amd/csrc); the HIP library is required' % LIB_PATH)
    try:                        # share torch's HIP runtime when torch is in the process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.fl_abi_version() != 1:
        raise FlameError('libflame_hip ABI mismatch')
    _lib = lib
    return lib


def check(rc):
    """Map a status code to the exception the reference would raise (render.py:140-147)."""
    if rc == FL_OK:
        return
    msg = load().fl_last_error().decode('utf-8', 'replace')
    if rc == FL_E_NOMEM:
        raise MemoryError(msg)
    if rc in (FL_E_INVAL, FL_E_UNSUPPORTED):
        raise ValueError(msg)
    raise FlameError(msg)
