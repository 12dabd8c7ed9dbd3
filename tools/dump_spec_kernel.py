#!/usr/bin/env python3
"""Disassemble the per-genome iterate kernel of a BASELINE config (no GPU needed):
    python tools/dump_spec_kernel.py cfg2 [nw=4] [acc=1] [pair=0] > /tmp/k.s
Uses FLAME_RTC_DUMP (csrc/rtc.hip) to keep the code object of fl_rtc_compile_check."""
import ctypes as C, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 4
acc = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pair = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 1: paired halves (nw = 8)
d = tempfile.mkdtemp()
os.environ['FLAME_RTC_DUMP'] = d
from cuburn_amd import _lib, configs
from cuburn_amd.packer import GenomePacker
lib = _lib.load()
gnm, prof = configs.CONFIGS[cfg]()
pk = GenomePacker(gnm)
prog = np.ascontiguousarray(pk.prog, np.int32); ops = np.ascontiguousarray(pk.ops_array, np.int32)
log = C.create_string_buffer(8192)
rc = lib.fl_rtc_compile_check(prog.ctypes.data, len(prog), ops.ctypes.data, len(ops), nw, 2 * pair, acc, log, len(log))
assert rc == 0, log.value.decode()
sys.stdout.write(open(os.path.join(d, 'flame_spec.h')).read())
sys.stdout.flush()
subprocess.check_call(['/opt/rocm/lib/llvm/bin/llvm-objdump', '-d', '--no-show-raw-insn', os.path.join(d, 'k_iter_spec.co')])
