"""The fused fp16 residual block (csrc/resblock_f16x3.hip) beside the two-launch form (csrc/trunk_bf16x6.hip) on one decode
chunk, stand-alone.   python profiles/time_resblock_f16x3.py [rows] [-DOCC4D_RB_ABL_...[,-D...]] ...
Every -D argument rebuilds csrc/resblock_f16x3.hip with those timing-only macros into /tmp and times it beside the shipped kernel
(results of an ablated kernel are garbage): NODMA (no weight stream after stage 0), NOLDS (a stage's first four fragments serve
every tile), NOX (no operand-row loads after stage 0), NOBAR (no stage barrier), NOSTORE."""
import ctypes as C
import subprocess
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import ops  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('-')]
variants = [a.split(',') for a in sys.argv[1:] if a.startswith('-')]
n = int(args[0]) if args else 32256
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()       # noqa: E731
x, w0, w1, b0, b1 = rnd(n, 416), rnd(416, 416) / 20, rnd(416, 416) / 20, rnd(416), rnd(416)
L = ops._lib.lib()
packed = torch.empty((int(L.occ4d_resblock_f16x3_packed_floats()),), device='cuda')
ops._lib.check(L.occ4d_pack_resblock_f16x3_f32(ops._ptr(w0), 416, ops._ptr(w1), 416, ops._ptr(packed), ops._stream()))
p0, p1 = ops.pack_rowlin_bf16x6(w0, 'f16x3'), ops.pack_rowlin_bf16x6(w1, 'f16x3')
h, y = torch.empty_like(x), torch.empty_like(x)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def two():
    ops.rowlin_bf16x6(x, None, b0, relu_in=True, out=h, packed=p0, n_out=416, scheme='f16x3')
    ops.rowlin_bf16x6(h, None, b1, relu_in=True, res=x, out=y, packed=p1, n_out=416, scheme='f16x3')


flop = 2 * 2.0 * n * 416 * 416 * 3
for name, fn in (('fused block', lambda: ops.resblock_f16x3(x, None, b0, None, b1, out=y, packed=packed)), ('two launches', two)):
    us = t(fn)
    print('%-14s %7.1f us per block of %d rows   %5.0f TFLOP/s of executed 16x16x32 MFMA = %.2f of 2.5 PF'
          % (name, us, n, flop / us / 1e6, flop / us / 1e6 / 2500))


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for defs in variants:
    csrc, build = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc'), os.path.join(ROOT, 'occlusions-4d_amd', 'build')
    tag = '_'.join(d.replace('-DOCC4D_RB_ABL_', '') for d in defs)
    obj, so = '/tmp/rb_%s.o' % tag, '/tmp/rb_%s.so' % tag
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-fno-honor-nans'] + defs +
                   ['-c', os.path.join(csrc, 'resblock_f16x3.hip'), '-o', obj], check=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so, obj] +
                   [os.path.join(build, f) for f in sorted(os.listdir(build)) if f.endswith('.o') and f != 'resblock_f16x3.o'],
                   check=True)
    K = C.CDLL(so)
    fn = K.occ4d_resblock_f16x3_f32
    fn.restype, fn.argtypes = ops._lib.SIGNATURES['occ4d_resblock_f16x3_f32']

    def run():
        assert fn(ops._ptr(x), 416, ops._ptr(y), 416, ops._ptr(packed), ops._ptr(b0), ops._ptr(b1), n, ops._stream()) == 0
    us = t(run)
    print('%-28s %7.1f us per block of %d rows   %5.0f TFLOP/s = %.2f of 2.5 PF' % (tag, us, n, flop / us / 1e6, flop / us / 1e6 / 2500))
