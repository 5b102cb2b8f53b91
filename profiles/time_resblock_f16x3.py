"""The fused fp16 residual block (csrc/resblock_f16x3.hip) beside the two-launch form (csrc/trunk_bf16x6.hip) on one decode
chunk, stand-alone.   python profiles/time_resblock_f16x3.py [rows]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32256
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
