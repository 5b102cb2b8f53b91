"""occ4d_rowlin_bf16x6_f32 stand-alone at the decode chunk (32256 rows): time per launch by shape / epilogue, the weight
packed once.  fp32-equivalent TFLOP/s = 2 n 416 n_out / time; executed bf16 MFMA FLOP = 6 x that.
Extra -D flags rebuild csrc/trunk_bf16x6.hip as a timing-only variant (-DOCC4D_X6T_ABL_NOSTORE: no y stores,
-DOCC4D_X6T_ABL_NORES: no residual loads, -DOCC4D_X6T_ABL_NOX: the per-stage x loads dropped).
Usage: python profiles/time_rowlin_x6.py [rows] [-D...]"""
import ctypes as C
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402
from occlusions4d_amd import ops  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [f for f in sys.argv[1:] if f.startswith('-')]
rows = [f for f in sys.argv[1:] if not f.startswith('-')]
n = int(rows[0]) if rows else 32256
dev = torch.device('cuda:0')
L = ops._lib.lib()
K = L
if flags:
    csrc, build = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc'), os.path.join(ROOT, 'occlusions-4d_amd', 'build')
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-fno-honor-nans'] + flags +
                   ['-c', os.path.join(csrc, 'trunk_bf16x6.hip'), '-o', '/tmp/x6t_var.o'], check=True)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', '/tmp/x6t_var.so', '/tmp/x6t_var.o'] +
                   [os.path.join(build, f) for f in sorted(os.listdir(build)) if f.endswith('.o') and f != 'trunk_bf16x6.o'],
                   check=True)
    K = C.CDLL('/tmp/x6t_var.so')
    K.occ4d_rowlin_bf16x6_f32.restype = C.c_int
    K.occ4d_rowlin_bf16x6_f32.argtypes = ops._lib.SIGNATURES['occ4d_rowlin_bf16x6_f32'][1]
    print('variant:', ' '.join(flags))
torch.manual_seed(0)
x = torch.randn(n, 416, device=dev)
res = torch.randn(n, 832, device=dev)
for n_out, relu_in, with_res in ((416, True, False), (416, True, True), (416, False, True), (832, False, False)):
    w = torch.randn(n_out, 416, device=dev) / 20
    b = torch.randn(n_out, device=dev)
    packed = torch.empty((int(L.occ4d_rowlin_bf16x6_packed_floats(n_out)),), dtype=torch.float32, device=dev)
    ops._lib.check(L.occ4d_pack_rowlin_bf16x6_f32(ops._ptr(w), 416, n_out, ops._ptr(packed), ops._stream()))
    y = torch.empty(n, n_out, device=dev)
    r = res[:, :n_out] if with_res else None

    def run():
        ops._lib.check(K.occ4d_rowlin_bf16x6_f32(ops._ptr(x), 416, ops._ptr(y), n_out, ops._ptr(packed), ops._ptr(b), n_out,
                                                 int(relu_in), ops._ptr(r) if r is not None else None, 832 if with_res else 0,
                                                 n, ops._stream()))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ref = (torch.relu(x) if relu_in else x).double() @ w.double().t() + b.double() + (r.double() if with_res else 0)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    tf = 2.0 * n * 416 * n_out / us / 1e6
    print('n=%d n_out=%4d relu_in=%d res=%d  %7.1f us  %6.1f TFLOP/s fp32-equivalent  (%.2f of the dense bf16 peak executed)  '
          'max rel err vs fp64 %.1e' % (n, n_out, relu_in, with_res, us, tf, 6 * tf / 2500, err))
