"""Ablation of the Linear kernel on the GPU box (timing only; ablated variants compute garbage)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402

CSRC = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc')
VARIANTS = {'bk32': ['-DOCC4D_LINEAR_BK=32'], 'bk32-nopipe': ['-DOCC4D_LINEAR_BK=32', '-DOCC4D_LINEAR_NO_PIPE'],
            'bk32-noload': ['-DOCC4D_LINEAR_BK=32', '-DOCC4D_ABLATE_NOLOAD'],
            'bk32-noepi': ['-DOCC4D_LINEAR_BK=32', '-DOCC4D_ABLATE_NOEPI'],
            'bk32-both': ['-DOCC4D_LINEAR_BK=32', '-DOCC4D_ABLATE_NOLOAD', '-DOCC4D_ABLATE_NOEPI']}


def main():
    M, K, N = 32768, 416, 416
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') * 0.05
    b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    for name, flags in VARIANTS.items():
        out = '/tmp/lin_%s.so' % name
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                        '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + flags +
                       [os.path.join(CSRC, 'linear.hip'), os.path.join(CSRC, 'error.hip'), '-o', out], check=True)
        lib = C.CDLL(out)
        lib.occ4d_linear_f32.restype = C.c_int
        lib.occ4d_linear_f32.argtypes = [C.POINTER(pk._lib.LinearArgs), C.c_void_p]
        for label, res in (('plain', None), ('relu+bias+residual', r)):
            a = pk._lib.LinearArgs()
            a.x, a.ldx, a.w, a.ldw, a.y, a.ldy = x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N
            a.M, a.K, a.N = M, K, N
            if res is not None:
                a.bias, a.residual, a.ldr, a.relu_in = b.data_ptr(), res.data_ptr(), N, 1
            run = lambda: lib.occ4d_linear_f32(C.byref(a), None)
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print('%-12s %-20s %.1f us  (%.1f TFLOP/s)' % (name, label, 1e3 * ms, 2.0 * M * K * N / ms / 1e9), flush=True)


if __name__ == '__main__':
    main()
