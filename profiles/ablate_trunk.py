"""Ablation of the fused residual-block kernel on the GPU box (timing only; ablated variants compute garbage).
Usage: python profiles/ablate_trunk.py [variant ...]"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402

CSRC = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc')
VARIANTS = {
    'base': [],
    'nodma': ['-DOCC4D_TR_NODMA'],
    'nobar': ['-DOCC4D_TR_NOBAR'],
    'nodma-nobar': ['-DOCC4D_TR_NODMA', '-DOCC4D_TR_NOBAR'],
    'nopro': ['-DOCC4D_TR_NOPRO'],
    'noepi': ['-DOCC4D_TR_NOEPI'],
    'nopro-noepi': ['-DOCC4D_TR_NOPRO', '-DOCC4D_TR_NOEPI'],
    'bare-loop': ['-DOCC4D_TR_NOPRO', '-DOCC4D_TR_NOEPI', '-DOCC4D_TR_NODMA', '-DOCC4D_TR_NOBAR'],
}


def main():
    n, H = 32256, 416
    x = torch.randn(n, H, device='cuda')
    w0 = torch.randn(H, H, device='cuda') * 0.05
    w1 = torch.randn(H, H, device='cuda') * 0.05
    b = torch.randn(H, device='cuda')
    y = torch.empty(n, H, device='cuda')
    p0, p1 = pk.ops.pack_trunk_rows(w0), pk.ops.pack_trunk_cols(w1)
    names = sys.argv[1:] or list(VARIANTS)
    extra = os.environ.get('OCC4D_ABLATE_EXTRA', '').split()
    for name in names:
        out = '/tmp/trunk_%s.so' % name
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                        '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + VARIANTS[name] + extra +
                       [os.path.join(CSRC, 'trunk.hip'), os.path.join(CSRC, 'error.hip'), '-o', out], check=True)
        lib = C.CDLL(out)
        fn = lib.occ4d_resblock_f32
        fn.restype = C.c_int
        fn.argtypes = pk._lib.SIGNATURES['occ4d_resblock_f32'][1]
        run = lambda: fn(x.data_ptr(), H, y.data_ptr(), H, p0.data_ptr(), b.data_ptr(), p1.data_ptr(), b.data_ptr(),
                         None, None, 0, None, None, 0, n, None)
        assert run() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%-14s %8.1f us  %6.1f TFLOP/s  %.3f of peak' % (name, us, 4.0 * n * H * H / us / 1e6, 4.0 * n * H * H / us / 1e6 / 157.3),
              flush=True)


if __name__ == '__main__':
    main()
