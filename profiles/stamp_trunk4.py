"""Per-wave cycle accounting of the half-CU residual-block kernel (csrc/trunk4.hip, debug build with s_memtime stamps).
Usage: python profiles/stamp_trunk4.py [extra -D flags]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occlusions4d_amd as pk  # noqa: E402

CSRC = os.path.join(ROOT, 'occlusions-4d_amd', 'csrc')


def main():
    n, H = 32256, 416
    x = torch.randn(n, H, device='cuda')
    w0 = torch.randn(H, H, device='cuda') * 0.05
    w1 = torch.randn(H, H, device='cuda') * 0.05
    b = torch.randn(H, device='cuda')
    y = torch.empty(n, H, device='cuda')
    p0, p1 = pk.ops.pack_trunk4_rows(w0), pk.ops.pack_trunk4_cols(w1)
    out = '/tmp/trunk4_stamp.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                    '-fno-honor-nans', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-DOCC4D_TR4_STAMP'] + sys.argv[1:] +
                   [os.path.join(CSRC, 'trunk4.hip'), os.path.join(CSRC, 'error.hip'), '-o', out], check=True,
                   stderr=subprocess.DEVNULL)
    lib = C.CDLL(out)
    fn = lib.occ4d_resblock4_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_resblock4_f32'][1]
    nwg = (n + 63) // 64
    stamps = torch.zeros((nwg * 4, 6), dtype=torch.int64, device='cuda')
    call = lambda: fn(x.data_ptr(), H, y.data_ptr(), H, p0.data_ptr(), b.data_ptr(), p1.data_ptr(), b.data_ptr(),   # noqa: E731
                      None, None, 0, None, stamps.data_ptr(), 0, n, None)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3
    s = stamps.cpu().numpy().astype(np.float64).reshape(nwg, 4, 6)
    names = ['stage A', 'wait at barrier 1', 'stage B', 'wait at barrier 2', 'loop total']
    print('cycles per wave, summed over the 26 hidden chunks; mean over %d workgroups' % nwg)
    for w in range(4):
        print('wave %d: ' % w + '  '.join('%s %.0f' % (names[i], s[:, w, i].mean()) for i in range(5)))
    print('mean over all waves: ' + '  '.join('%s %.0f' % (names[i], s[:, :, i].mean()) for i in range(5)))
    rt = s[:, :, 5].mean()      # s_memrealtime ticks (100 MHz) across the loop
    print('loop: %.0f shader cycles in %.1f us of s_memrealtime (100 MHz) -> shader clock %.3f GHz' % (
        s[:, :, 4].mean(), rt / 100.0, s[:, :, 4].mean() / (rt * 10.0)))
    print('kernel wall time %.1f us = %.0f cycles at 2.4 GHz' % (wall_us, wall_us * 2400))
    print('ideal per wave: 26 x 208 MFMAs x 32 cycles = %d MFMA-pipe cycles (the pipe is shared by 2 waves -> %d wall)'
          % (26 * 208 * 32, 26 * 208 * 64))


if __name__ == '__main__':
    main()
