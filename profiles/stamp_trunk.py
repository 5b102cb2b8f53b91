"""Per-wave cycle accounting of the fused residual-block kernel (debug build with s_memtime stamps).
Usage: python profiles/stamp_trunk.py [extra -D flags]"""
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
    p0, p1 = pk.ops.pack_trunk_rows(w0), pk.ops.pack_trunk_cols(w1)
    out = '/tmp/trunk_stamp.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                    '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-DOCC4D_TR_STAMP'] + sys.argv[1:] +
                   [os.path.join(CSRC, 'trunk.hip'), os.path.join(CSRC, 'error.hip'), '-o', out], check=True,
                   stderr=subprocess.DEVNULL)
    lib = C.CDLL(out)
    fn = lib.occ4d_resblock_f32
    fn.restype = C.c_int
    fn.argtypes = pk._lib.SIGNATURES['occ4d_resblock_f32'][1]
    nwg = (n + 127) // 128
    stamps = torch.zeros((nwg * 8, 6), dtype=torch.int64, device='cuda')
    for _ in range(3):
        assert fn(x.data_ptr(), H, y.data_ptr(), H, p0.data_ptr(), b.data_ptr(), p1.data_ptr(), b.data_ptr(),
                  None, None, 0, None, stamps.data_ptr(), 0, n, None) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(x.data_ptr(), H, y.data_ptr(), H, p0.data_ptr(), b.data_ptr(), p1.data_ptr(), b.data_ptr(),
       None, None, 0, None, stamps.data_ptr(), 0, n, None)
    e1.record()
    torch.cuda.synchronize()
    wall_us = e0.elapsed_time(e1) * 1e3
    s = stamps.cpu().numpy().astype(np.float64).reshape(nwg, 8, 6)
    names = ['stage A (issue .. last MFMA)', 'wait at barrier 1', 'stage B', 'wait at barrier 2', 'loop total']
    print('cycles (s_memtime ticks) per wave, summed over the 13 iterations; mean over workgroups')
    for w in range(8):
        print('wave %d: ' % w + '  '.join('%s %.0f' % (names[i].split(' (')[0], s[:, w, i].mean()) for i in range(5)))
    print('mean over all waves: ' + '  '.join('%s %.0f' % (names[i], s[:, :, i].mean()) for i in range(5)))
    raw = stamps.cpu().numpy().reshape(nwg, 8, 6)
    t0 = raw[:, :, 5]
    t1 = raw[:, :, 5] + raw[:, :, 4]
    span = int(t1.max() - t0.min())
    print('kernel wall time %.1f us; first loop entry -> last loop exit across the grid: %d ticks -> %.3f ticks per ns'
          % (wall_us, span, span / (wall_us * 1e3)))
    print('loop start skew inside a workgroup (max - min): %.0f ticks; across the grid: %d' % (
        (t0.max(1) - t0.min(1)).mean(), int(t0.max() - t0.min())))
    print('ideal per wave: 13 x 2 x 208 MFMAs x 32 cycles = %d MFMA-pipe cycles (the pipe is shared by 2 waves -> %d wall)'
          % (13 * 2 * 208 * 32, 13 * 2 * 208 * 64))


if __name__ == '__main__':
    main()
