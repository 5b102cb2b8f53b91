"""Row-resident Linear at a row count that is NOT a whole number of dispatch rounds (training: 68812 query rows = 2.1
rounds of 256 x 128-row workgroups): 8-wave kernel (one workgroup per CU) against the half-CU re-cut (two per CU: a
workgroup that has its CU to itself in the last round runs faster).  HIP events, 20 repeats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

ops = pk.ops
g = torch.Generator(device='cuda').manual_seed(0)
for n in (68812, 65536, 32768, 45864, 98304, 100000):
    x = torch.randn((n, 416), device='cuda', generator=g)
    w = torch.randn((416, 416), device='cuda', generator=g) * 0.05
    b = torch.zeros(416, device='cuda')
    packs = {'8-wave': ops.pack_trunk_rows(w), 'half-CU': ops.pack_trunk4_rows(w)}
    for name, pw in packs.items():
        out = torch.empty((n, 416), device='cuda')
        for _ in range(3):
            ops.rowlin(x, pw, b, 416, relu_in=True, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.rowlin(x, pw, b, 416, relu_in=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('n=%6d %-8s %7.1f us  %6.1f TFLOP/s  (%.2f rounds of 256 x 128 rows)' % (n, name, 1e3 * ms, 2.0 * n * 416 * 416 / ms / 1e9, n / 32768.0), flush=True)
