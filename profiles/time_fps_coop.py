"""Cooperative (multi-workgroup) FPS with several samples per exchange against the single-workgroup kernel: the training
cloud (28672 -> 7168), the dataloader's clip (172032 -> 14336); HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402

ops = pk.ops
g = torch.Generator(device='cuda').manual_seed(0)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for n, m in ((28672, 7168), (28672, 9558), (20000, 5000), (57344, 14336), (172032, 14336)):
    p = torch.rand((n, 3), device='cuda', generator=g) * torch.tensor([40.0, 32.0, 7.4], device='cuda')
    line = 'n=%6d m=%5d ' % (n, m)
    if n <= 28672:
        t1 = timed(lambda: ops.fps(p, m, start=3))
        line += ' single workgroup %7.2f ms (%.2f us/sample)' % (t1, 1e3 * t1 / m)
        a = ops.fps(p, m, start=3, return_order=True)[1]
    for wgs in ((16, 8) if n <= 65536 else (16,)):
        t2 = timed(lambda: ops.fps_coop(p, m, start=3, n_workgroups=wgs, check=False))
        line += '   cooperative x%d %7.2f ms (%.2f us/sample)' % (wgs, t2, 1e3 * t2 / m)
    if n <= 28672:
        b = ops.fps_coop(p, m, start=3, n_workgroups=16, return_order=True)[1]
        line += '   same order: %s' % bool(torch.equal(a, b))
    print(line, flush=True)
ops.check_pending()
