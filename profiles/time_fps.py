"""Times the single-workgroup and the cooperative FPS kernels on the GPU box (HIP events, 3 repeats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device='cuda').manual_seed(0)
for n, m in [(14336, 4779), (28672, 9558), (4779, 1593), (172032, 14336)]:
    p = torch.rand((n, 3), device='cuda', generator=g) * 10 - 5
    if n <= 32768:
        t = timeit(lambda: pk.ops.fps(p, m))
        print('single  n=%6d m=%5d  %8.3f ms  %.2f us/step' % (n, m, t, 1e3 * t / m), flush=True)
    for wgs in ([2, 4, 8, 16] if n <= 32768 else [16]):
        if -(-n // wgs) > 16 * (512 if n <= 65536 else 1024):
            continue
        t = timeit(lambda: pk.ops.fps_coop(p, m, n_workgroups=wgs, check=False))
        print('coop%-3d n=%6d m=%5d  %8.3f ms  %.2f us/step' % (wgs, n, m, t, 1e3 * t / m), flush=True)
