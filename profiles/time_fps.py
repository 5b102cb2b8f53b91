"""Times the single-workgroup FPS (pruned: csrc/fps_bucket.hip, exhaustive: OCC4D_FPS_PRUNE=0 -> csrc/fps.hip) and
the cooperative multi-workgroup kernel on the GPU box (HIP events, 3 repeats), on uniform clouds and on the bench's
synthetic GREATER cloud (whose three encoder levels are what the encode runs)."""
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


tag = 'exhaustive' if os.environ.get('OCC4D_FPS_PRUNE') == '0' else 'pruned'
g = torch.Generator(device='cuda').manual_seed(0)
for n, m in [(14336, 4779), (28672, 9558), (9558, 3186), (4779, 1593), (2049, 683), (16384, 5461)]:
    p = torch.rand((n, 3), device='cuda', generator=g) * 10 - 5
    try:
        t = timeit(lambda: pk.ops.fps(p, m))
    except AssertionError:            # (a forced thread count that cannot hold n points)
        continue
    print('%-10s uniform n=%6d m=%5d  %8.3f ms  %.3f us/step' % (tag, n, m, t, 1e3 * t / m), flush=True)
level = pk.configs.synthetic_pcl('greater', 14336, 12)[0].cuda()[:, :3]
total = 0.0
for m in (4779, 1593, 531):
    t = timeit(lambda: pk.ops.fps(level, m))
    total += t
    print('%-10s greater n=%6d m=%5d  %8.3f ms  %.3f us/step' % (tag, level.shape[0], m, t, 1e3 * t / m), flush=True)
    level = level[pk.ops.fps(level, m).long()]
print('%-10s greater encoder levels total %.3f ms' % (tag, total), flush=True)
if '--coop' in sys.argv:
    for n, m in [(14336, 4779), (28672, 9558)]:
        p = torch.rand((n, 3), device='cuda', generator=g) * 10 - 5
        for wgs in [2, 4, 8, 16]:
            if -(-n // wgs) > 16 * 512:
                continue
            t = timeit(lambda: pk.ops.fps_coop(p, m, n_workgroups=wgs, check=False))
            print('coop%-3d n=%6d m=%5d  %8.3f ms  %.3f us/step' % (wgs, n, m, t, 1e3 * t / m), flush=True)
