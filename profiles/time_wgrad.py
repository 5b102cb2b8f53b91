"""Times the weight-gradient GEMM (csrc/backward.hip wgrad_kernel + reduce) and the forward / data-gradient Linear at the
shapes of a training step (HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occlusions4d_amd as pk  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (M, N, K) in [(17203, 416, 416), (17203, 832, 416), (17203, 416, 832), (4096 * 14, 416, 832), (4096 * 14, 832, 32),
                  (4096 * 14, 416, 32), (68812, 416, 416), (28672, 36, 36), (9558, 72, 72)]:
    g = torch.randn(M, N, device='cuda')
    x = torch.randn(M, K, device='cuda')
    w = torch.randn(N, K, device='cuda') * 0.05
    t = timeit(lambda: pk.ops.linear_wgrad(g, x, bias=True))
    fl = 2.0 * M * N * K
    t2 = timeit(lambda: pk.ops.linear(x, w))
    t3 = timeit(lambda: pk.ops.linear(g, w.t().contiguous()))
    print('M=%6d N=%4d K=%4d  wgrad %8.1f us %6.1f TFLOP/s | forward %8.1f us %6.1f | dgrad %8.1f us %6.1f' %
          (M, N, K, 1e3 * t, fl / t * 1e-9, 1e3 * t2, fl / t2 * 1e-9, 1e3 * t3, fl / t3 * 1e-9), flush=True)
