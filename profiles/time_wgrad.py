"""Times the weight-gradient GEMM (csrc/wgrad16.hip for the wide decoder layers, csrc/backward.hip wgrad_kernel
otherwise; + the partial reduce) and the forward / data-gradient Linear at the shapes of a training step (HIP
events), and checks the weight gradient against fp64."""
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


for (M, N, K) in [(458752, 416, 832), (68812, 416, 416), (458752, 832, 416), (68812, 832, 416), (17203, 416, 416),
                  (45864, 416, 832), (4096 * 14, 832, 32), (28672, 36, 36), (9558, 72, 72)]:
    g = torch.randn(M, N, device='cuda')
    x = torch.randn(M, K, device='cuda')
    t = timeit(lambda: pk.ops.linear_wgrad(g, x, bias=True, relu_x=True))
    dw, db = pk.ops.linear_wgrad(g, x, bias=True, relu_x=True)
    sub = slice(0, min(M, 20000))
    err = 0.0
    if M <= 70000:
        ref = g.double().t() @ torch.relu(x.double())
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        errb = float((db.double() - g.double().sum(0)).abs().max() / g.double().sum(0).abs().max())
        err = max(err, errb)
    fl = 2.0 * M * N * K
    print('M=%6d N=%4d K=%4d  wgrad %8.1f us %6.1f TFLOP/s  %.3f of fp32 MFMA peak   max rel err vs fp64 %.2g' %
          (M, N, K, 1e3 * t, fl / t * 1e-9, fl / t * 1e-9 / 157.3, err), flush=True)
